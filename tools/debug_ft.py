import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, 'tests')
from conftest import GOLDEN, golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
case = 'gc'
g = golden('mainflow_' + case)
init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')}
DATA = os.path.join(GOLDEN, 'data')
argv = T.COMMON + T.CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                               '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml')] + sys.argv[1:]
MAIN.main(argv, init_state=init)
run = MAIN.LAST_RUN
for h in run['finetune_history']:
    print({k: v for k, v in h.items() if not isinstance(v, dict)})
print('ref', g['ft/train'], g['ft/train_collisions'], g['ft/val'])
