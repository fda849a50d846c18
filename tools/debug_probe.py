import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import GOLDEN, golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
from piml_amd.data import dataset as DATASET
from piml_amd.models import simulators as SIM
import piml_amd.models.model as MODEL
for case in ('gc', 'ucy'):
    g = golden('mainflow_' + case)
    pr = np.load(f'tests/golden/mainflow_probe_{case}.npz')
    DATA = os.path.join(GOLDEN, 'data')
    argv = T.COMMON + T.CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                                       '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml')]
    args = MAIN.get_args(argv)
    real = DATASET.TimeIndexedPedDataset2(); real.load_data(args.ft_data_config); real.build_dataset(args)
    sim = SIM.BaseSimulator(args); sim.set_ft_model(args)
    sd = {k[len('best_ft_full/'):]: torch.tensor(g[k]) for k in g.files if k.startswith('best_ft_full/')}
    sim.model.load_state_dict(sd); sim.model.eval()
    d = real.test_data[0]
    t = 25
    def cmp(name, a, b):
        a = np.nan_to_num(a.detach().cpu().numpy() if torch.is_tensor(a) else a); b = np.nan_to_num(b)
        print(f'  {case} {name:10s} shape {a.shape} vs {b.shape}: max abs diff {np.abs(a - b).max():.3e}')
    cmp('ped', d.ped_features[t], pr['ped']); cmp('obs', d.obs_features[t], pr['obs']); cmp('selff', d.self_features[t], pr['selff'])
    ins = [torch.tensor(pr[k], device='cuda:0') for k in ('ped', 'obs', 'selff')]
    with torch.no_grad():
        for fe in (True, False):
            MODEL.FUSED_ENCODER = fe
            cmp(f'acc fe={fe}', sim.model(*ins)[0], pr['acc'])
            cmp(f'acc own-in fe={fe}', sim.model(d.ped_features[t], d.obs_features[t], d.self_features[t])[0], pr['acc'])
        MODEL.FUSED_GLUE = False
        cmp('acc plain', sim.model(*ins)[0], pr['acc'])
        MODEL.FUSED_GLUE = True; MODEL.FUSED_ENCODER = True
