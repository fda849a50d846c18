import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import GOLDEN, golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
from piml_amd.data import dataset as DATASET
from piml_amd.models import simulators as SIM
case = sys.argv[1] if len(sys.argv) > 1 else 'ucy'
g = golden('mainflow_' + case)
DATA = os.path.join(GOLDEN, 'data')
argv = T.COMMON + T.CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                                   '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml')]
args = MAIN.get_args(argv)
real = DATASET.TimeIndexedPedDataset2()
real.load_data(args.ft_data_config)
real.build_dataset(args)
sim = SIM.BaseSimulator(args)
sim.set_ft_model(args)
sd = {k[len('best_ft_full/'):]: torch.tensor(g[k]) for k in g.files if k.startswith('best_ft_full/')}
sim.model.load_state_dict(sd)
sim.model.eval()
d = real.test_data[0]
skip = args.skip_frames
want = g['test/rollout_head']
wmask = g['test/mask_head']
for mode in ('fused+graph', 'eager-torch'):
    with torch.no_grad():
        if mode == 'fused+graph':
            pred = sim.get_multiple_rollouts(d, t_start=skip, load_model=False)
        else:
            pred = sim.get_multiple_rollouts(d, t_start=skip, load_model=False, use_graph=False, fused=False)
    head = pred.position[:want.shape[0]].cpu().numpy()
    m = pred.mask_p[:want.shape[0]].cpu().numpy()
    print('==', mode)
    for t in range(skip, want.shape[0]):
        e = np.abs(np.nan_to_num(head[t]) - np.nan_to_num(want[t])).max(-1)
        nanmis = np.isnan(head[t, :, 0]) != np.isnan(want[t, :, 0])
        if e.max() > 1e-4 or nanmis.any():
            bad = np.nonzero((e > 1e-4) | nanmis)[0]
            print(f'frame {t}: max err {e.max():.3e}, agents {bad[:10]}, nan mismatch {np.nonzero(nanmis)[0][:10]}')
            for a in bad[:3]:
                print('   agent', a, 'here', head[t, a], 'ref', want[t, a], 'prev here', head[t - 1, a], 'prev ref', want[t - 1, a],
                      'mask here/ref', m[t, a], wmask[t, a], 'data pos', d.position[t, a].cpu().numpy(), 'mask_p', int(d.mask_p[t, a]), int(d.mask_p_pred[t, a]))
            break
    else:
        print('no deviation > 1e-4 in the head')
