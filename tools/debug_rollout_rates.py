import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piml_amd.scenes import synthetic_gc_scene, synthetic_rollout_data
from piml_amd.models.mlapm import MLAPM
dev='cuda:0'
sc = synthetic_gc_scene(4096, 2000, seed=0)
ok = ~np.isnan(sc['position'][:, 0])
m = MLAPM(version='GC', tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)
for name, sel in (('present only', ok), ('all rows', np.ones_like(ok))):
    a = [torch.tensor(sc[k][sel], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    for T in (600, 2000):
        m.rollout(*a, 0.08, 0.3, 60)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tp, tv = m.rollout(*a, 0.08, 0.3, T)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, a[0].shape[0], T, f'{dt/T*1e6:.1f} us/frame', 'absent at end', int(torch.isnan(tp[-1,:,0]).sum()))
from piml_amd.models.simulators import BaseSimulator
from piml_amd.main import get_args
a = get_args(['--dataset_name', 'gc1560', '--model', 'pinnsf_m'])
a.ped_feature_dim, a.obs_feature_dim, a.self_feature_dim, a.device = 6, 6, 7, dev
a.exp_name, a.model_name_suffix = 'bench', 'bench'
data = synthetic_rollout_data(4096, 2000, 200, dev)
sim = BaseSimulator(a); sim.model.eval()
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.get_multiple_rollouts(data, 0, load_model=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('pinnsf rollout', f'{dt/200*1e6:.1f} us/frame')
