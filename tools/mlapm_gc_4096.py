"""MLAPM.step (GC law) forward + analytic backward on the present agents of the bench scene, 30 times: the launches the
PMC pass of tools/profile_round.sh counts for `secondary.mlapm_gc_step` (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
sc = synthetic_gc_scene(4096, 2000, seed=0)
ok = ~np.isnan(sc['position'][:, 0])
leaves = [torch.tensor(sc[k][ok], device='cuda:0').requires_grad_(True) for k in ('position', 'velocity', 'desired_speed', 'destination')]
gc = dict(version='GC', tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)
w = torch.ones(int(ok.sum()), 2, device='cuda:0')
for _ in range(30):
    act = ops.mlapm_step(*leaves, 0.08, 0.3, **gc)
    torch.autograd.grad(act, leaves, w)
torch.cuda.synchronize()
