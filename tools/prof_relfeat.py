"""Launch relfeat fwd a few times for rocprofv3 (development aid): N M thr [reps]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
N, M, thr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sc = synthetic_gc_scene(N, M, seed=0)
t = [torch.tensor(sc[k], device='cuda:0') for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
for _ in range(reps):
    ops.relative_features(*t, dist_threshold_ped=thr, dist_threshold_obs=thr)
torch.cuda.synchronize()
