"""Per-wave work statistics of relfeat fwd (evals / drain rounds / insertions / candidates)
from the -DPIML_RELFEAT_STATS build (development aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from piml_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libpiml_hip_stats.so')
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
for N, M in ((4096, 2000), (16384, 2000)):
    sc = synthetic_gc_scene(N, M, seed=0)
    t = [torch.tensor(sc[k], device='cuda:0') for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    stats = torch.zeros(N, 4, dtype=torch.int32, device='cuda:0')
    os.environ['PIML_RELFEAT_STATS_PTR'] = str(stats.data_ptr())
    ops.relative_features(*t)
    torch.cuda.synchronize()
    s = stats.cpu().numpy()
    for q, name in enumerate(('evals', 'drain rounds', 'insertions', 'candidates')):
        print(f'N={N} {name:13s}: mean {s[:, q].mean():7.2f}  p50 {np.percentile(s[:, q], 50):6.0f}  p99 {np.percentile(s[:, q], 99):6.0f}  max {s[:, q].max():6d}')
