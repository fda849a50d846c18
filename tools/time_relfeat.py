"""Quick device timing of the relfeat kernels (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene, pair_count

for N, M in ((1024, 100), (4096, 2000), (16384, 2000)):
    sc = synthetic_gc_scene(N, M, seed=0)
    t = [torch.tensor(sc[k], device='cuda:0') for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    for _ in range(5):
        out = ops.relative_features(*t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        out = ops.relative_features(*t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f'N={N} M={M}: {us:.1f} us/call incl. host launch, {pair_count(N, M) / us * 1e6:.3e} pairs/s')
