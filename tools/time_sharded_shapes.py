"""relfeat fwd/bwd time for one rank's share under agent-block sharding (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
dev = 'cuda:0'
for N, fc in ((16384, 2048), (8192, 4096), (16384, 4096), (32768, 4096)):
    sc = synthetic_gc_scene(N, 2000, seed=0)
    state = torch.tensor(__import__('numpy').concatenate([sc[k] for k in ('position', 'velocity', 'acceleration')], -1), device=dev)
    dest = torch.tensor(sc['destination'][:fc], device=dev); obs = torch.tensor(sc['obstacles'], device=dev)
    outs = ops.relative_features_packed(state, dest, obs, 0, fc, return_index=True)
    for _ in range(5): ops.relative_features_packed_into(outs, state, dest, obs, 0, fc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.relative_features_packed_into(outs, state, dest, obs, 0, fc)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    pairs = fc * (N + 2000)
    print(f'N={N} focal block {fc}: {us:.1f} us  ({pairs / us * 1e6:.3e} pairs/s per GPU)')
