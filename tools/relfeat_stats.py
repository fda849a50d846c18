"""Per-wave work statistics and in-kernel phase stamps of relfeat fwd from the -DPIML_RELFEAT_STATS
build (development aid; the stamps perturb the kernel, read shares not totals)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from piml_amd import _lib
# run with PIML_LIB=piml_amd/libpiml_hip_stats.so (python -m piml_amd.build --variant stats relfeat.hip:-DPIML_RELFEAT_STATS)
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
for N, M in ((4096, 2000),):
    sc = synthetic_gc_scene(N, M, seed=0)
    t = [torch.tensor(sc[k], device='cuda:0') for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    stats = torch.zeros(2 * N, 12, dtype=torch.int32, device='cuda:0')      # split launches: the obstacle workgroups' rows in the second half
    os.environ['PIML_RELFEAT_STATS_PTR'] = str(stats.data_ptr())
    for _ in range(3):
        ops.relative_features(*t)
    torch.cuda.synchronize()
    s2 = stats.cpu().numpy()
    s = s2[:N]
    alive = ~np.isnan(sc['position'][:, 0])
    for q, name in enumerate(('evals', 'drain rounds', 'insertions', 'candidates')):
        print(f'N={N} {name:13s}: mean {s[alive, q].mean():7.2f}  p50 {np.percentile(s[alive, q], 50):6.0f}  p99 {np.percentile(s[alive, q], 99):6.0f}  max {s[alive, q].max():6d}')
    names = ['entry->ped tile staged', 'ped pass', 'obs tile staged (incl. barrier wait)', 'obs pass']
    st = s[alive, 4:9].astype(np.int64)
    for q, name in enumerate(names):
        dphase = st[:, q + 1] - st[:, q]
        print(f'  phase {name:38s}: median {np.median(dphase):8.0f} cycles  p90 {np.percentile(dphase, 90):8.0f}  max {dphase.max():8d}')
    print('  total to end of obs pass: median', np.median(st[:, 4]), 'cycles')
    if s2[N:].any():
        print('  split launch: the rows above are the pedestrian workgroups (entry, staged, pass done); obstacle workgroups:')
        so = s2[N:][alive, 4:9].astype(np.int64)
        for q, name in enumerate(('entry->obs tile staged', 'obs pass')):
            d = so[:, q + 1] - so[:, q]
            print(f'  phase {name:38s}: median {np.median(d):8.0f} cycles  p90 {np.percentile(d, 90):8.0f}  max {d.max():8d}')
