"""Device timing of the MLAPM / collision kernels launched back to back (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene

dev = 'cuda:0'


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for N in (1024, 4096, 16384):
    sc = synthetic_gc_scene(N, 0, seed=0, nan_frac=0.0)
    p, v, v0, d = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    for ver, pr in (('raw', dict(tau=0.5, A=7.55, B=-3.0)), ('GC', dict(tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)),
                    ('UCY', dict(tau=5 / 6, A=10.67, B=-3.33, C=0.5, theta=20))):
        us = timeit(lambda: ops.mlapm_step(p, v, v0, d, 0.08, 0.3, version=ver, **pr))
        # backward through the C entry the operator calls (an eager autograd.grad costs ~95 us of host time per call, more
        # than the kernels below 8192 agents)
        from piml_amd import _lib
        L = _lib.lib()
        q = dict(C=0.0, D=0.0, theta=0.0); q.update(pr)
        w = torch.randn(N, 2, device=dev)
        outs = [torch.empty(N, 2, device=dev), torch.empty(N, 2, device=dev), torch.empty(N, device=dev), torch.empty(N, 2, device=dev)]
        var = ops.MLAPM_VARIANTS[ver]
        need = int(L.piml_mlapm_bwd_workspace_floats(N, var))
        ws = torch.empty(max(need, 1), device=dev)
        args = [w.data_ptr(), p.data_ptr(), v.data_ptr(), v0.data_ptr(), d.data_ptr(), N, var, q['tau'], q['A'], q['B'], q['C'], q['D'],
                float(q['theta']), 0.3, 0.08] + [o.data_ptr() for o in outs] + [ws.data_ptr() if need else None, need, None]
        usb = timeit(lambda: L.piml_mlapm_step_bwd_ws(*args), reps=50)
        print(f'MLAPM {ver:3s} N={N}: fwd {us:8.1f} us ({N * N / us * 1e6:.3e} pairs/s, alg {16 * N * N / us / 1e3:.0f} GB/s)   bwd {usb:8.1f} us')
    if N <= 4096:
        pc = torch.tensor(synthetic_gc_scene(N, 0, seed=1, channels=4)['position'], device=dev)
        print(f'collision_counts S=4 N={N} (2 thr): {timeit(lambda: ops.collision_counts(pc, (0.5, 0.25)), 20):.1f} us;  '
              f'collision_detection matrix S=4: {timeit(lambda: ops.collision_detection(pc, 0.5), 20):.1f} us')

# the general path of collision_counts (more than 25 slices): the evaluation's (t, N, 2) rollouts, S = 750 frames
for N in (122, 248, 1024):
    g = torch.Generator().manual_seed(0)
    pc = (torch.rand(750, N, 2, generator=g) * (12.0 if N < 1000 else 40.0)).to(dev)
    pc[:, ::11] = float('nan')
    print(f'collision_counts general path S=750 N={N} (2 thr): {timeit(lambda: ops.collision_counts(pc, (0.5, 0.25)), 10):.1f} us;  '
          f'matrix path collision_detection(...).sum(-1) x 2: '
          f'{timeit(lambda: (ops.collision_detection(pc, 0.5).sum(-1), ops.collision_detection(pc, 0.25).sum(-1)), 5):.1f} us')
