"""MLAPM backward, the two C entries back to back on one scene: piml_mlapm_step_bwd (both roles of a pair in the owning
wavefront) against piml_mlapm_step_bwd_ws (every ordered pair once, rotating focal agents; PIML_MLAPM_BWD_SPLIT=1|2|4 picks
the wavefronts per block pair).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import _lib, ops
from piml_amd.scenes import synthetic_gc_scene

dev = 'cuda:0'
LAWS = (('raw', dict(tau=0.5, A=7.55, B=-3.0, C=0.0, D=0.0, theta=0.0)), ('GC', dict(tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56.0)),
        ('UCY', dict(tau=5 / 6, A=10.67, B=-3.33, C=0.5, D=0.0, theta=20.0)))
L = _lib.lib()
for N in [int(a) for a in sys.argv[1:]] or (2048, 4096, 8192, 16384):
    sc = synthetic_gc_scene(N, 0, seed=0, nan_frac=0.0)
    p, v, v0, d = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    w = torch.randn(N, 2, device=dev)
    out = [torch.empty(N, 2, device=dev), torch.empty(N, 2, device=dev), torch.empty(N, device=dev), torch.empty(N, 2, device=dev)]
    for ver, pr in LAWS:
        var = ops.MLAPM_VARIANTS[ver]
        args = [w.data_ptr(), p.data_ptr(), v.data_ptr(), v0.data_ptr(), d.data_ptr(), N, var, pr['tau'], pr['A'], pr['B'], pr['C'],
                pr['D'], pr['theta'], 0.3, 0.08] + [o.data_ptr() for o in out]
        need = int(L.piml_mlapm_bwd_workspace_floats(N, var))
        ws = torch.empty(max(need, 1), device=dev)

        def timed(fn, reps=50):
            for _ in range(5):
                fn()
            tm = _lib.StreamTimer()
            tm.start()
            for _ in range(reps):
                fn()
            tm.stop()
            return tm.elapsed_ms() * 1e3 / reps
        two = timed(lambda: L.piml_mlapm_step_bwd(*args, None))
        once = timed(lambda: L.piml_mlapm_step_bwd_ws(*args, ws.data_ptr(), need, None)) if need else float('nan')
        print(f'MLAPM {ver:3s} N={N:6d}: two-role kernel {two:8.1f} us   once per pair {once:8.1f} us   '
              f'(workspace {need * 4 / 1e6:.1f} MB, split {os.environ.get("PIML_MLAPM_BWD_SPLIT", "default")})', flush=True)
