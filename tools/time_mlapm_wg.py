"""MLAPM forward and rollout frame with 4-wave / 16-wave workgroups (PIML_MLAPM_WG16_MIN picks the changeover).  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib
from piml_amd.scenes import synthetic_gc_scene
from piml_amd.models.mlapm import MLAPM
dev = 'cuda:0'
gc = dict(version='GC', tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)
m = MLAPM(**gc)
for N in (512, 1024, 1536, 2048, 3000, 4005):
    sc = synthetic_gc_scene(N, 0, seed=0, nan_frac=0.0)
    a = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    for _ in range(5):
        ops.mlapm_step(*a, 0.08, 0.3, **gc)
    tm = _lib.StreamTimer(); tm.start()
    for _ in range(100):
        ops.mlapm_step(*a, 0.08, 0.3, **gc)
    tm.stop(); fwd = tm.elapsed_ms() * 10
    m.rollout(*a, 0.08, 0.3, 60)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.rollout(*a, 0.08, 0.3, 800)
    torch.cuda.synchronize(); fr = (time.perf_counter() - t0) / 800 * 1e6
    print(f'WG16_MIN={os.environ.get("PIML_MLAPM_WG16_MIN", "default")} N={N}: forward {fwd:6.1f} us   rollout frame {fr:6.1f} us', flush=True)
