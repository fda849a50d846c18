"""Forward + backward step of every PINNSF variant at the bench shape (4096 agents, k = 6 / 10), captured graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import piml_amd.models.model as MODEL
from piml_amd.scenes import synthetic_gc_scene

dev = torch.device('cuda:0')
N, M = 4096, 2000
scene = synthetic_gc_scene(N, M, seed=0)
import types
_margs = bench.model_args
bench.model_args = lambda: types.SimpleNamespace(**dict(_margs().__dict__, res_hidden_layers=3))
NAMES = ('PINNSF_multitask', 'PINNSF', 'PINNSF_bottleneck_multitask', 'PINNSF_bottleneck', 'PINNSF_residual')
if '--models' in sys.argv:          # e.g. --models PINNSF_bottleneck_multitask (the program to put under rocprofv3 for one model's kernel mix)
    NAMES = tuple(sys.argv[sys.argv.index('--models') + 1].split(','))
for name in NAMES:
  for train in (False, True):
    try:
        st = bench.Step(scene, N, N, 0, M, dev, None, False, False, True, model_name=name, train_mode=train)
        st.capture()
        for _ in range(20):
            st.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            st.run()
        torch.cuda.synchronize()
        print(f'{name:32s} {"train() dropout 0.5" if train else "eval()":20s} {st.mode}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step', flush=True)
    except Exception as ex:   # noqa
        print(name, 'failed:', ex)
