"""pinnsf_res forward + backward step at the bench shape (development aid; run under rocprofv3 --stats for the kernel mix)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from piml_amd.scenes import synthetic_gc_scene
dev = torch.device('cuda:0')
N, M = 4096, 2000
scene = synthetic_gc_scene(N, M, seed=0)
_margs = bench.model_args
bench.model_args = lambda: types.SimpleNamespace(**dict(_margs().__dict__, res_hidden_layers=3))
st = bench.Step(scene, N, N, 0, M, dev, None, False, False, True, model_name='PINNSF_residual')
st.capture()
for _ in range(30):
    st.run()
torch.cuda.synchronize()
