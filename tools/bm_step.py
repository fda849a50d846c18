#!/usr/bin/env python3
"""The bench step (forward + backward at cfg3) of one model class, graph-replayed: the program to put under rocprofv3
(tools/prof_script.sh tools/bm_step.py [model class] [train 0|1])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch
import bench
from piml_amd.scenes import synthetic_gc_scene
name = sys.argv[1] if len(sys.argv) > 1 else 'PINNSF_bottleneck_multitask'
train = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
N, M = 4096, 2000
dev = torch.device('cuda:0')
scene = synthetic_gc_scene(N, M, seed=0)
st = bench.Step(scene, N, N, 0, M, dev, None, False, False, True, model_name=name, train_mode=train)
st.capture()
el = st.time_steps(200, 20)
print(f'{name} train={train}: {el / 200 * 1e3:.4f} ms/step ({st.mode})')
