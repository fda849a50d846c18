"""Forward + backward step (bench.Step, captured) by agent count with the few-rows kernels (four waves per tile) against the
many-rows kernels (one wave per tile, one-pass backward, sums path where served): where is the cross-over?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from piml_amd import _lib
from piml_amd.scenes import synthetic_gc_scene
dev = torch.device('cuda', 0)
L = _lib.lib()
for n in [int(x) for x in os.environ.get("NS", "256,512,768,1024,1280,1536,2048,3072").split(",")]:
    scene = synthetic_gc_scene(n, 2000, seed=0)
    row = {'agents': n, 'tiles': (n * 6 + 31) // 32 + (n * 10 + 31) // 32}
    for label, thr in (('few_rows', 1 << 30), ('many_rows', 0)):
        L.piml_encoder_split_tiles(thr)
        for msg in (0, 1):
            st = bench.Step(scene, n, n, 0, 2000, dev, None, False, False, True, messages=bool(msg))
            st.capture()
            el = st.time_steps(100, 20)
            row[f'{label}_msg{msg}'] = round(el / 100 * 1e6, 1)
            del st
    print(json.dumps(row), flush=True)
