#!/usr/bin/env python3
"""In-kernel stamps of the two-crew backward of the sums path (encoder_bwd5.hip): cycles between the stamps, every wave.
Build: python -m piml_amd.build --variant f5stamps encoder_bwd5.hip:-DPIML_F5_STAMPS ; run with PIML_LIB=piml_amd/libpiml_hip_f5stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net, run
NAMES = {15: 'prologue (once)', 0: 'barrier wait', 1: 'products + fills', 2: 'A: G1 mask + tile', 6: 'A: H1 mma + G2 | B: g_x', 7: 'A: H1 pieces', 5: 'A: stage + requests | B: dW1', 3: 'loop exit (once)', 4: 'B: last tile (once)', 11: 'epilogue (once)'}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for _ in range(3):
    run(brs, sf, head, wa, 0.5, True)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (256 * 8 * 16))()
L.piml_f5_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_f5_stamps(buf) == 0
st = np.array(buf[:], dtype=np.float64).reshape(256, 8, 16)
tiles = n * 16 / 32 / 256
med = np.median(st, axis=0)                # [wave][stamp]
print(f'{n} agents, {tiles:.1f} tiles per workgroup; cycles of waves 0 .. 7 (A: 0-3, B: 4-7; median over 256 workgroups)')
for i, name in NAMES.items():
    once = 'once' in name
    print(f'  {name:24s} ' + ' '.join(f'{med[w, i] / (1 if once else tiles):7.0f}' for w in range(8)) + ('' if once else '  per tile'))
tot = st.sum(axis=2)
mx = np.max(st, axis=0); mn = np.min(st, axis=0)
for i in (0, 1, 6, 5):
    print(f'  [{i}] min/max per tile     ' + ' '.join(f'{mn[w, i] / tiles:5.0f}/{mx[w, i] / tiles:<5.0f}' for w in (0, 3, 4, 7)))
print('  total per wave           ' + ' '.join(f'{tot[:, w].mean():7.0f}' for w in range(8)) + f' (max {tot.max():.0f})')

