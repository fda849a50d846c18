#!/usr/bin/env python3
"""tools/f3_stamps.py for the SUMS form of the one-pass backward (fused_pinnsf(sums=True)): cycles of wave 0 between the stamps.
Build: python -m piml_amd.build --variant stamps encoder_bwd3.hip:-DPIML_F3_STAMPS ; run with PIML_LIB=piml_amd/libpiml_hip_stamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net, run
NAMES = {15: 'prologue (once)', 0: 'B1 wait', 1: 'region X', 2: 'mask + split G2 + M writes', 3: 'B2 wait', 4: 'layer B', 6: 'dW2 (+ G1 mask, dW1, g_x)',
         11: 'epilogue (once)'}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sums = int(os.environ.get('SUMS', '1'))
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for _ in range(3):
    run(brs, sf, head, wa, 0.5, bool(sums))
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (256 * 64))()
L.piml_f3_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_f3_stamps(buf) == 0
st = np.array(buf[:], dtype=np.float64).reshape(256, 4, 16)
tiles = n * 16 / 32 / 256
med = np.median(st, axis=0)                # [wave][stamp]
print(f'{n} agents sums={sums}, {tiles:.1f} tiles per workgroup; cycles of waves 0 .. 3 (median over 256 workgroups)')
for i, name in NAMES.items():
    once = 'once' in name
    print(f'  {name:32s} ' + ' '.join(f'{med[w, i] / (1 if once else tiles):8.0f}' for w in range(4)) + ('' if once else '  per tile'))
tot = st.sum(axis=2)
print('  total per wave ' + ' '.join(f'{tot[:, w].mean():8.0f}' for w in range(4)) + f' (max {tot.max():.0f})')
