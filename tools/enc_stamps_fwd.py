#!/usr/bin/env python3
"""In-kernel stamps of enc_fwd_sum_x3_kernel (diagnostic build): python -m piml_amd.build --variant encstamps encoder_x3.hip:-DPIML_ENC_STAMPS ;
run with PIML_LIB=piml_amd/libpiml_hip_encstamps.so.  Thread 0 = wave 0 of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net
NAMES = ['entry -> weight image requested, landed and written to LDS', 'barrier', 'layer 1 + h1 sign bits', 'split', 'layer 2 (exchanged)',
         'h2 sign bits + h2 rows (requests)', 'register sums + their stores (requests)', 'stores landed']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for _ in range(3):
    res = ops.fused_pinnsf(brs, sf, 0.5, fold_epilogue=True, head=head, sums=True)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (512 * 16))()
L.piml_enc_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_enc_stamps(buf) == 0
st = np.array(buf[:], dtype=np.float64).reshape(512, 16)[:256, :9]
st = st[st[:, 8] > st[:, 0]]
d = np.diff(st, axis=1)
print(f'{n} agents, {len(st)} workgroups with a tile for wave 0; shader clocks between stamps (median / max)')
for i, name in enumerate(NAMES):
    print(f'  {name:60s} {np.median(d[:, i]):8.0f} {d[:, i].max():8.0f}')
print(f'  entry -> last stamp: median {np.median(st[:, 8] - st[:, 0]):.0f}, max {(st[:, 8] - st[:, 0]).max():.0f}')
