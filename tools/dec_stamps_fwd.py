#!/usr/bin/env python3
"""In-kernel stamps of dec_fwd_head_sum_kernel (diagnostic build, see tools/dec_stamps.py): decoder workgroups and head workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net
DEC = ['entry -> loads issued', 'loads landed', 'layer 1 products + LDS', 'barrier 1', 'completed sums back, layer 2 + LDS (stores landed)', 'barrier 2',
       'predictor + LDS', 'barrier 3', 'epilogue (store landed)']
HEAD = ['first weight request -> row loads issued', 'rows landed', 'split', 'products (both blocks) + second layer', 'store landed']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
with torch.no_grad():
    pass
for _ in range(3):
    res = ops.fused_pinnsf(brs, sf, 0.5, fold_epilogue=True, head=head, sums=True)     # forward only: the last decoder launch is the forward's
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (1024 * 16))()
L.piml_dec_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_dec_stamps(buf) == 0
ndec = 2 * ((n + 31) // 32)
nhead = ((n * 6 + 31) // 32 + 3) // 4
st = np.array(buf[:], dtype=np.float64).reshape(1024, 16)
for name, rows, labels in (('decoder', st[:ndec, :10], DEC), ('head', st[ndec:ndec + nhead, :6], HEAD)):
    d = np.diff(rows, axis=1)
    print(f'{name}: {len(rows)} workgroups; shader clocks of thread 0 between stamps (median / max)')
    for i, lab in enumerate(labels):
        print(f'  {lab:56s} {np.median(d[:, i]):8.0f} {d[:, i].max():8.0f}')
    print(f'  entry -> last stamp: median {np.median(rows[:, -1] - rows[:, 0]):.0f}, max {(rows[:, -1] - rows[:, 0]).max():.0f}')
