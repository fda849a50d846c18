#!/usr/bin/env python3
"""In-kernel stamps of enc_fwd_x3_kernel (message path / PIML_POOL_MSGS), diagnostic build (see tools/enc_stamps_fwd.py):
python tools/enc_stamps_msgs.py [agents] [sums 0|1]  -- a dropout mask drawn in the kernel (p = 0.5)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net
NAMES = ['entry -> weight image staged (W3 part requested)', 'W3 part landed at once (exchanged form) / nothing', 'layer 1 + sign bits', 'split + layer 2 + sign bits + h2 rows',
         'W3 landed, LO / tail requests, keep words (draw, gather), split', 'layer 3 + stores / sums', '-', 'stores landed']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sums = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for br in brs:
    br['keep_bits'] = ('draw', 0.5); br['scale'] = 4.0
for _ in range(3):
    res = ops.fused_pinnsf(brs, sf, 0.5, fold_epilogue=True, head=head, sums=sums)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (512 * 16))()
L.piml_enc_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_enc_stamps(buf) == 0
st = np.array(buf[:], dtype=np.float64).reshape(512, 16)[:256, :9]
d = np.diff(st, axis=1)
print(f'{n} agents, sums={sums} (messages returned: {res[1][0] is not None}); shader clocks of wave 0 between stamps (median / max over 256 workgroups)')
for i, name in enumerate(NAMES):
    print(f'  {name:66s} {np.median(d[:, i]):8.0f} {d[:, i].max():8.0f}')
print(f'  entry -> last stamp: median {np.median(st[:, 8] - st[:, 0]):.0f}, max {(st[:, 8] - st[:, 0]).max():.0f}')
