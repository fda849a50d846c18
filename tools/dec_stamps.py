#!/usr/bin/env python3
"""In-kernel stamps of dec_bwd_split_kernel (diagnostic build): shader clocks of thread 0 of every workgroup between the phases.
Build: python -m piml_amd.build --variant decstamps decoder.hip:-DPIML_DEC_STAMPS ; run with PIML_LIB=piml_amd/libpiml_hip_decstamps.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from piml_amd import ops, _lib
from test_sums_gpu import make_net, run
NAMES = ['entry -> chain loads issued', 'chain loads landed', 'g2, g1 products + LDS part written', 'barrier 1', 'masked g1, g_pooled products, stores landed',
         'desired-force part', 'barrier 2', 'dW operands requested', 'dW operands landed', 'dW products issued', 'dW products done', 'slot stores issued',
         'slot stores landed']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for _ in range(3):
    run(brs, sf, head, wa, 0.5, True)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * (1024 * 16))()
L.piml_dec_stamps.argtypes = [ctypes.c_void_p]
assert L.piml_dec_stamps(buf) == 0
nwg = 2 * ((n + 31) // 32)
st = np.array(buf[:], dtype=np.float64).reshape(1024, 16)[:nwg, :14]
d = np.diff(st, axis=1)
print(f'{n} agents, {nwg} workgroups; shader clocks of thread 0 between stamps (median / max over the workgroups)')
for i, name in enumerate(NAMES):
    print(f'  {name:48s} {np.median(d[:, i]):8.0f} {d[:, i].max():8.0f}')
print(f'  entry -> last stamp: median {np.median(st[:, 13] - st[:, 0]):.0f}, max {(st[:, 13] - st[:, 0]).max():.0f};  '
      f'first entry -> last exit over the launch: {st[:, 13].max() - st[:, 0].min():.0f};  entry spread {st[:, 0].max() - st[:, 0].min():.0f}')
