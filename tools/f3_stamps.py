#!/usr/bin/env python3
"""Where a tile of the one-pass encoder backward (piml_amd/csrc/encoder_bwd3.hip) spends its cycles: runs the bench-shape
backward on a PIML_F3_STAMPS build (PIML_LIB=...; python -m piml_amd.build --variant stamps encoder_bwd3.hip:-DPIML_F3_STAMPS)
and prints wave 0's cycles between the stamps, median over the workgroups, per tile."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from piml_amd import ops, _lib  # noqa: E402
from tools.time_encoder import branch  # noqa: E402

NAMES = ['B1 wait', 'region X: requests, g_x store, H1, layer A', 'mask + split G2 + M writes', 'B2 wait',
         'layer B (+ next G3)', '', 'dW2 (+ G1 mask, dW1, g_x)', '', '', '', '', 'epilogue of phase 1 (slot stores)', 'phase 2 (dW3), whole', '', '', 'prologue']


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    L = _lib.lib()
    L.piml_encoder_fused_bwd(1)
    brs = [branch(n, 6, 1), branch(n, 10, 2)]
    outs = ops.fused_encoders(brs)
    gp = [torch.randn_like(p) for _, p in outs]
    for _ in range(3):
        o = ops.fused_encoders(brs)
        torch.autograd.backward([p for _, p in o], gp)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 16))()
    L.piml_f3_stamps.argtypes = [ctypes.c_void_p]
    assert L.piml_f3_stamps(buf) == 0
    st = np.array(buf[:], dtype=np.float64).reshape(256, 16)
    tiles = n * 16 / 32 / 256
    med = np.median(st, axis=0)
    print(f'{n} agents, {tiles:.1f} tiles per workgroup; cycles of wave 0 (median over 256 workgroups)')
    for i, name in enumerate(NAMES):
        if name:
            per = med[i] / tiles if i not in (11, 12, 15) else med[i]
            print(f'  {name:32s} {per:10.0f}' + ('  per tile' if i not in (11, 12, 15) else '  once'))
    print(f'  total per workgroup {st.sum(axis=1).mean():.0f} (max {st.sum(axis=1).max():.0f})')


if __name__ == '__main__':
    main()
