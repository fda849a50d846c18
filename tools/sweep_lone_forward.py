#!/usr/bin/env python3
"""Lone encoder forward (no backward: torch.no_grad) over agent counts, few-rows kernels (four waves per tile) against the
many-rows kernels (one wave per tile, wave-major tile order): piml_encoder_split_tiles(huge) / (0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib
from time_encoder import branch, timed
L = _lib.lib()
for n in (32, 64, 122, 256, 488, 976, 1500, 2048, 3000, 4096):
    brs = [branch(n, 6, 1), branch(n, 10, 2)]
    tiles = (n * 6 + 31) // 32 + (n * 10 + 31) // 32
    def fwd():
        with torch.no_grad():
            return ops.fused_encoders(brs)
    res = []
    for bound in (1 << 20, 0):
        L.piml_encoder_split_tiles(bound)
        fwd(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                    # 20 forwards per graph: the host's 80 us per eager call would hide the kernels
        with torch.cuda.graph(g):
            for _ in range(20):
                fwd()
        res.append(timed(g.replay, reps=20) / 20)
    L.piml_encoder_split_tiles(-2)
    print(f'{n:5d} agents {tiles:5d} tiles: few-rows {res[0]:6.1f} us, many-rows {res[1]:6.1f} us')
