#!/usr/bin/env python3
"""forward + backward of the sums path (fused_pinnsf(sums=True)) at cfg3's shape, repeated: the program to put under
tools/prof_script.sh / tools/pmc_script.sh for the kernels of the sums step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_sums_gpu import make_net, run
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
brs, sf, head, wa, g = make_net(n, (6, 10), True, seed=1)
for _ in range(reps):
    run(brs, sf, head, wa, 0.5, True)
torch.cuda.synchronize()
