"""Repeat forward + backward of the fused encoders at several shapes and compare every output / gradient bitwise with the
first repetition (races in the split-product kernels would show as run-to-run differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_encoder_gpu import make_branch

L = _lib.lib()
for shapes in ([(122 * 20, 6, 6), (122 * 20, 10, 6)], [(4096, 12, 6), (4096, 4, 6)], [(37, 10, 6)], [(700, 6, 6), (300, 10, 6)],
               [(5000, 7, 5), (33, 3, 8)], [(2440, 6, 6), (2440, 2, 6)]):
    brs = [make_branch(n, k, d, seed=10 * i + n % 7) for i, (n, k, d) in enumerate(shapes)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    first, bad = None, 0
    for rep in range(40):
        outs = ops.fused_encoders(brs)
        loss = sum((m * 1e-2).sum() + p.square().sum() for m, p in outs)
        res = [t.detach().clone() for o in outs for t in o] + list(torch.autograd.grad(loss, leaves))
        if first is None:
            first = res
        else:
            diff = [i for i, (a, b) in enumerate(zip(first, res)) if not torch.equal(a, b)]
            bad += len(diff)
            if diff and bad < 20:
                print('   rep', rep, 'differs in tensors', diff, [float((first[i] - res[i]).abs().max()) for i in diff[:4]])
    print(shapes, 'mismatching tensors over 39 repeats:', bad, 'nan:', sum(int(torch.isnan(t).any()) for t in first))
