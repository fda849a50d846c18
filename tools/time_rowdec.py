"""The row decoder of the bottleneck variants at the reference's row counts (4096 agents x 6 + 4096 x 10 neighbour rows... cfg3:
65536 rows), forward and backward launches timed with HIP events, split bf16 products against the f32 matrix instruction.
usage: python tools/time_rowdec.py [rows0 rows1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piml_amd import _lib, ops  # noqa: E402


def main():
    rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096 * 6, 4096 * 10)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    brs = []
    for r in rows:
        emb = (torch.randn(r, 128, generator=g) * 0.7).to(dev).requires_grad_(True)
        ws = [(torch.randn(*shp, generator=g) * 0.15).to(dev).requires_grad_(True)
              for shp in ((64, 128), (64,), (64, 64), (64,), (2, 64), (2,))]
        brs.append(dict(emb=emb, decoder=ws[:4], predictor=ws[4:]))
    gp = [torch.randn(r, 2, generator=g).to(dev) for r in rows]
    gd = [torch.randn(r, 64, generator=g).to(dev) * 0.1 for r in rows]
    leaves = [t for br in brs for t in (br['emb'], *br['decoder'], *br['predictor'])]
    L = _lib.lib()
    for name, mode in (('f32 instruction', 0), ('split bf16', 1), ('f32 instruction', 0), ('split bf16', 1)):
        L.piml_rowdecoder_products(mode)
        for _ in range(5):
            outs = ops.fused_row_decoder(brs)
            loss = sum((o[0] * a).sum() + (o[1] * b).sum() for o, a, b in zip(outs, gp, gd))
            torch.autograd.grad(loss, leaves)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        n = 30
        for _ in range(n):
            e[0].record()
            outs = ops.fused_row_decoder(brs)
            e[1].record()
            loss = sum((o[0] * a).sum() + (o[1] * b).sum() for o, a, b in zip(outs, gp, gd))
            torch.cuda.synchronize()
            e[1].record()
            torch.autograd.grad(loss, leaves)
            e[2].record()
            torch.cuda.synchronize()
            tb += e[1].elapsed_time(e[2])
        # the forward alone, back to back
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(n):
            ops.fused_row_decoder(brs)
        e[1].record()
        torch.cuda.synchronize()
        tf = e[0].elapsed_time(e[1])
        print(f'{name:16s} rows {rows}: forward {tf / n * 1e3:7.1f} us   backward (dx + dW + slot sums, eager) {tb / n * 1e3:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
