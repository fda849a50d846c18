"""Weight-gradient GEMM dW = G^T X (K = rows, huge; M = N = 128): library split-K GEMM vs a batched
formulation (bmm over row chunks + a column sum of the per-chunk results)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import tuning, _lib, ops
print('tuned file accepted:', tuning.load())
dev = 'cuda'
def timed(fn, reps=200):
    for _ in range(10): fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    t = _lib.StreamTimer(); t.start()
    for _ in range(reps // 20): g.replay()
    t.stop(); return t.elapsed_ms() * 1e3 / reps
for rows, cin, cout in ((40960, 128, 128), (24576, 128, 128), (40960, 6, 128), (4096, 128, 64)):
    G = torch.randn(rows, cout, device=dev); X = torch.randn(rows, cin, device=dev)
    ref = G.t().mm(X)
    print(f'rows {rows} in {cin} out {cout}:  mm {timed(lambda: G.t().mm(X)):.1f} us', end='')
    for B in (16, 32, 64, 128, 256):
        if rows % B: continue
        def f():
            part = torch.bmm(G.view(B, rows // B, cout).transpose(1, 2), X.view(B, rows // B, cin))   # (B, cout, cin)
            return ops.act_bwd_colsum(part.view(B, cout * cin))[1].view(cout, cin) if (cout * cin) % 4 == 0 and cout * cin <= 1024 else part.sum(0)
        got = f()
        err = float((got - ref).abs().max() / ref.abs().max())
        print(f' | B={B}: {timed(f):.1f} us (rel err {err:.1e})', end='')
    print()
