"""Tall-skinny GEMMs of the MLP (rows x 128 x 128): plain tuned mm / addmm vs a strided-batched formulation over
row chunks with the weight broadcast (stride 0) -- does the batched solution pool hold a faster kernel?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from piml_amd import _lib, tuning
print('committed selections loaded:', tuning.load())
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(30); torch.cuda.tunable.set_max_tuning_iterations(100)
torch.cuda.tunable.set_filename(os.path.join(ROOT, 'gpurun_out', 'tune_tall.csv'))
dev = 'cuda'
def timed(fn, reps=200):
    for _ in range(10): fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    t = _lib.StreamTimer(); t.start()
    for _ in range(reps // 20): g.replay()
    t.stop(); return t.elapsed_ms() * 1e3 / reps
for rows in (40960, 24576):
    X = torch.randn(rows, 128, device=dev); W = torch.randn(128, 128, device=dev); b = torch.randn(128, device=dev)
    f0 = lambda: torch.addmm(b, X, W.t())
    f1 = lambda: X.mm(W)
    f0(); f1(); torch.cuda.synchronize()
    torch.cuda.tunable.tuning_enable(False)
    print(f'rows {rows}: addmm(x, W^T) {timed(f0):.1f} us   mm(g, W) {timed(f1):.1f} us', flush=True)
    for B in (4, 8, 16, 32):
        Wt = W.t().unsqueeze(0).expand(B, 128, 128)
        Wn = W.unsqueeze(0).expand(B, 128, 128)
        torch.cuda.tunable.tuning_enable(True)
        g0 = lambda: torch.bmm(X.view(B, rows // B, 128), Wt)
        g1 = lambda: torch.bmm(X.view(B, rows // B, 128), Wn)
        g0(); g1(); torch.cuda.synchronize()
        torch.cuda.tunable.tuning_enable(False)
        err = float((g0().reshape(rows, 128) - X.mm(W.t())).abs().max())
        print(f'   B={B}: bmm(x, W^T) {timed(g0):.1f} us   bmm(g, W) {timed(g1):.1f} us   (max abs diff {err:.1e})', flush=True)
