"""Does torch._addmm_activation fuse ReLU into the hipBLASLt epilogue on this stack, and what does it cost?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import tuning, _lib
print('tuned file accepted:', tuning.load())
dev = 'cuda'
for rows, k, n in ((24576, 6, 128), (24576, 128, 128), (40960, 6, 128), (40960, 128, 128), (4096, 128, 64), (24576, 128, 64)):
    x = torch.randn(rows, k, device=dev); w = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev)
    ref = torch.relu(torch.addmm(b, x, w.t()))
    got = torch._addmm_activation(b, x, w.t(), use_gelu=False)
    print(rows, k, n, 'max diff', float((ref - got).abs().max()))
    def timed(fn, reps=50):
        for _ in range(5): fn()
        t = _lib.StreamTimer(); t.start()
        for _ in range(reps): fn()
        t.stop(); return t.elapsed_ms() * 1e3 / reps
    print('   addmm+relu_ %.1f us   _addmm_activation %.1f us   addmm only %.1f us' % (
        timed(lambda: torch.relu_(torch.addmm(b, x, w.t()))), timed(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)),
        timed(lambda: torch.addmm(b, x, w.t()))))
