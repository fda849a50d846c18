"""Tune (PyTorch TunableOp) the strided-batched GEMMs of the chunked weight-gradient formulation and time
them; writes gpurun_out/tune_bmm.csv (development aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from piml_amd import _lib
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(30)
torch.cuda.tunable.set_max_tuning_iterations(30)
out = os.path.join(ROOT, 'gpurun_out', 'tune_bmm.csv')
os.makedirs(os.path.dirname(out), exist_ok=True)
torch.cuda.tunable.set_filename(out)
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
shapes = [(40960, 128, 128), (24576, 128, 128), (4096, 128, 64), (4096, 64, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for rows, cin, cout in shapes:
    G = torch.randn(rows, cout, device=dev); X = torch.randn(rows, cin, device=dev)
    torch.cuda.tunable.tuning_enable(False)
    print(f'rows {rows} in {cin} out {cout}: plain mm {timed(lambda: G.t().mm(X)):.1f} us', flush=True)
    torch.cuda.tunable.tuning_enable(True)
    for B in (8, 16, 32, 64):
        f = lambda: torch.bmm(G.view(B, rows // B, cout).transpose(1, 2), X.view(B, rows // B, cin))
        f(); torch.cuda.synchronize()         # tunes here
        torch.cuda.tunable.tuning_enable(False)
        print(f'   B={B}: bmm alone {timed(f):.1f} us', flush=True)
        torch.cuda.tunable.tuning_enable(True)
print(open(out).read()[-2500:] if os.path.exists(out) else 'no file yet (written at exit)')
