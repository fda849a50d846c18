"""Probe: does a captured graph that contains memset nodes (torch's multi-block reductions zero their semaphores with
hipMemsetAsync) keep replaying correctly after OTHER graphs are captured / replayed in the same process?"""
import torch
dev = 'cuda:0'
x = torch.rand(1 << 20, device=dev)
want = int((x > 0.5).sum())
want_f = float(x.double().sum())

def capture(fn, pool=None):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        out = fn()
    return g, out

def fa():
    return (x > 0.5).sum(), x.sum(), torch.zeros(1000, device=dev) + 1

ga, outa = capture(fa)
def check(tag):
    ga.replay(); torch.cuda.synchronize()
    ok = int(outa[0]) == want and abs(float(outa[1]) - want_f) < 1.0 and float(outa[2].sum()) == 1000.0
    print(f'{tag}: int-sum {int(outa[0])} (want {want}), f32 sum {float(outa[1]):.1f} (want {want_f:.1f}), zeros+1 sum {float(outa[2].sum())} -> {"OK" if ok else "BROKEN"}')
check('A after capture')
y = torch.rand(1 << 20, device=dev)
gb, outb = capture(lambda: ((y * 2).sum(), (y > 0.1).sum()))
check('A after capturing B')
gb.replay(); torch.cuda.synchronize()
check('A after replaying B')
for _ in range(5):
    gb.replay()
torch.cuda.synchronize()
check('A after replaying B x5')
# many eager allocations / kernels in between
for _ in range(20):
    z = torch.rand(1 << 22, device=dev); (z > 0.3).sum().item()
check('A after eager work')
gc_, outc = capture(lambda: torch.zeros(1 << 20, device=dev).add_(1).sum())
check('A after capturing C (zeros)')
gc_.replay(); gb.replay(); torch.cuda.synchronize()
check('A after replaying C, B')
print('B check', float(outb[0]), float((y * 2).sum()), int(outb[1]), int((y > 0.1).sum()))
