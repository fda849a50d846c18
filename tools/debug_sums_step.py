"""Debug: the bench step on the sums path (--messages 0), eager twice and replayed, parameter by parameter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from piml_amd.scenes import synthetic_gc_scene

dev = torch.device('cuda:0')
scene = synthetic_gc_scene(4096, 2000, seed=0)
msg = int(os.environ.get('MSG', '0'))
st = bench.Step(scene, 4096, 4096, 0, 2000, dev, None, False, False, True, messages=bool(msg))
names = [n for n, _ in st.model.named_parameters()]


def grads():
    torch.cuda.synchronize()
    return [None if p.grad is None else p.grad.clone() for p in st.params] + [st.state_own.grad.clone()]


def cmp(a, b, tag):
    worst = 0.0
    for n, x, y in zip(names + ['state'], a, b):
        if (x is None) != (y is None):
            print(tag, n, 'None mismatch', x is None, y is None)
            continue
        if x is None:
            continue
        x, y = torch.nan_to_num(x), torch.nan_to_num(y)
        e = float((x - y).abs().max() / y.abs().max().clamp_min(1e-12))
        worst = max(worst, e)
        if e > 1e-4:
            print(tag, n, tuple(x.shape), 'rel', e, 'max a', float(x.abs().max()), 'max b', float(y.abs().max()))
    print(tag, 'worst', worst)


st.reset_grads(); st.step_body(); g1 = grads()
st.reset_grads(); st.step_body(); g2 = grads()
cmp(g2, g1, 'eager2 vs eager1')
ref = bench.Step(scene, 4096, 4096, 0, 2000, dev, None, False, False, True, messages=True)
ref.model.load_state_dict(st.model.state_dict())
ref.reset_grads(); ref.step_body()
torch.cuda.synchronize()
g_ref = [None if p.grad is None else p.grad.clone() for p in ref.params] + [ref.state_own.grad.clone()]
cmp(g1, g_ref, 'sums eager vs message eager')
st.capture()
print('mode', st.mode)
for _ in range(3):
    st.run()
g3 = grads()
cmp(g3, g1, 'replay vs eager1')
st.reset_grads(); st.step_body(); g4 = grads()
cmp(g4, g1, 'eager after replay vs eager1')
