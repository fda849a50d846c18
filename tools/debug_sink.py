import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, contextlib
from test_simulator_gpu import sim_args
from piml_amd import ops
from piml_amd.models.simulators import BaseSimulator
DEV = 'cuda:0'
torch.manual_seed(5)
sim = BaseSimulator(sim_args(model='pinnsf_m', dropout=0.0, learning_rate=1e-3))
net = sim.model
net.train(False)
print('messages_wanted', net.messages_wanted)
g = torch.Generator().manual_seed(11)
nf = int(os.environ.get('NF', '3'))
frames = [(torch.randn(4, 122, 6, 6, generator=g).to(DEV), torch.randn(4, 122, 10, 6, generator=g).to(DEV),
           torch.randn(4, 122, 7, generator=g).to(DEV)) for _ in range(nf)]
def run(use_sink):
    for p in net.parameters():
        p.grad = None
    sink = ops.ParamGradSink()
    with (sink.step() if use_sink else contextlib.nullcontext()):
        loss = 0
        for pf, of, sf in frames:
            out = net(pf, of, sf)
            loss = loss + out[0].square().sum()
        loss.backward()
    return [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
a, b, c, d = run(False), run(True), run(False), run(True)
for (name, _), w, gt, w2, gt2 in zip(net.named_parameters(), a, b, c, d):
    if w is None: continue
    print(f'{name:40s} sink-vs-autograd {float((w-gt).abs().max()):.3e}  autograd rerun {float((w-w2).abs().max()):.3e} sink rerun {float((gt-gt2).abs().max()):.3e}  max {float(w.abs().max()):.3e}')
