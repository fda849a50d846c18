"""per-tensor fused vs torch.nn comparison for pinnsf_bm at 4096 agents, eval and train mode (development aid)"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import piml_amd.models.model as MODEL
from piml_amd import ops, _lib
from test_dropout_gpu import model_args, _passes
DEV = 'cuda:0'
name = sys.argv[1] if len(sys.argv) > 1 else 'PINNSF_bottleneck_multitask'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def run(label, ones=False, rowdec=True, x3=1, p=0.5):
    torch.manual_seed(0)
    net = getattr(MODEL, name)(model_args(dropout=p)).to(DEV).train(True)
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(n, 6, 6, generator=g).to(DEV), torch.randn(n, 10, 6, generator=g).to(DEV), torch.randn(n, 7, generator=g).to(DEV)]
    kp = torch.rand(n * 6, 128, generator=g) >= p
    ko = torch.rand(n * 10, 128, generator=g) >= p
    if ones:
        kp[:], ko[:] = True, True
    net.ped_processor.keep_bits = ops.pack_keep_bits(kp).to(DEV)
    net.obs_processor.keep_bits = ops.pack_keep_bits(ko).to(DEV)
    with torch.no_grad():
        probe = net(*base)
    weights = [torch.randn(o.shape, generator=g).to(DEV) * (1.0 if i == 0 else 1e-2) for i, o in enumerate(probe)]
    res = {}
    old = _lib.lib().piml_encoder_products(x3)
    for fused in (True, False):
        MODEL.FUSED_GLUE = fused
        MODEL.FUSED_ROW_DECODER = rowdec
        res[fused] = _passes(net, base, weights)
    MODEL.FUSED_GLUE = True
    MODEL.FUSED_ROW_DECODER = True
    _lib.lib().piml_encoder_products(old)
    names = [f'out{i}' for i in range(len(probe))] + ['g_ped', 'g_obs', 'g_self'] + [k for k, p in net.named_parameters() if p.grad is not None]
    print(label)
    for nm, a, b in zip(names, res[True], res[False]):
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
        if e > 1e-5:
            print(f'  {nm:50s} {tuple(a.shape)} {e:.2e}  max|b| {float(b.abs().max()):.3e}')


run('train p=0.5')
run('train, all-ones masks', ones=True)
run('train, library row decoder', rowdec=False)
run('train, f32 products', x3=0)
run('train p=0.1', p=0.1)
