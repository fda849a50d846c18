"""GPU: the corrector of `pinnsf_res` on the hand-written kernels (ops.fused_corrector, piml_amd/csrc/corrector.hip) against
the module-by-module expression of src/models/model.py:1016-1020, :1050-1052 (ResDNN -> attn_pooling -> MLP(128, [64, 2])) in
float64: outputs and every gradient, eval mode and train mode with an injected keep-mask (north-star bar 1e-5, relative to the
tensor's largest entry), ragged row counts, channelled input, other k; and the model with / without the fused corrector."""
import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _modules(seed=0):
    import piml_amd.models.model as MODEL
    torch.manual_seed(seed)
    res = MODEL.ResDNN(128, [[128] for _ in range(3)], nn.ReLU(), 0.5)
    pool = MODEL.attn_pooling(128)
    tail = MODEL.MLP(128, [64, 2])
    with torch.no_grad():       # scores of O(1): the exponential of an exponential is steep
        pool.get_weights.mlp[2].weight.mul_(0.3)
    return res.to(DEV), pool.to(DEV), tail.to(DEV)


def _reference(enc, scale, keep, pool, tail, g):
    """float64, written out: r = keep * scale * enc; attn = softmax(exp(w(r))); pooled = sum_k attn r; tail."""
    e = enc.detach().double().requires_grad_(True)
    ws = [p.detach().double().requires_grad_(True) for p in (*pool.parameters(), *tail.parameters())]
    r = e * scale if keep is None else e * scale * keep.double()
    hid = torch.relu(r @ ws[0].t() + ws[1])
    s = hid @ ws[2].t() + ws[3]
    attn = torch.softmax(torch.exp(s), dim=-2)
    pooled = (r * attn).sum(-2)
    out = torch.relu(pooled @ ws[4].t() + ws[5]) @ ws[6].t() + ws[7]
    grads = torch.autograd.grad(out, [e] + ws, g.double())
    return out, grads


@pytest.mark.parametrize('shape', [(4096, 6), (301, 6), (3, 50, 6), (77, 10), (1, 6)])
@pytest.mark.parametrize('train', [False, True])
def test_fused_corrector_matches_float64(shape, train):
    from piml_amd import ops
    res, pool, tail = _modules()
    g = torch.Generator().manual_seed(7)
    enc = (torch.randn(*shape, 128, generator=g) * 0.5).to(DEV).requires_grad_(True)
    gout = torch.randn(*shape[:-1], 2, generator=g).to(DEV)
    rows = enc.numel() // 128
    keep, bits, scale = None, None, 2.0
    if train:
        keep = (torch.rand(rows, 128, generator=g) < 0.5).to(DEV)
        bits = ops.pack_keep_bits(keep)
        keep = keep.view(*shape, 128)
        scale = 4.0
    gw, tl = pool.get_weights.mlp, tail.mlp
    params = [gw[0].weight, gw[0].bias, gw[2].weight, gw[2].bias, tl[0].weight, tl[0].bias, tl[2].weight, tl[2].bias]
    out = ops.fused_corrector(enc, scale, bits, params[:4], params[4:])
    grads = torch.autograd.grad(out, [enc] + params, gout)
    want, wgrads = _reference(enc, scale, keep, pool, tail, gout)
    torch.cuda.synchronize()

    def rel(a, b, scale=None):
        return float((a.detach().double() - b).abs().max() / (b.abs().max() if scale is None else scale).clamp_min(1e-30))
    assert out.shape == want.shape
    worst = {'out': rel(out, want)}
    for name, a, b in zip(['enc', 'wa', 'ba', 'wb', 'bb', 'wc', 'bc', 'wd', 'bd'], grads, wgrads):
        assert a.shape == b.shape, name
        # d/d(bb) = sum of the score gradients, which cancel (the softmax's gradients sum to zero per agent): its error is
        # measured against the size of the terms (d/d(wb) = sum of score gradient x hidden unit, same terms), not of the sum
        worst['g_' + name] = rel(a, b, scale=torch.maximum(wgrads[3].abs().max(), b.abs().max()) if name == 'bb' else None)
    print(shape, 'train' if train else 'eval', {k: f'{v:.1e}' for k, v in worst.items()})
    for name, v in worst.items():
        assert v <= 1e-5, (name, v)


def test_fused_corrector_is_bit_reproducible():
    from piml_amd import ops
    res, pool, tail = _modules()
    enc = torch.randn(4096, 6, 128, device=DEV).requires_grad_(True)
    gw, tl = pool.get_weights.mlp, tail.mlp
    params = [gw[0].weight, gw[0].bias, gw[2].weight, gw[2].bias, tl[0].weight, tl[0].bias, tl[2].weight, tl[2].bias]
    runs = []
    for _ in range(2):
        out = ops.fused_corrector(enc, 2.0, None, params[:4], params[4:])
        runs.append([out] + list(torch.autograd.grad(out, [enc] + params, torch.ones_like(out))))
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('train', [False, True])
def test_pinnsf_res_with_and_without_the_fused_corrector(train, monkeypatch):
    import piml_amd.models.model as MODEL
    from test_mlpglue_gpu import model_args
    from piml_amd import ops
    torch.manual_seed(11)
    m = MODEL.PINNSF_residual(model_args(res_hidden_layers=3)).to(DEV).train(train)
    N = 700
    pf, of, sf = torch.randn(N, 6, 6, device=DEV), torch.randn(N, 10, 6, device=DEV), torch.randn(N, 7, device=DEV)
    if train:       # the same masks on both paths
        for i, mod in enumerate((m.ped_processor, m.obs_processor, m.corrector[0])):
            rows = N * (10 if mod is m.obs_processor else 6)
            mod.keep_bits = ops.pack_keep_bits(torch.rand(rows, 128, generator=torch.Generator().manual_seed(i)) < 0.5).to(DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(MODEL, 'FUSED_CORRECTOR', fused)
        for p in m.parameters():
            p.grad = None
        acc = m(pf, of, sf)[0]
        acc.backward(torch.ones_like(acc))
        outs.append([acc.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None])
    assert len(outs[0]) == len(outs[1])
    for a, b in zip(*outs):
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) <= 2e-5
