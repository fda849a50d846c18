"""GPU: the processor's train-mode dropout INSIDE the fused kernels (reference: ResDNN.forward = Dropout_p(2 x),
src/models/model.py:82-119 with quirk Q3; model.train() at src/models/simulators.py:311; --dropout 0.5 at
src/main.py:45).  A CPU dropout stream cannot be reproduced on a GPU, so parity is checked with INJECTED masks: the
fused kernels against the plain torch.nn expression of the same network given the same keep-mask -- every output and
every gradient, bar 1e-5 of the tensor's largest magnitude -- and the mask generator bit-for-bit against its numpy
restatement (tests/philox_ref.py, pinned on Philox's known-answer vectors)."""
import types

import numpy as np
import pytest
import torch

import philox_ref

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def model_args(**kw):
    a = dict(ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128, processor_hidden_size=128,
             decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5,
             activation='relu', dataset_name='gc1560')
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize('rows,cols,p', [(100, 128, 0.5), (4097, 128, 0.3), (33, 100, 0.5), (33, 200, 0.25), (7, 128, 1.0),
                                         (7, 128, 0.0)])
def test_keep_bits_equal_the_philox_restatement(rows, cols, p):
    from piml_amd import ops
    st = ops.dropout_state(DEV, seed=1234)
    st[1] = 5                                               # draw counter
    torch.cuda.synchronize()
    for call in range(3):                                    # the launch advances the counter itself
        bits = ops.dropout_keep_bits(rows, cols, p, DEV, stream_id=call)
        want = philox_ref.keep_bits(1234, 5 + call, rows, cols, p, stream=call)
        assert np.array_equal(bits.cpu().numpy(), want), f'call {call}'
    assert int(st[1]) == 8 and int(st[2]) == 0
    ops.dropout_state(DEV, seed=int(torch.cuda.initial_seed()))


@pytest.mark.parametrize('n,p,products', [(4096, 0.5, 'x3'), (150, 0.5, 'x3'), (4096, 0.25, 'x3'), (150, 0.25, 'x3'),
                                          (4096, 0.5, 'f32'), (150, 0.5, 'f32')])
def test_masks_drawn_by_the_encoder_launch_equal_the_restatement(n, p, products):
    """('draw', p): the forward launch draws the masks itself -- for p = 0.5 on the split-product kernels INSIDE the forward
    kernel (one-wave and four-waves-per-tile forms), otherwise by one generator launch in front of it.  Either way the
    dropped features of the messages are exactly those of the numpy restatement (branch b = stream b of draw `offset`),
    the kept ones equal the eval-mode messages / (1 - p), and the draw counter advances by one per launch."""
    from piml_amd import _lib, ops
    old = _lib.lib().piml_encoder_products(1 if products == 'x3' else 0)
    try:
        g = torch.Generator().manual_seed(7)
        H = 128

        def branch(k, keep):
            x = torch.randn(n, k, 6, generator=torch.Generator().manual_seed(k)).to(DEV)
            w = [(torch.randn(*d, generator=torch.Generator().manual_seed(3 + i)) * (0.3 if len(d) == 2 else 0.1)).to(DEV)
                 for i, d in enumerate([(H, 6), (H,), (H, H), (H,), (H, H), (H,)])]
            return dict(x=x, scale=2.0 / (1 - p), weights=w, pooled=True, keep_bits=keep)
        st = ops.dropout_state(DEV, seed=4321)
        st[1] = 11
        with torch.no_grad():
            drawn = ops.fused_encoders([branch(6, ('draw', p)), branch(10, ('draw', p))])
            plain = ops.fused_encoders([branch(6, None), branch(10, None)])
        torch.cuda.synchronize()
        assert int(st[1]) == 12
        for b, k in enumerate((6, 10)):
            keep = torch.from_numpy(philox_ref.keep_mask(4321, 11, n * k, H, p, stream=b)).to(DEV).view(n, k, H)
            m, ref = drawn[b][0], plain[b][0]                   # plain carries scale 2 / (1 - p) too
            assert bool((m[~keep] == 0).all())
            assert torch.equal(m[keep], ref[keep])
            assert torch.allclose(drawn[b][1], (ref * keep).sum(-2), rtol=1e-5, atol=1e-5)
    finally:
        _lib.lib().piml_encoder_products(old)
        ops.dropout_state(DEV, seed=int(torch.cuda.initial_seed()))


def test_keep_fraction_and_graph_replays_draw_fresh_masks():
    from piml_amd import ops
    ops.dropout_state(DEV)
    out = torch.empty(3, 65536, 4, dtype=torch.int32, device=DEV)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        tmp = ops.dropout_keep_bits(65536, 128, 0.5, DEV)    # warm-up outside the capture
        with torch.cuda.graph(g, stream=s):
            tmp = ops.dropout_keep_bits(65536, 128, 0.5, DEV)
    for i in range(3):
        g.replay()
        out[i].copy_(tmp)
    torch.cuda.synchronize()
    keep = [ops.unpack_keep_bits(out[i], 128) for i in range(3)]
    for k in keep:
        assert abs(float(k.float().mean()) - 0.5) < 1e-3
    assert float((keep[0] != keep[1]).float().mean()) > 0.45 and float((keep[1] != keep[2]).float().mean()) > 0.45


def avoid_relu_kinks(net, base, rel=1e-5, rounds=12):
    """Re-draw the inputs of every agent that has a ReLU pre-activation within `rel` of zero (relative to the layer's mean
    magnitude) anywhere in the network.  Two correct float32 evaluations of the same network round such a value to
    different sides of zero, and the comparison would then measure one flipped ReLU (a whole gradient row appears or
    vanishes), not the arithmetic.  Evaluated on the plain torch.nn expression; the injected masks belong to the row
    index, not to the data, so they stay valid."""
    import piml_amd.models.model as MODEL
    import torch.nn as nn
    n_lead = base[2].shape[:-1]
    flagged = []

    def hook(_m, _inp, out):
        near = out.detach().abs() < rel * out.detach().abs().mean()
        near = near.any(-1)
        while near.dim() > len(n_lead):
            near = near.any(-1)
        flagged.append(near)
    hooks = []
    for mod in net.modules():
        if isinstance(mod, MODEL.MLP):
            layers = list(mod.mlp)
            for lin, act in zip(layers[0::2], layers[1::2]):
                if isinstance(act, nn.ReLU):
                    hooks.append(lin.register_forward_hook(hook))
    g = torch.Generator().manual_seed(99)
    old = MODEL.FUSED_GLUE
    MODEL.FUSED_GLUE = False
    try:
        for _ in range(rounds):
            flagged.clear()
            with torch.no_grad():
                net(*base)
            bad = torch.stack([f.reshape(n_lead) for f in flagged]).any(0)
            nbad = int(bad.sum())
            if nbad == 0:
                break
            for t in base:
                t[bad] = torch.randn(nbad, *t.shape[len(n_lead):], generator=g).to(t.device)
        else:
            raise AssertionError('avoid_relu_kinks: still near a kink after re-drawing')
    finally:
        MODEL.FUSED_GLUE = old
        for h in hooks:
            h.remove()


def _passes(net, base, weights):
    ins = [t.clone().requires_grad_(True) for t in base]
    net.zero_grad(set_to_none=True)
    out = net(*ins)
    loss = sum((o * w).sum() for o, w in zip(out, weights))
    loss.backward()
    return [o.detach() for o in out] + [t.grad for t in ins] + [p.grad for p in net.parameters() if p.grad is not None]


@pytest.mark.parametrize('name,n,p', [('PINNSF_multitask', 4096, 0.5), ('PINNSF_multitask', 122, 0.5), ('PINNSF', 700, 0.3),
                                      ('PINNSF_bottleneck_multitask', 4096, 0.5), ('PINNSF_bottleneck_multitask', 122, 0.5),
                                      ('PINNSF_bottleneck', 1000, 0.8), ('PINNSF_multitask', (3, 250), 0.5)])
def test_train_mode_fused_kernels_match_torch_nn_with_the_same_mask(name, n, p):
    """model.train() with dropout p: fused kernels (encoder forward / dX / dW incl. the few-rows forms, decoder tails, row
    decoders) vs the plain torch.nn expression (PIML_FUSED_GLUE off), same injected keep-masks."""
    import piml_amd.models.model as MODEL
    from piml_amd import ops
    shape = n if isinstance(n, tuple) else (n,)
    agents = int(np.prod(shape))
    torch.manual_seed(0)
    net = getattr(MODEL, name)(model_args(dropout=p)).to(DEV).train()
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(*shape, 6, 6, generator=g).to(DEV), torch.randn(*shape, 10, 6, generator=g).to(DEV),
            torch.randn(*shape, 7, generator=g).to(DEV)]
    kp = torch.rand(agents * 6, 128, generator=g) >= p
    ko = torch.rand(agents * 10, 128, generator=g) >= p
    net.ped_processor.keep_bits = ops.pack_keep_bits(kp).to(DEV)
    net.obs_processor.keep_bits = ops.pack_keep_bits(ko).to(DEV)
    avoid_relu_kinks(net, base)
    base[0][..., : max(shape[-1] // 7, 1), 3:, :] = 0.0      # zero-padded neighbour rows (quirk Q4)
    with torch.no_grad():
        probe = net(*base)
    weights = [torch.randn(o.shape, generator=g).to(DEV) * (1.0 if i == 0 else 1e-2) for i, o in enumerate(probe)]
    # the fused kernels really ran with the mask: dropped features of the messages are exact zeros
    msgs = probe[1].reshape(-1, probe[1].shape[-1])
    if msgs.shape[-1] == 128:
        assert bool((msgs[~kp.to(DEV)] == 0).all()) and float((msgs[kp.to(DEV)] != 0).float().mean()) > 0.99
    res = {}
    try:
        for fused in (True, False):
            MODEL.FUSED_GLUE = fused
            res[fused] = _passes(net, base, weights)
    finally:
        MODEL.FUSED_GLUE = True
    assert len(res[True]) == len(res[False])
    worst = 0.0
    for a, b in zip(res[True], res[False]):
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
    print(f'{name} n={n} p={p}: train-mode fused kernels vs torch.nn with the same mask, max rel err {worst:.1e}')
    assert worst <= 1e-5


@pytest.mark.parametrize('train', [False, True])
def test_pinnsf_res_encoders_on_the_fused_kernels(train):
    """`--model pinnsf_res` (src/models/model.py:973-1059): its corrector reads the pedestrian encoder's raw output, so that
    encoder runs on the fused kernels with scale 1 + the mask-aware processor pass, the obstacle branch on the
    standard fused path (one launch for both when no mask is involved); against the plain torch.nn expression, eval mode and
    train mode with injected masks."""
    import piml_amd.models.model as MODEL
    from piml_amd import ops
    torch.manual_seed(0)
    net = MODEL.PINNSF_residual(model_args(res_hidden_layers=3, dropout=0.5)).to(DEV).train(train)
    g = torch.Generator().manual_seed(11)
    n = 700
    base = [torch.randn(n, 6, 6, generator=g).to(DEV), torch.randn(n, 10, 6, generator=g).to(DEV), torch.randn(n, 7, generator=g).to(DEV)]
    for proc, rows in ((net.ped_processor, n * 6), (net.obs_processor, n * 10), (net.corrector[0], n * 6)):
        proc.keep_bits = ops.pack_keep_bits(torch.rand(rows, 128, generator=g) >= 0.5).to(DEV)
    avoid_relu_kinks(net, base)
    with torch.no_grad():
        probe = net(*base)
    weights = [torch.randn(o.shape, generator=g).to(DEV) * (1.0 if i == 0 else 1e-2) for i, o in enumerate(probe)]
    launches = []
    real = ops.fused_encoders
    ops.fused_encoders = lambda brs: (launches.append(len(brs)), real(brs))[1]
    res = {}
    try:
        for fused in (True, False):
            MODEL.FUSED_GLUE = fused
            res[fused] = _passes(net, base, weights)
    finally:
        MODEL.FUSED_GLUE = True
        ops.fused_encoders = real
    # both encoders took the fused kernels, in ONE launch (round 4: the many-rows kernels and the one-pass backward instead of
    # two few-rows launches; in train mode the pedestrian branch rides along under an all-ones mask)
    assert launches == [2]
    worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(res[True], res[False]))
    print(f'pinnsf_res train={train}: fused encoders vs torch.nn, max rel err {worst:.1e}')
    assert worst <= 1e-5


@pytest.mark.parametrize('products', ['x3', 'f32'])
def test_train_mode_f32_instruction_kernels_match_too(products):
    """Both product forms of the encoder kernels carry the mask (piml_encoder_products)."""
    import piml_amd.models.model as MODEL
    from piml_amd import _lib, ops
    old = _lib.lib().piml_encoder_products(1 if products == 'x3' else 0)
    try:
        torch.manual_seed(0)
        net = MODEL.PINNSF_multitask(model_args()).to(DEV).train()
        g = torch.Generator().manual_seed(2)
        n = 2048
        base = [torch.randn(n, 6, 6, generator=g).to(DEV), torch.randn(n, 10, 6, generator=g).to(DEV), torch.randn(n, 7, generator=g).to(DEV)]
        net.ped_processor.keep_bits = ops.pack_keep_bits(torch.rand(n * 6, 128, generator=g) >= 0.5).to(DEV)
        net.obs_processor.keep_bits = ops.pack_keep_bits(torch.rand(n * 10, 128, generator=g) >= 0.5).to(DEV)
        avoid_relu_kinks(net, base)
        with torch.no_grad():
            probe = net(*base)
        weights = [torch.randn(o.shape, generator=g).to(DEV) * (1.0 if i == 0 else 1e-2) for i, o in enumerate(probe)]
        res = {}
        try:
            for fused in (True, False):
                MODEL.FUSED_GLUE = fused
                res[fused] = _passes(net, base, weights)
        finally:
            MODEL.FUSED_GLUE = True
        worst = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(res[True], res[False]))
        print(f'{products}: max rel err {worst:.1e}')
        assert worst <= 1e-5
    finally:
        _lib.lib().piml_encoder_products(old)


@pytest.mark.parametrize('n', [300, 1500, 2100])
def test_train_mode_draws_its_own_masks(n):
    """Without an injected mask every forward pass draws one: outputs differ from pass to pass, the kept fraction is
    1 - p, the kept features are scale / (1 - p) times the eval-mode output, and backward uses the SAME mask.
    The backward here arrives through the pedestrian messages ALONE, i.e. as an encoder launch of one branch: at 300 / 2100 agents
    that branch is on the other side of a kernel-choice bound than the two branches together were in the forward (until round 5
    the second case raised 'the forward did not store the layer-1 activations')."""
    import piml_amd.models.model as MODEL
    torch.manual_seed(0)
    net = MODEL.PINNSF_multitask(model_args(dropout=0.25)).to(DEV)
    g = torch.Generator().manual_seed(3)
    base = [torch.randn(n, 6, 6, generator=g).to(DEV), torch.randn(n, 10, 6, generator=g).to(DEV), torch.randn(n, 7, generator=g).to(DEV)]
    with torch.no_grad():
        ev = net.eval()(*base)
        net.train()
        a, b = net(*base), net(*base)
    assert not torch.equal(a[1], b[1]) and not torch.equal(a[0], b[0])
    kept = a[1] != 0
    assert abs(float(kept.float().mean()) - 0.75) < 5e-3
    assert torch.allclose(a[1][kept], (ev[1] / 0.75)[kept], rtol=1e-6, atol=1e-7)
    x = base[0].clone().requires_grad_(True)
    out = net(x, base[1], base[2])
    out[1].sum().backward()                                  # d(sum msgs)/dx flows only through kept features
    w3 = net.ped_encoder.mlp[4].weight
    keep = (out[1] != 0).float().detach()
    with torch.enable_grad():
        x2 = base[0].clone().requires_grad_(True)
        h = net.ped_encoder.mlp[:4](x2)
        ((h @ w3.t() + net.ped_encoder.mlp[4].bias) * keep * (2 / 0.75)).sum().backward()
    assert float((x.grad - x2.grad).abs().max() / x2.grad.abs().max()) < 1e-5


def test_scale_ksum_with_keep_bits():
    from piml_amd import ops
    g = torch.Generator().manual_seed(4)
    for cols in (128, 64, 100):
        e = torch.randn(50, 6, cols, generator=g).to(DEV).requires_grad_(True)
        keep = (torch.rand(300, cols, generator=g) >= 0.5).to(DEV)
        bias = torch.randn(cols, generator=g).to(DEV)
        m, pooled = ops.scale_ksum(e, 4.0, bias=bias, keep_bits=ops.pack_keep_bits(keep))
        gm, gp = torch.randn(m.shape, generator=g).to(DEV), torch.randn(pooled.shape, generator=g).to(DEV)
        ((m * gm).sum() + (pooled * gp).sum()).backward()
        e2 = e.detach().clone().requires_grad_(True)
        m2 = 4.0 * (e2 + bias) * keep.view(50, 6, cols)
        ((m2 * gm).sum() + (m2.sum(-2) * gp).sum()).backward()
        assert torch.allclose(m, m2, rtol=1e-6, atol=1e-6) and torch.allclose(pooled, m2.sum(-2), rtol=1e-5, atol=1e-5)
        assert torch.allclose(e.grad, e2.grad, rtol=1e-6, atol=1e-6)
