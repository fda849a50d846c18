"""GPU: the fused encoder kernels (piml_amd/csrc/encoder_x3.hip: split bf16 products, the default; encoder.hip: the f32
matrix instruction; ops.fused_encoders) against a float64
restatement of the reference's arithmetic -- src/models/model.py:40-65 (Linear / ReLU chain), :82-119 (processor =
2 x, quirk Q3), :1279-1283 (sum over the k neighbours) -- for outputs and every gradient.  Tolerance: 1e-5 relative
to the tensor's largest magnitude (north-star bar); the measured error is printed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
H = 128


def make_branch(n, k, in_dim, seed, scale=2.0, x_grad=True, lead=()):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(*lead, n, k, in_dim, generator=g) * 2).to(DEV)
    x[..., : max(n // 7, 1), k // 2:, :] = 0.0            # zero-padded neighbour rows (quirk Q4)
    dims = [(H, in_dim), (H,), (H, H), (H,), (H, H), (H,)]
    w = [(torch.randn(*d, generator=g) * (0.3 if len(d) == 2 else 0.1)).to(DEV).requires_grad_(True) for d in dims]
    return dict(x=x.requires_grad_(x_grad), scale=scale, weights=w, pooled=True)


def reference(br, g_msgs, g_pooled):
    """float64 autograd of the same network."""
    x = br['x'].detach().double().requires_grad_(True)
    w = [t.detach().double().requires_grad_(True) for t in br['weights']]
    h = torch.relu(x @ w[0].t() + w[1])
    h = torch.relu(h @ w[2].t() + w[3])
    msgs = br['scale'] * (h @ w[4].t() + w[5])
    pooled = msgs.sum(-2)
    loss = 0
    if g_msgs is not None:
        loss = loss + (msgs * g_msgs.double()).sum()
    if g_pooled is not None:
        loss = loss + (pooled * g_pooled.double()).sum()
    loss.backward()
    return msgs.detach(), pooled.detach(), x.grad, [t.grad for t in w]


def dodge_relu_kinks(br, rel=1e-5, rounds=12):
    """Re-draw the input rows that put a ReLU pre-activation within `rel` of zero (relative to the layer's mean magnitude):
    two correct float32 evaluations round such a value to different sides of zero, and a comparison at 1e-5 would then
    measure one flipped ReLU -- a whole gradient row appearing or vanishing -- instead of the arithmetic (at 5 M
    pre-activations a float32 ulp around zero is hit about every other draw)."""
    g = torch.Generator().manual_seed(1234)
    w = [t.detach().double() for t in br['weights']]
    with torch.no_grad():
        for _ in range(rounds):
            x = br['x'].detach().double()
            z1 = x @ w[0].t() + w[1]
            z2 = torch.relu(z1) @ w[2].t() + w[3]
            bad = ((z1.abs() < rel * z1.abs().mean()) | (z2.abs() < rel * z2.abs().mean())).any(-1)
            n = int(bad.sum())
            if n == 0:
                return br
            br['x'][bad] = (torch.randn(n, br['x'].shape[-1], generator=g) * 2).to(DEV)
    raise AssertionError('dodge_relu_kinks: still near a kink after re-drawing')


def relerr(got, want):
    want = want.to(got.device)
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shapes,upstream', [
    ([(300, 6, 6)], 'both'), ([(37, 10, 6)], 'pooled'), ([(64, 6, 4)], 'msgs'), ([(1, 1, 1)], 'both'),
    ([(4096, 6, 6), (4096, 10, 6)], 'pooled'), ([(700, 6, 6), (300, 10, 6)], 'both'), ([(33, 3, 8), (2000, 7, 5)], 'both'),
])
def test_fused_encoders_match_float64_reference(shapes, upstream):
    from piml_amd import ops
    branches = [make_branch(n, k, d, seed=10 * i + n % 7) for i, (n, k, d) in enumerate(shapes)]
    outs = ops.fused_encoders(branches)
    gen = torch.Generator().manual_seed(5)
    ups, loss = [], 0
    for (msgs, pooled) in outs:
        gm = torch.randn(msgs.shape, generator=gen).to(DEV) if upstream in ('both', 'msgs') else None
        gp = torch.randn(pooled.shape, generator=gen).to(DEV) if upstream in ('both', 'pooled') else None
        ups.append((gm, gp))
        if gm is not None:
            loss = loss + (msgs * gm).sum()
        if gp is not None:
            loss = loss + (pooled * gp).sum()
    loss.backward()
    worst = {}
    for br, (msgs, pooled), (gm, gp) in zip(branches, outs, ups):
        rm, rp, rgx, rgw = reference(br, gm, gp)
        worst['msgs'] = max(worst.get('msgs', 0), relerr(msgs.detach(), rm))
        worst['pooled'] = max(worst.get('pooled', 0), relerr(pooled.detach(), rp))
        worst['g_x'] = max(worst.get('g_x', 0), relerr(br['x'].grad, rgx))
        for name, t, r in zip(('dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3'), br['weights'], rgw):
            worst[name] = max(worst.get(name, 0), relerr(t.grad, r))
    print(f'fused encoder {shapes} upstream={upstream}: max rel err vs float64 ' +
          ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    assert max(worst.values()) <= 1e-5, worst


def test_fused_encoders_leading_dims_no_input_grad_and_inference():
    """(C, N, k, 6) channelled input; inputs without gradient (pointwise training: features are data); no_grad."""
    from piml_amd import ops
    br = make_branch(50, 6, 6, seed=3, x_grad=False, lead=(4,))
    (msgs, pooled), = ops.fused_encoders([br])
    assert msgs.shape == (4, 50, 6, H) and pooled.shape == (4, 50, H)
    pooled.square().sum().backward()
    assert br['x'].grad is None and all(t.grad is not None for t in br['weights'])
    rm, rp, _, rgw = reference(br, None, 2 * pooled.detach())
    assert relerr(msgs.detach(), rm) <= 1e-5 and relerr(br['weights'][2].grad, rgw[2]) <= 1e-5
    with torch.no_grad():
        (m2, p2), = ops.fused_encoders([br])
    assert torch.equal(m2, msgs.detach()) and torch.equal(p2, pooled.detach())
    br2 = dict(br, pooled=False)
    (m3, p3), = ops.fused_encoders([br2])
    assert p3 is None and torch.equal(m3.detach(), msgs.detach())


def test_fused_encoders_are_deterministic():
    """No atomics anywhere: two runs give bit-identical outputs and gradients."""
    from piml_amd import ops
    res = []
    for _ in range(2):
        branches = [make_branch(1000, 6, 6, seed=1), make_branch(1000, 10, 6, seed=2)]
        outs = ops.fused_encoders(branches)
        sum((p * p).sum() + m.sum() for m, p in outs).backward()
        res.append([o.detach() for pair in outs for o in pair] + [b['x'].grad for b in branches] +
                   [t.grad for b in branches for t in b['weights']])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize('name', ['PINNSF_multitask', 'PINNSF_bottleneck_multitask', 'PINNSF'])
def test_models_with_fused_encoder_match_library_chain(name):
    """Whole networks: fused encoder kernels vs the library-GEMM chain (PIML_FUSED_ENCODER off), outputs and all
    gradients, at a size where both branches take the fused path."""
    import types
    import piml_amd.models.model as MODEL
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16,
        decoder_hidden_layers=2, dropout=0.5, activation='relu', dataset_name='gc1560')
    torch.manual_seed(0)
    net = getattr(MODEL, name)(args).to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    n = 600
    base = [torch.randn(n, 6, 6, generator=g).to(DEV), torch.randn(n, 10, 6, generator=g).to(DEV),
            torch.randn(n, 7, generator=g).to(DEV)]
    res = {}
    for fused in (True, False):
        MODEL.FUSED_ENCODER = fused
        ins = [t.clone().requires_grad_(True) for t in base]
        net.zero_grad(set_to_none=True)
        out = net(*ins)
        (out[0].square().sum() + out[1].sum() * 1e-2 + out[-1].sum()).backward()
        res[fused] = [o.detach() for o in out] + [t.grad for t in ins] + \
            [p.grad for p in net.parameters() if p.grad is not None]
    MODEL.FUSED_ENCODER = True
    worst = 0.0
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
    print(f'{name}: fused encoder vs library chain, max rel err {worst:.1e}')
    assert worst <= 2e-5


@pytest.mark.parametrize('name,shape', [('PINNSF_multitask', (600,)), ('PINNSF', (600,)), ('PINNSF_multitask', (3, 250)),
                                        ('PINNSF_multitask', (33,)), ('PINNSF_multitask', (4096,))])
def test_fused_network_matches_plain_torch(name, shape):
    """ops.fused_pinnsf (encoders + decoder tail + desired force as one node) vs the plain torch.nn expression of the
    same network (PIML_FUSED_GLUE off): every output and every gradient, incl. channelled (C, N, .) input with the
    reference's dim=1 agent norm (quirk Q2) and a ragged last tile."""
    import types
    import piml_amd.models.model as MODEL
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16,
        decoder_hidden_layers=2, dropout=0.5, activation='relu', dataset_name='gc1560')
    torch.manual_seed(0)
    net = getattr(MODEL, name)(args).to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(*shape, 6, 6, generator=g).to(DEV), torch.randn(*shape, 10, 6, generator=g).to(DEV),
            torch.randn(*shape, 7, generator=g).to(DEV)]
    base[2][..., 0, :2] = 0.0          # |dest| == 0 corner of the desired-force term
    res = {}
    old_rows = MODEL.FUSED_ENCODER_MIN_ROWS
    MODEL.FUSED_ENCODER_MIN_ROWS = 1
    try:
        for fused in (True, False):
            MODEL.FUSED_GLUE = fused
            ins = [t.clone().requires_grad_(True) for t in base]
            net.zero_grad(set_to_none=True)
            out = net(*ins)
            (out[0].square().sum() + out[1].sum() * 1e-2 + out[2].square().sum() * 1e-3 + out[-1].sum()).backward()
            res[fused] = [o.detach() for o in out] + [t.grad for t in ins] + \
                [p.grad for p in net.parameters() if p.grad is not None]
    finally:
        MODEL.FUSED_GLUE = True
        MODEL.FUSED_ENCODER_MIN_ROWS = old_rows
    worst = 0.0
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
    print(f'{name} {shape}: fused network vs plain torch, max rel err {worst:.1e}')
    assert worst <= 2e-5


def _multitask_net(seed=0):
    import types
    import piml_amd.models.model as MODEL
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16,
        decoder_hidden_layers=2, dropout=0.5, activation='relu', dataset_name='gc1560')
    torch.manual_seed(seed)
    return MODEL.PINNSF_multitask(args).to(DEV).eval()


def _net_pass(net, base):
    ins = [t.clone().requires_grad_(True) for t in base]
    net.zero_grad(set_to_none=True)
    out = net(*ins)
    (out[0].square().sum() + out[1].sum() * 1e-2 + out[2].square().sum() * 1e-3).backward()
    return [o.detach().clone() for o in out] + [t.grad.clone() for t in ins] + \
        [p.grad.clone() for p in net.parameters() if p.grad is not None]


def test_network_streams_and_prepack_are_bitwise_neutral():
    """The network call in program order (default), its forked form (PIML_FORK: side streams inside piml_pinnsf_fwd /
    bwd) and the prepacked form (`packed_weights()`: one pack per block, skipped in the forward passes) run the same
    kernels on the same data: every output and gradient is bitwise equal.  A weight update between two `packed_weights()` blocks
    must be seen by the second one."""
    from piml_amd import ops
    net = _multitask_net()
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(700, 6, 6, generator=g).to(DEV), torch.randn(700, 10, 6, generator=g).to(DEV),
            torch.randn(700, 7, generator=g).to(DEV)]
    ref = _net_pass(net, base)
    assert len(ref) > 20                                     # head output + every weight gradient present
    old = ops.FORK_NETWORK
    try:
        ops.FORK_NETWORK = True
        serial = _net_pass(net, base)
    finally:
        ops.FORK_NETWORK = old
    with net.packed_weights():
        packed = _net_pass(net, base)
        packed2 = _net_pass(net, base)                       # second pass inside the block: no pack at all
    for name, other in (('forked', serial), ('packed', packed), ('packed, 2nd pass', packed2)):
        for a, b in zip(ref, other):
            assert torch.equal(a, b), name
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)
    ref2 = _net_pass(net, base)
    with net.packed_weights():
        packed3 = _net_pass(net, base)
    assert not torch.equal(ref2[0], ref[0])
    for a, b in zip(ref2, packed3):
        assert torch.equal(a, b)
    # an in-place weight change INSIDE a block (optimizer step, load_state_dict) leaves stale operand images: refused
    with net.packed_weights():
        with torch.no_grad():
            net.ped_encoder.mlp[2].weight.mul_(1.5)
        with pytest.raises(ValueError, match='modified in place'):
            net(*base)
    torch.cuda.synchronize()


def test_prepacked_network_in_a_captured_graph():
    """`packed_weights()` inside a captured step: the pack is part of the graph, re-run by every replay (so weight
    updates between replays are seen), and the replay equals the eager pass bitwise."""
    net = _multitask_net(3)
    g = torch.Generator().manual_seed(6)
    base = [torch.randn(512, 6, 6, generator=g).to(DEV), torch.randn(512, 10, 6, generator=g).to(DEV),
            torch.randn(512, 7, generator=g).to(DEV)]
    params = [p for p in net.parameters()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            with net.packed_weights():
                _net_pass(net, base)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    net.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), net.packed_weights():
        out = net(*base)
        (out[0].square().sum() + out[1].sum() * 1e-2).backward()
    for scale in (1.0, 0.97):
        with torch.no_grad():
            for p in params:
                p.mul_(scale)
        graph.replay()
        torch.cuda.synchronize()
        got = [o.detach().clone() for o in out] + [p.grad.clone() for p in params if p.grad is not None]
        saved = [p.grad for p in params]
        for p in params:
            p.grad = None
        eager_out = net(*base)
        (eager_out[0].square().sum() + eager_out[1].sum() * 1e-2).backward()
        want = [o.detach() for o in eager_out] + [p.grad for p in params if p.grad is not None]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        for p, gsave in zip(params, saved):
            p.grad = gsave


@pytest.fixture(params=['x3', 'f32'])
def rowdec_products(request):
    """the many-rows kernels of the row decoder on split bf16 products (default) and on the f32 matrix instruction"""
    from piml_amd import _lib
    old = _lib.lib().piml_rowdecoder_products(1 if request.param == 'x3' else 0)
    yield request.param
    _lib.lib().piml_rowdecoder_products(old)


@pytest.mark.parametrize('rows', [(24576, 40960), (700, 33), (64, 2048), (65536 + 17, 40960 - 5)])
def test_fused_row_decoder_matches_float64(rows, rowdec_products):
    """ops.fused_row_decoder (decoder + predictor of the bottleneck variants per neighbour row: the decoder kernels with
    the rows in the role of the agents, two branches of different sizes in one launch) against the float64 expression:
    both outputs, and the gradients for upstream gradients on the predictions AND on the decoder output (the `decoded`
    collision head of pinnsf_bm), incl. ragged last tiles and multi-chunk dW slabs."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(11)
    brs, refs = [], []
    for r in rows:
        emb = (torch.randn(r, 128, generator=g) * 0.7).to(DEV)
        ws = [(torch.randn(*shp, generator=g) * 0.15).to(DEV).requires_grad_(True)
              for shp in ((64, 128), (64,), (64, 64), (64,), (2, 64), (2,))]
        # a hidden unit within rounding of zero is on either side of the ReLU's step depending on the summation order (at 10^5 rows
        # x 64 units one such unit turns up): those rows get the zero embedding, whose pre-activation is b1
        z = emb.double() @ ws[0].detach().double().t() + ws[1].detach().double()
        emb[(z.abs() < 1e-5).any(1)] = 0.0
        assert float(ws[1].detach().abs().min()) > 1e-5
        emb.requires_grad_(True)
        brs.append(dict(emb=emb, decoder=ws[:4], predictor=ws[4:]))
    outs = ops.fused_row_decoder(brs)
    gp = [torch.randn(r, 2, generator=g).to(DEV) for r in rows]
    gd = [torch.randn(r, 64, generator=g).to(DEV) * 0.1 for r in rows]
    loss = sum((o[0] * a).sum() + (o[1] * b).sum() for o, a, b in zip(outs, gp, gd))
    leaves = [t for br in brs for t in (br['emb'], *br['decoder'], *br['predictor'])]
    grads = torch.autograd.grad(loss, leaves)
    # float64 reference
    ref_out, ref_leaves = [], []
    for br in brs:
        e, (w1, b1, w2, b2), (wp, bp) = [t.detach().double().requires_grad_(True) for t in (br['emb'],)][0], \
            [t.detach().double().requires_grad_(True) for t in br['decoder']], \
            [t.detach().double().requires_grad_(True) for t in br['predictor']]
        d = torch.relu(e @ w1.t() + b1) @ w2.t() + b2
        ref_out.append((d @ wp.t() + bp, d))
        ref_leaves += [e, w1, b1, w2, b2, wp, bp]
    ref_loss = sum((o[0] * a.double()).sum() + (o[1] * b.double()).sum() for o, a, b in zip(ref_out, gp, gd))
    ref_grads = torch.autograd.grad(ref_loss, ref_leaves)
    worst = 0.0
    for o, ro in zip(outs, ref_out):
        for a, b in zip(o, ro):
            worst = max(worst, float((a.double() - b).abs().max() / b.abs().max()))
    for a, b in zip(grads, ref_grads):
        worst = max(worst, float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)))
    print(f'fused_row_decoder rows {rows} ({rowdec_products}): max rel err vs float64 {worst:.1e}')
    assert worst <= 1e-5
    # deterministic
    outs2 = ops.fused_row_decoder(brs)
    grads2 = torch.autograd.grad(sum((o[0] * a).sum() + (o[1] * b).sum() for o, a, b in zip(outs2, gp, gd)), leaves)
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b)


@pytest.mark.parametrize('name', ['PINNSF_bottleneck_multitask', 'PINNSF_bottleneck'])
def test_bottleneck_models_with_fused_row_decoder_match_library_decoders(name):
    """The bottleneck models on the fused row decoder vs the same models with their decoders on library GEMMs
    (PIML_FUSED_ROW_DECODER=0): every output and gradient."""
    import types
    import piml_amd.models.model as MODEL
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16,
        decoder_hidden_layers=2, dropout=0.5, activation='relu', dataset_name='gc1560')
    torch.manual_seed(0)
    net = getattr(MODEL, name)(args).to(DEV).eval()
    g = torch.Generator().manual_seed(2)
    base = [torch.randn(600, 6, 6, generator=g).to(DEV), torch.randn(600, 10, 6, generator=g).to(DEV),
            torch.randn(600, 7, generator=g).to(DEV)]
    res = {}
    try:
        for fused in (True, False):
            MODEL.FUSED_ROW_DECODER = fused
            ins = [t.clone().requires_grad_(True) for t in base]
            net.zero_grad(set_to_none=True)
            out = net(*ins)
            (out[0].square().sum() + out[1].sum() * 1e-2 + out[2].square().sum() * 1e-3 + out[-1].sum()).backward()
            res[fused] = [o.detach() for o in out] + [t.grad for t in ins] + \
                [p.grad for p in net.parameters() if p.grad is not None]
    finally:
        MODEL.FUSED_ROW_DECODER = True
    assert len(res[True]) == len(res[False])
    worst = 0.0
    for a, b in zip(res[True], res[False]):
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
    print(f'{name}: fused row decoder vs library decoders, max rel err {worst:.1e}')
    assert worst <= 2e-5


@pytest.mark.parametrize('nbr,agents,ks,with_head', [(1, 70, (13,), False), (1, 4096, (6,), True), (2, 45, (3, 11), True),
                                                     (2, 1, (1, 1), False)])
def test_fused_pinnsf_edge_geometries_match_float64(nbr, agents, ks, with_head):
    """ops.fused_pinnsf off the beaten path: ONE branch (no obstacle branch: plain store instead of the two-branch atomic
    combine), more neighbours than the pooling keeps in flight at once (k = 13 > 10), k = 1, a single agent, with and
    without the collision head -- outputs and every gradient against float64."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(21)
    tau = 0.5

    def mk(*shape, scale=0.2):
        return (torch.randn(*shape, generator=g) * scale).to(DEV).requires_grad_(True)
    brs = []
    for b in range(nbr):
        x = mk(agents, ks[b], 6, scale=1.0)
        brs.append(dict(x=x, scale=2.0, encoder=[mk(128, 6), mk(128), mk(128, 128), mk(128), mk(128, 128), mk(128)],
                        decoder=[mk(64, 128), mk(64), mk(64, 64), mk(64)], predictor=[mk(2, 64), mk(2)]))
    sf = mk(agents, 7, scale=1.0)
    head = [mk(64, 128), mk(64), mk(1, 64), mk(1)] if with_head else None
    res = ops.fused_pinnsf(brs, sf, tau, fold_epilogue=True, head=head)
    acc, msgs = res[0], res[1]
    wa = torch.randn(agents, 2, generator=g).to(DEV)
    loss = (acc * wa).sum() + sum((m * 1e-2).sum() for m in msgs) + (res[2].sum() if with_head else 0.0)
    leaves = [sf] + [t for br in brs for t in (br['x'], *br['encoder'], *br['decoder'], *br['predictor'])] + (head or [])
    grads = torch.autograd.grad(loss, leaves)

    d = lambda t: t.detach().double().requires_grad_(True)
    sf64 = d(sf)
    leaves64, acc64, msgs64 = [sf64], 0.0, []
    for br in brs:
        x, e, dd, p = d(br['x']), [d(t) for t in br['encoder']], [d(t) for t in br['decoder']], [d(t) for t in br['predictor']]
        h = torch.relu(x @ e[0].t() + e[1])
        h = torch.relu(h @ e[2].t() + e[3])
        m = 2.0 * (h @ e[4].t() + e[5])
        pooled = m.sum(dim=-2)
        acc64 = acc64 + (torch.relu(pooled @ dd[0].t() + dd[1]) @ dd[2].t() + dd[3]) @ p[0].t() + p[1]
        msgs64.append(m)
        leaves64 += [x, *e, *dd, *p]
    t = torch.norm(sf64[:, :2], dim=-1, keepdim=True)
    t = torch.where(t == 0, t + 0.1, t)
    acc64 = acc64 + (sf64[:, 6:7] * sf64[:, :2] / t - sf64[:, 2:4]) / tau
    loss64 = (acc64 * wa.double()).sum() + sum((m * 1e-2).sum() for m in msgs64)
    if with_head:
        h64 = [d(t) for t in head]
        loss64 = loss64 + torch.sigmoid(torch.relu(msgs64[0] @ h64[0].t() + h64[1]) @ h64[2].t() + h64[3]).sum()
        leaves64 += h64
    grads64 = torch.autograd.grad(loss64, leaves64)
    worst = float((acc.double() - acc64).abs().max() / acc64.abs().max())
    for a, b in zip(grads, grads64):
        worst = max(worst, float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)))
    print(f'fused_pinnsf nbr={nbr} agents={agents} k={ks} head={with_head}: max rel err vs float64 {worst:.1e}')
    assert worst <= 2e-5


@pytest.mark.parametrize('agents', [1, 37, 122, 1024])
def test_split_tile_forward_is_bitwise_the_one_wave_forward(agents):
    """enc_fwd_split_kernel (four waves per 32-row tile, few rows) against enc_fwd_kernel (one wave per tile): every
    accumulator sees its k-steps in the same order, so the messages, the saved activations (seen through the gradients)
    and the pooled sums are bitwise identical.  Both branches, ragged tiles, an odd number of tiles."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(31)

    def branch(k):
        x = torch.randn(agents, k, 6, generator=g).to(DEV).requires_grad_(True)
        w = [(torch.randn(*d, generator=g) * 0.2).to(DEV).requires_grad_(True)
             for d in [(128, 6), (128,), (128, 128), (128,), (128, 128), (128,)]]
        return dict(x=x, scale=2.0, weights=w, pooled=True)
    brs = [branch(6), branch(10)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    res = {}
    old = L.piml_encoder_split_tiles(-1)
    old_products = L.piml_encoder_products(0)        # the four-wave kernels use the f32 matrix instruction: compare like with like
    try:
        for split in (True, False):
            L.piml_encoder_split_tiles(1 << 30 if split else 0)
            outs = ops.fused_encoders(brs)
            loss = sum((m * 1e-2).sum() + p.square().sum() for m, p in outs)
            res[split] = [t.detach().clone() for o in outs for t in o] + list(torch.autograd.grad(loss, leaves))
    finally:
        L.piml_encoder_split_tiles(-2)          # both bounds back to their defaults
        L.piml_encoder_products(old_products)
    assert len(res[True]) == len(res[False])
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


def test_split_bf16_products_are_f32_arithmetic():
    """encoder_x3.hip evaluates every f32 product of the two 128 x 128 layers as six bf16 x bf16 partial products of exact
    three-way splits (hi + mid + lo == the f32 value), accumulated in f32, instead of one v_mfma_f32_32x32x2_f32 step.
    The claim that this IS f32 arithmetic is measured here: at the bench shape (4096 agents, one-wave kernels, both
    branches) outputs and every gradient of both forms against float64.  Bars: the split form within 1e-6 of float64 on
    every tensor (north-star bar: 1e-5) and not further from float64 than 1.5 x the f32 instruction's error + 1e-7."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    shapes = [(4096, 12, 6), (4096, 4, 6)]
    brs = [make_branch(n, k, d, seed=3 + i) for i, (n, k, d) in enumerate(shapes)]
    gen = torch.Generator().manual_seed(5)
    gps = [torch.randn(n, H, generator=gen).to(DEV) for n, _, _ in shapes]
    refs = [reference(br, None, gp) for br, gp in zip(brs, gps)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    names = ('msgs', 'pooled', 'g_x', 'dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3')
    err = {}
    old = L.piml_encoder_products(-1)
    try:
        for mode in (0, 1):
            L.piml_encoder_products(mode)
            outs = ops.fused_encoders(brs)
            grads = torch.autograd.grad(sum((p * gp).sum() for (m, p), gp in zip(outs, gps)), leaves)
            worst = dict.fromkeys(names, 0.0)
            for i, ((m, p), ref) in enumerate(zip(outs, refs)):
                got = (m.detach(), p.detach(), *grads[7 * i:7 * i + 7])
                want = (ref[0], ref[1], ref[2], *ref[3])
                for nm, a, b in zip(names, got, want):
                    worst[nm] = max(worst[nm], relerr(a, b))
            err[mode] = worst
    finally:
        L.piml_encoder_products(old)
    for mode, label in ((0, 'f32 instruction'), (1, 'split bf16    ')):
        print(f'{label}: max rel err vs float64 ' + ', '.join(f'{k} {v:.1e}' for k, v in err[mode].items()))
    for nm in names:
        assert err[1][nm] <= 1e-6, (nm, err[1][nm])
        assert err[1][nm] <= 1.5 * err[0][nm] + 1e-7, (nm, err[1][nm], err[0][nm])


@pytest.mark.parametrize('shapes', [[(4096, 12, 6), (4096, 4, 6)], [(5000, 7, 5), (33, 3, 8)], [(2440, 6, 6), (2440, 10, 6)]])
def test_split_product_kernels_are_bit_reproducible(shapes):
    """Forward + backward of the one-wave split-product kernels (sign-bit masks, ragged last tile, both branches) repeated
    30 times: every output and gradient bitwise equal to the first run.  (A version of the masked dX chain returned
    run-to-run different d/dx while every other tensor was right; this is its regression test.)"""
    from piml_amd import ops
    brs = [make_branch(n, k, d, seed=10 * i + n % 7) for i, (n, k, d) in enumerate(shapes)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    first = None
    for rep in range(30):
        outs = ops.fused_encoders(brs)
        loss = sum((m * 1e-2).sum() + p.square().sum() for m, p in outs)
        res = [t.detach().clone() for o in outs for t in o] + list(torch.autograd.grad(loss, leaves))
        if first is None:
            first = res
        else:
            bad = [i for i, (a, b) in enumerate(zip(first, res)) if not torch.equal(a, b)]
            assert not bad, (rep, bad)


@pytest.mark.parametrize('agents', [1, 37, 122, 1024])
def test_split_tile_x3_forward_is_bitwise_the_one_wave_x3_forward(agents):
    """enc_fwd_split_x3_kernel (four waves per tile, few rows: the rollouts of real clips) against enc_fwd_x3_kernel: every
    accumulator sees its k-blocks and the six products of a k-block in the same order, so messages and pooled sums are
    bitwise identical (inference call: the forward alone)."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(33)

    def branch(k):
        x = torch.randn(agents, k, 6, generator=g).to(DEV)
        w = [(torch.randn(*d, generator=g) * 0.2).to(DEV) for d in [(128, 6), (128,), (128, 128), (128,), (128, 128), (128,)]]
        return dict(x=x, scale=2.0, weights=w, pooled=True)
    brs = [branch(6), branch(10)]
    res = {}
    old, old_products = L.piml_encoder_split_tiles(-1), L.piml_encoder_products(1)
    try:
        for split in (True, False):
            L.piml_encoder_split_tiles(1 << 30 if split else 0)
            with torch.no_grad():
                res[split] = [t.clone() for o in ops.fused_encoders(brs) for t in o]
    finally:
        L.piml_encoder_split_tiles(-2)          # both bounds back to their defaults
        L.piml_encoder_products(old_products)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('agents,drop', [(1, False), (37, False), (122, True), (700, False), (1024, True)])
def test_split_tile_x3_backward_is_bitwise_the_one_wave_x3_backward(agents, drop):
    """enc_bwd_dx_split_x3_kernel (four waves per tile: the dX chain of the fine-tuning loop on real clips) against
    enc_bwd_dx_x3_kernel: same k-block order, same six products, same masks -- every gradient is bitwise identical, with and
    without a dropout keep-mask, with upstream gradients on the messages and on the pooled sums."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(35)

    def branch(k):
        x = torch.randn(agents, k, 6, generator=g).to(DEV).requires_grad_(True)
        w = [(torch.randn(*d, generator=g) * 0.2).to(DEV).requires_grad_(True)
             for d in [(128, 6), (128,), (128, 128), (128,), (128, 128), (128,)]]
        keep = ops.pack_keep_bits(torch.rand(agents * k, 128, generator=g) >= 0.5).to(DEV) if drop else None
        return dict(x=x, scale=4.0 if drop else 2.0, weights=w, pooled=True, keep_bits=keep)
    brs = [branch(6), branch(10)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    res = {}
    old, old_products = L.piml_encoder_split_tiles(-1), L.piml_encoder_products(1)
    old_dw2 = L.piml_encoder_dw2(0)      # the same weight-gradient kernel on both sides: this is about the two dX chains
    try:
        for split in (True, False):
            L.piml_encoder_split_tiles(1 << 30 if split else 0)
            outs = ops.fused_encoders(brs)
            loss = sum((m * 1e-2).sum() + p.square().sum() for m, p in outs)
            res[split] = [t.detach().clone() for o in outs for t in o] + list(torch.autograd.grad(loss, leaves))
    finally:
        L.piml_encoder_split_tiles(-2)          # both bounds back to their defaults
        L.piml_encoder_products(old_products)
        L.piml_encoder_dw2(old_dw2)
    assert len(res[True]) == len(res[False]) == 4 + 14
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('shapes,upstream,drop', [
    ([(4096, 6, 6), (4096, 10, 6)], 'pooled', False), ([(4096, 6, 6), (4096, 10, 6)], 'both', True),
    ([(5000, 7, 5), (3001, 5, 8)], 'msgs', False), ([(9000, 5, 6)], 'both', False), ([(4100, 9, 6), (4100, 9, 6)], 'mixed', True),
    ([(8192, 6, 6), (8192, 10, 6)], 'pooled', True),          # a rank's share of the 16384-agent scene on two GPUs: 2 tiles per wave
])
@pytest.mark.parametrize('recompute', [True, False])
def test_layer_split_weight_gradients(shapes, upstream, drop, recompute, monkeypatch):
    """piml_amd/csrc/encoder_dw2.hip (layer-split workgroups, producer / consumer waves; h1 recomputed from x when the forward
    did not store it) against float64 and against the slab kernel (encoder_dww.hip) on the same inputs: same products, another
    order of the row sums.  'mixed': the two branches carry different kinds of upstream gradient (two launches)."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    monkeypatch.setattr(ops, 'H1_RECOMPUTE', recompute)
    assert sum((n * k + 31) // 32 for n, k, _ in shapes) > L.piml_encoder_split_tiles_train(-1)
    g = torch.Generator().manual_seed(77)
    branches = [dodge_relu_kinks(make_branch(n, k, d, seed=3 * i + 1, scale=4.0 if drop else 2.0)) for i, (n, k, d) in enumerate(shapes)]
    if drop:
        for br, (n, k, d) in zip(branches, shapes):
            br['keep_bits'] = ops.pack_keep_bits(torch.rand(n * k, H, generator=g) >= 0.5).to(DEV)
    kinds = [upstream] * len(shapes) if upstream != 'mixed' else ['pooled', 'both']
    ups = []
    for (n, k, d), kind in zip(shapes, kinds):
        gm = torch.randn(n, k, H, generator=g).to(DEV) if kind in ('both', 'msgs') else None
        gp = torch.randn(n, H, generator=g).to(DEV) if kind in ('both', 'pooled') else None
        ups.append((gm, gp))
    leaves = [t for br in branches for t in (br['x'], *br['weights'])]
    res = {}
    old = L.piml_encoder_dw2(-1)
    try:
        for on in (1, 0):
            L.piml_encoder_dw2(on)
            outs = ops.fused_encoders(branches)
            loss = 0
            for (msgs, pooled), (gm, gp) in zip(outs, ups):
                loss = loss + ((msgs * gm).sum() if gm is not None else 0) + ((pooled * gp).sum() if gp is not None else 0)
            res[on] = torch.autograd.grad(loss, leaves)
    finally:
        L.piml_encoder_dw2(old)
    worst = {}
    it1, it0 = iter(res[1]), iter(res[0])
    for br, (gm, gp) in zip(branches, ups):
        ref = dict(br)
        if drop:        # the processor with an injected mask: keep * scale * x (scale = 2 / (1 - p) = 4)
            keep = ops.unpack_keep_bits(br['keep_bits'], H).to(DEV).double().view(*br['x'].shape[:-1], H)
            x = br['x'].detach().double().requires_grad_(True)
            w = [t.detach().double().requires_grad_(True) for t in br['weights']]
            h = torch.relu(torch.relu(x @ w[0].t() + w[1]) @ w[2].t() + w[3])
            msgs = keep * br['scale'] * (h @ w[4].t() + w[5])
            l64 = ((msgs * gm.double()).sum() if gm is not None else 0) + ((msgs.sum(-2) * gp.double()).sum() if gp is not None else 0)
            l64.backward()
            rgx, rgw = x.grad, [t.grad for t in w]
        else:
            _, _, rgx, rgw = reference(ref, gm, gp)
        for name, r in zip(('g_x', 'dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3'), [rgx, *rgw]):
            a, b = next(it1), next(it0)
            worst[name] = max(worst.get(name, 0), relerr(a, r))
            worst[name + ' vs slab'] = max(worst.get(name + ' vs slab', 0), relerr(a, b.double()))
    print(f'layer-split dW {shapes} upstream={upstream} drop={drop} recompute={recompute}: ' +
          ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    assert max(worst.values()) <= 1e-5, worst


def _float64_grads(br, gm, gp, drop):
    """float64 autograd of one branch, with the injected keep mask when `drop`."""
    from piml_amd import ops
    if not drop:
        _, _, rgx, rgw = reference(br, gm, gp)
        return [rgx, *rgw]
    keep = ops.unpack_keep_bits(br['keep_bits'], H).to(DEV).double().view(*br['x'].shape[:-1], H)
    x = br['x'].detach().double().requires_grad_(True)
    w = [t.detach().double().requires_grad_(True) for t in br['weights']]
    h = torch.relu(torch.relu(x @ w[0].t() + w[1]) @ w[2].t() + w[3])
    msgs = keep * br['scale'] * (h @ w[4].t() + w[5])
    l64 = ((msgs * gm.double()).sum() if gm is not None else 0) + ((msgs.sum(-2) * gp.double()).sum() if gp is not None else 0)
    l64.backward()
    return [x.grad, *[t.grad for t in w]]


@pytest.mark.parametrize('shapes,upstream,drop', [
    ([(4096, 6, 6), (4096, 10, 6)], 'pooled', False), ([(4096, 6, 6), (4096, 10, 6)], 'pooled', True),
    ([(4096, 6, 6), (4096, 10, 6)], 'both', True), ([(5000, 7, 5), (3001, 5, 8)], 'msgs', False), ([(9000, 5, 6)], 'both', False),
    ([(4099, 9, 3), (2050, 13, 1)], 'pooled', False),         # ragged last tiles, in_dim 3 and 1, k that does not divide 32
    ([(8192, 6, 6), (8192, 10, 6)], 'pooled', True), ([(16384, 6, 6), (16384, 10, 6)], 'pooled', False),
])
@pytest.mark.parametrize('form', [1, 2])
def test_one_pass_backward(shapes, upstream, drop, form):
    """piml_amd/csrc/encoder_bwd3.hip -- dX chain + dW2 / dW1 / db2 / db1 in one launch, g2 / g1 never stored -- against float64
    and against the two-kernel form (enc_bwd_dx_x3_kernel + enc_bwd_dw2_x3_kernel) on the same inputs; twice, bitwise equal
    (no atomics).  The no-input-gradient form (pointwise training) rides along: same weight gradients bit for bit."""
    from piml_amd import ops, _lib
    L = _lib.lib()
    assert sum((n * k + 31) // 32 for n, k, _ in shapes) > L.piml_encoder_split_tiles_train(-1)
    g = torch.Generator().manual_seed(78)
    branches = [dodge_relu_kinks(make_branch(n, k, d, seed=3 * i + 2, scale=4.0 if drop else 2.0)) for i, (n, k, d) in enumerate(shapes)]
    if drop:
        for br, (n, k, d) in zip(branches, shapes):
            br['keep_bits'] = ops.pack_keep_bits(torch.rand(n * k, H, generator=g) >= 0.5).to(DEV)
    ups = []
    for (n, k, d) in shapes:
        gm = torch.randn(n, k, H, generator=g).to(DEV) if upstream in ('both', 'msgs') else None
        gp = torch.randn(n, H, generator=g).to(DEV) if upstream in ('both', 'pooled') else None
        ups.append((gm, gp))
    leaves = [t for br in branches for t in (br['x'], *br['weights'])]

    def run():
        outs = ops.fused_encoders(branches)
        loss = 0
        for (msgs, pooled), (gm, gp) in zip(outs, ups):
            loss = loss + ((msgs * gm).sum() if gm is not None else 0) + ((pooled * gp).sum() if gp is not None else 0)
        return torch.autograd.grad(loss, leaves)
    res = {}
    old = L.piml_encoder_fused_bwd(-1)
    try:
        for on in (form, 0):
            L.piml_encoder_fused_bwd(on)
            res[1 if on else 0] = run()
        L.piml_encoder_fused_bwd(form)
        again = run()
        for br in branches:
            br['x'].requires_grad_(False)
        wleaves = [t for br in branches for t in br['weights']]
        outs = ops.fused_encoders(branches)
        loss = 0
        for (msgs, pooled), (gm, gp) in zip(outs, ups):
            loss = loss + ((msgs * gm).sum() if gm is not None else 0) + ((pooled * gp).sum() if gp is not None else 0)
        no_gx = torch.autograd.grad(loss, wleaves)
    finally:
        L.piml_encoder_fused_bwd(old)
        for br in branches:
            br['x'].requires_grad_(True)
    for a, b in zip(res[1], again):
        assert torch.equal(a, b), 'one-pass backward is not reproducible'
    for a, b in zip([t for i, t in enumerate(res[1]) if i % 7], no_gx):
        assert torch.equal(a, b), 'weight gradients depend on whether g_x is wanted'
    worst = {}
    it1, it0 = iter(res[1]), iter(res[0])
    for br, (gm, gp) in zip(branches, ups):
        for name, r in zip(('g_x', 'dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3'), _float64_grads(br, gm, gp, drop)):
            a, b = next(it1), next(it0)
            worst[name] = max(worst.get(name, 0), relerr(a, r))
            worst[name + ' vs two kernels'] = max(worst.get(name + ' vs two kernels', 0), relerr(a, b.double()))
    print(f'one-pass backward {shapes} upstream={upstream} drop={drop}: ' + ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    assert max(worst.values()) <= 1e-5, worst


@pytest.mark.parametrize('shapes', [[(4096, 6, 6), (4096, 10, 6)], [(5000, 7, 5), (33, 3, 8)]])
def test_sign_bit_masks_equal_the_saved_activations(shapes, monkeypatch):
    """The dX chain of the split-product kernels masks with the signs of h1 / h2 read as bits (piml_encoder_branch.relu_mask,
    written by the forward) or, without that buffer, with the saved activations themselves: the same predicate, so every
    output and gradient is bitwise identical."""
    from piml_amd import ops, _lib
    old = _lib.lib().piml_encoder_fused_bwd(0)         # (the one-pass backward needs the bits: test_one_pass_backward)
    res = {}
    for masks in (True, False):
        monkeypatch.setattr(ops, 'RELU_MASK', masks)
        brs = [make_branch(n, k, d, seed=10 * i + n % 7) for i, (n, k, d) in enumerate(shapes)]
        leaves = [t for br in brs for t in (br['x'], *br['weights'])]
        outs = ops.fused_encoders(brs)
        loss = sum((m * 1e-2).sum() + p.square().sum() for m, p in outs)
        res[masks] = [t.detach().clone() for o in outs for t in o] + list(torch.autograd.grad(loss, leaves))
    _lib.lib().piml_encoder_fused_bwd(old)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('xscale,wscale', [(1e-6, 1.0), (1e4, 1.0), (1.0, 1e-4), (1e3, 30.0)])
def test_split_products_hold_across_magnitudes(xscale, wscale):
    """The three-way split is exact whatever the magnitude (bf16 has f32's exponent range): inputs of 1e-6 .. 1e4 and
    weights of 1e-4 .. 30 times the usual scale, outputs and gradients against float64 at the one-wave kernels' row counts.
    Bar 1e-6 of each tensor's largest magnitude."""
    from piml_amd import ops
    shapes = [(4096, 6, 6), (4096, 10, 6)]
    brs = [make_branch(n, k, d, seed=21 + i) for i, (n, k, d) in enumerate(shapes)]
    with torch.no_grad():
        for br in brs:
            br['x'].mul_(xscale)
            for w in br['weights'][0::2]:
                w.mul_(wscale)
    gen = torch.Generator().manual_seed(9)
    gps = [torch.randn(n, H, generator=gen).to(DEV) for n, _, _ in shapes]
    outs = ops.fused_encoders(brs)
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    grads = torch.autograd.grad(sum((p * gp).sum() for (m, p), gp in zip(outs, gps)), leaves)
    worst = 0.0
    for i, (br, (m, p), gp) in enumerate(zip(brs, outs, gps)):
        rm, rp, rgx, rgw = reference(br, None, gp)
        for a, b in zip((m.detach(), p.detach(), *grads[7 * i:7 * i + 7]), (rm, rp, rgx, *rgw)):
            assert torch.isfinite(a).all()
            worst = max(worst, relerr(a, b))
    print(f'split products, x scale {xscale:g}, weight scale {wscale:g}: max rel err vs float64 {worst:.1e}')
    assert worst <= 1e-6


@pytest.mark.parametrize('rows', [(4096, 6), (122, 6), (1, 1), (37, 5), (3, 250, 6), (4096, 17), (4 * 122, 6)])       # (4096 x 17: above the one-tile-per-workgroup bound)
def test_collision_head64_matches_float64(rows):
    """ops.collision_head64 (`pinnsf_bm`'s collision head, src/models/model.py:1183, 1214-1215, on head64.hip) against a
    float64 evaluation: output and every gradient; and bit-reproducible (no atomics)."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(41)
    x = torch.randn(*rows, 64, generator=g).to(DEV).requires_grad_(True)
    w = [(torch.randn(*d, generator=g) * (0.3 if len(d) == 2 else 0.1)).to(DEV).requires_grad_(True) for d in [(64, 64), (64,), (1, 64), (1,)]]
    go = torch.randn(*rows, generator=g).to(DEV)
    runs = []
    for _ in range(2):
        out = ops.collision_head64(x, *w)
        runs.append([out.detach().clone()] + [t.clone() for t in torch.autograd.grad((out * go).sum(), [x, *w])])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    xd = x.detach().double().requires_grad_(True)
    wd = [t.detach().double().requires_grad_(True) for t in w]
    ref = torch.sigmoid(torch.relu(xd @ wd[0].t() + wd[1]) @ wd[2].t() + wd[3]).squeeze(-1)
    gref = torch.autograd.grad((ref * go.double()).sum(), [xd, *wd])
    worst = 0.0
    for a, b in zip(runs[0], [ref.detach(), *gref]):
        assert a.shape == b.shape
        worst = max(worst, float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)))
    print(f'collision_head64 rows={rows}: max rel err vs float64 {worst:.1e}')
    assert worst <= 1e-5          # north-star bar; measured 2e-7 .. 2e-6 (the largest for a single row: nothing averages)


@pytest.mark.parametrize('name', ['PINNSF_bottleneck_multitask', 'PINNSF_bottleneck'])
def test_bottleneck_variants_with_prepacked_weights_are_bitwise_neutral(name):
    """`packed_weights()` for the bottleneck variants: ONE pack launch for the encoders' and the row decoders' operand
    images, skipped by every forward pass inside the block (a rollout, the frames of a training window).  Same kernels on
    the same data: bitwise equal outputs and gradients; a stale image (weights changed inside the block) is refused."""
    import types
    import piml_amd.models.model as MODEL
    from piml_amd import ops
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128, processor_hidden_size=128,
        decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5,
        activation='relu', dataset_name='gc1560')
    torch.manual_seed(0)
    net = getattr(MODEL, name)(args).to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    base = [torch.randn(700, 6, 6, generator=g).to(DEV), torch.randn(700, 10, 6, generator=g).to(DEV), torch.randn(700, 7, generator=g).to(DEV)]

    def run():
        ins = [t.clone().requires_grad_(True) for t in base]
        net.zero_grad(set_to_none=True)
        out = net(*ins)
        sum((o * 0.5).sum() for o in out).backward()
        return [o.detach().clone() for o in out] + [t.grad.clone() for t in ins] + [p.grad.clone() for p in net.parameters() if p.grad is not None]
    ref = run()
    calls = []
    real = ops.pinnsf_prepack
    ops.pinnsf_prepack = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with net.packed_weights():
            packed, packed2 = run(), run()
    finally:
        ops.pinnsf_prepack = real
    assert calls == [1] and net._packs is not None
    for other in (packed, packed2):
        assert len(other) == len(ref)
        for a, b in zip(ref, other):
            assert torch.equal(a, b)
    with net.packed_weights():
        with torch.no_grad():
            net.ped_decoder.mlp[0].weight.mul_(1.5)
        with pytest.raises(ValueError, match='modified'):
            net(*base)
