"""GPU: the fused glue kernels around the PINNSF GEMMs (piml_amd/csrc/mlpglue.hip) against the plain
torch expression of the same arithmetic, and the fused model against the reference's golden outputs
(tests/golden/model.npz, captured from src/models/model.py)."""
import types

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def torch_desired(acc_p, acc_o, sf, tau):
    """src/models/model.py:1289-1294 (per-row norm)."""
    v0 = sf[..., -1].unsqueeze(-1)
    t = torch.norm(sf[..., :2], p=2, dim=-1, keepdim=True)
    t = torch.where(t == 0, t + 0.1, t)
    acc = acc_p if acc_o is None else acc_p + acc_o
    return acc + (v0 * (sf[..., :2] / t) - sf[..., 2:4]) / tau


@pytest.mark.parametrize('shape', [(1000,), (3, 257)])
@pytest.mark.parametrize('with_obs', [True, False])
def test_pinnsf_epilogue_matches_torch(shape, with_obs):
    from piml_amd import ops
    sf = rnd(*shape, 7, seed=1)
    sf.view(-1, 7)[::7, :2] = 0.0                      # arrived / absent agents: dest - p == 0
    acc_p, acc_o = rnd(*shape, 2, seed=2), (rnd(*shape, 2, seed=3) if with_obs else None)
    leaves = [x.clone().requires_grad_(True) for x in (acc_p, sf)] + ([acc_o.clone().requires_grad_(True)] if with_obs else [])
    ref = torch_desired(leaves[0], leaves[2] if with_obs else None, leaves[1], 0.5)
    w = rnd(*shape, 2, seed=4)
    g_ref = torch.autograd.grad(ref, leaves, w)
    leaves2 = [x.detach().clone().requires_grad_(True) for x in leaves]
    out = ops.pinnsf_epilogue(leaves2[0], leaves2[2] if with_obs else None, leaves2[1], 0.5)
    g_out = torch.autograd.grad(out, leaves2, w)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    for a, b in zip(g_out, g_ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()


def test_self_features_packed_matches_cat():
    from piml_amd import ops
    n = 777
    dest, state, v0 = rnd(n, 2, seed=1), rnd(n, 6, seed=2), rnd(n, 1, seed=3)
    a = [x.clone().requires_grad_(True) for x in (dest, state, v0)]
    ref = torch.cat((a[0], a[1][:, 2:4], a[1][:, 4:6], a[2]), -1)
    w = rnd(n, 7, seed=4)
    g_ref = torch.autograd.grad(ref, a, w)
    b = [x.clone().requires_grad_(True) for x in (dest, state, v0)]
    out = ops.self_features_packed(*b)
    g_out = torch.autograd.grad(out, b, w)
    assert torch.equal(out, ref)
    for x, y in zip(g_out, g_ref):
        assert torch.equal(x, y)


@pytest.mark.parametrize('rows,cols', [(24576, 128), (40960, 128), (4096, 64), (4096, 2), (24576, 1), (1, 128),
                                       (63, 64), (65, 12), (1000, 6), (16385, 256), (300, 1024), (5000, 3), (0, 64),
                                       (129, 1028), (77, 300)])
@pytest.mark.parametrize('masked', [True, False])
def test_act_bwd_colsum(rows, cols, masked):
    from piml_amd import ops
    g = rnd(rows, cols, seed=rows + cols)
    y = torch.relu(rnd(rows, cols, seed=7)) if masked else None
    g_pre, db = ops.act_bwd_colsum(g, y)
    want_pre = torch.where(y > 0, g, torch.zeros((), device=DEV)) if masked else g
    assert torch.equal(g_pre, want_pre)
    want = want_pre.double().sum(0)
    tol = 1e-6 * max(1.0, float(want_pre.abs().double().sum(0).max()) if rows else 1.0)
    assert db.shape == (cols,)
    assert (db.double() - want).abs().max() <= tol, ((db.double() - want).abs().max(), tol)
    # deterministic: fixed two-level summation order
    for _ in range(5):
        _, again = ops.act_bwd_colsum(g, y)
        assert torch.equal(again, db)


def test_act_bwd_colsum_concurrent_streams():
    """Two streams launching colsums at the same time (each call owns its scratch)."""
    from piml_amd import ops
    g1, g2 = rnd(40960, 128, seed=1), rnd(24576, 128, seed=2)
    want1, want2 = ops.act_bwd_colsum(g1)[1], ops.act_bwd_colsum(g2)[1]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(20):
        with torch.cuda.stream(s1):
            outs.append((ops.act_bwd_colsum(g1)[1], want1))
        with torch.cuda.stream(s2):
            outs.append((ops.act_bwd_colsum(g2)[1], want2))
    torch.cuda.synchronize()
    for got, want in outs:
        assert torch.equal(got, want)


@pytest.mark.parametrize('shape,out_f,relu', [((4096, 6, 6), 128, True), ((4096, 10, 128), 128, False),
                                             ((4096, 128), 64, True), ((4096, 64), 2, False), ((5, 7, 3, 16), 12, True)])
def test_linear_act_matches_torch(shape, out_f, relu):
    from piml_amd import ops
    x = rnd(*shape, seed=1)
    lin = torch.nn.Linear(shape[-1], out_f).to(DEV)
    xa = x.clone().requires_grad_(True)
    ref = lin(xa)
    ref = torch.relu(ref) if relu else ref
    w = rnd(*ref.shape, seed=2)
    g_ref = torch.autograd.grad(ref, [xa, lin.weight, lin.bias], w)
    xb = x.clone().requires_grad_(True)
    out = ops.linear_act(xb, lin.weight, lin.bias, relu)
    g_out = torch.autograd.grad(out, [xb, lin.weight, lin.bias], w)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    for a, b in zip(g_out, g_ref):
        scale = max(1.0, float(b.abs().max()))
        assert (a - b).abs().max() <= 2e-5 * scale, ((a - b).abs().max(), scale)


@pytest.mark.parametrize('use', ['pooled', 'both', 'msgs'])
@pytest.mark.parametrize('with_bias', [False, True])
def test_scale_ksum_matches_torch(use, with_bias):
    from piml_amd import ops
    e = rnd(513, 6, 128, seed=1)
    bias = rnd(128, seed=9) if with_bias else None
    ea = e.clone().requires_grad_(True)
    eb_ = ea + bias if with_bias else ea
    m_ref = eb_ + eb_
    p_ref = m_ref.sum(dim=-2)
    eb = e.clone().requires_grad_(True)
    m, p = ops.scale_ksum(eb, 2.0, bias=bias)
    assert torch.equal(m, m_ref)
    assert torch.allclose(p, p_ref, rtol=1e-5, atol=1e-5)
    wm, wp = rnd(513, 6, 128, seed=2), rnd(513, 128, seed=3)
    loss = lambda mm, pp: ((pp * wp).sum() if use != 'msgs' else 0) + ((mm * wm).sum() if use != 'pooled' else 0)
    (g_ref,) = torch.autograd.grad(loss(m_ref, p_ref), ea)
    (g_out,) = torch.autograd.grad(loss(m, p), eb)
    assert torch.allclose(g_out, g_ref, rtol=1e-5, atol=1e-5)


# ---- the whole network: fused glue vs the reference's outputs, and vs the unfused torch.nn path ----
CASES = {
    'pinnsf_m': ('PINNSF_multitask', {}),
    'pinnsf_m_gc': ('PINNSF_multitask', dict(dataset_name='gc1560')),
    'pinnsf_bm': ('PINNSF_bottleneck_multitask', {}),
    'pinnsf': ('PINNSF', {}),
    'pinnsf_bottleneck': ('PINNSF_bottleneck', {}),
    'pinnsf_res': ('PINNSF_residual', {}),
    'pinnsf_m_p1': ('PINNSF_multitask', dict(processor_hidden_layers=1)),
}


def model_args(**kw):
    a = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3,
        processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5, activation='relu',
        dataset_name='ucy', res_hidden_layers=3, correction_hidden_layers=1, time_unit=0.08,
        collision_threshold=0.5)
    a.__dict__.update(kw)
    return a


def load_model(name):
    import piml_amd.models.model as MODEL
    g = golden('model')
    cls, kw = CASES[name]
    m = getattr(MODEL, cls)(model_args(**kw)).eval()
    sd = {k[len(name) + 4:]: torch.tensor(g[k]) for k in g.files if k.startswith(name + '/sd/')}
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), g, MODEL


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_model_matches_reference_outputs(name):
    m, g, MODEL = load_model(name)
    assert MODEL.FUSED_GLUE
    with torch.no_grad():
        for tag, keys in (('n', ('ped', 'obs', 'selff')), ('c', ('pedc', 'obsc', 'selfc'))):
            outs = m(*[torch.tensor(g[k]).to(DEV) for k in keys])
            q = 0
            while f'{name}/out_{tag}{q}' in g.files:
                ref = g[f'{name}/out_{tag}{q}']
                got = outs[q].cpu().numpy()
                assert got.shape == ref.shape
                scale = max(1.0, np.abs(ref).max())
                assert np.abs(got - ref).max() <= 2e-5 * scale, (tag, q, np.abs(got - ref).max())
                q += 1
            assert q == len(outs)


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('mode', ['rows', 'channelled_fix', 'channelled_quirk'])
def test_fused_model_equals_unfused(name, mode, monkeypatch):
    """Outputs and every gradient (parameters and the three inputs) of the fused path against the plain
    torch.nn path on the GPU; the loss touches all outputs (acc, messages, collision head)."""
    m, g, MODEL = load_model(name)
    keys = ('ped', 'obs', 'selff') if mode == 'rows' else ('pedc', 'obsc', 'selfc')
    m.fix_dest_norm = mode == 'channelled_fix'
    base = [torch.tensor(g[k]).to(DEV) for k in keys]

    def run(fused):
        monkeypatch.setattr(MODEL, 'FUSED_GLUE', fused)
        ins = [x.clone().requires_grad_(True) for x in base]
        for p in m.parameters():
            p.grad = None
        outs = m(*ins)
        loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=DEV).view_as(o)).sum() for o in outs)
        loss.backward()
        grads = {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()}
        return [o.detach() for o in outs], [x.grad for x in ins], grads

    o1, i1, p1 = run(True)
    o0, i0, p0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    for a, b in zip(i1, i0):
        scale = max(1.0, float(b.abs().max()))
        assert (a - b).abs().max() <= 1e-4 * scale
    for k in p0:
        assert (p0[k] is None) == (p1[k] is None), k
        if p0[k] is not None:
            scale = max(1.0, float(p0[k].abs().max()))
            assert (p1[k] - p0[k]).abs().max() <= 1e-4 * scale, (k, (p1[k] - p0[k]).abs().max(), scale)


@pytest.mark.parametrize('sizes,relus', [((6, 128, 128, 128), (True, True, False)), ((128, 64, 64), (True, False)),
                                         ((64, 2), (False,)), ((128, 64, 1), (True, False))])
def test_mlp_chain_matches_torch(sizes, relus):
    """One autograd node for a whole MLP."""
    from piml_amd import ops
    torch.manual_seed(3)
    lins = [torch.nn.Linear(a, b).to(DEV) for a, b in zip(sizes[:-1], sizes[1:])]
    x = rnd(4096, 6, sizes[0], seed=1)

    def ref_forward(inp):
        h = inp
        for lin, r in zip(lins, relus):
            h = lin(h)
            h = torch.relu(h) if r else h
        return h
    params = [t for lin in lins for t in (lin.weight, lin.bias)]
    xa = x.clone().requires_grad_(True)
    ref = ref_forward(xa)
    w = rnd(*ref.shape, seed=2)
    g_ref = torch.autograd.grad(ref, [xa] + params, w)
    xb = x.clone().requires_grad_(True)
    out = ops.mlp_chain(xb, relus, *params)
    g_out = torch.autograd.grad(out, [xb] + params, w)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    for a, b in zip(g_out, g_ref):
        scale = max(1.0, float(b.abs().max()))
        assert (a - b).abs().max() <= 2e-5 * scale, ((a - b).abs().max(), scale)
    # only some gradients requested (frozen weights / no input gradient)
    (gw_last,) = torch.autograd.grad(ops.mlp_chain(x, relus, *params), [params[-2]], w)
    assert (gw_last - g_ref[-2]).abs().max() <= 2e-5 * max(1.0, float(g_ref[-2].abs().max()))


@pytest.mark.parametrize('f0,fc', [(0, 300), (100, 120)])
def test_relative_features_packed_self_matches_separate_ops(f0, fc):
    """relfeat + self-feature rows as one autograd node == relative_features_packed + torch.cat."""
    from piml_amd import ops
    from piml_amd.scenes import synthetic_gc_scene
    sc = synthetic_gc_scene(300, 200, seed=5)
    rng = np.random.default_rng(1)
    acc = (rng.standard_normal((300, 2)) * 0.3).astype(np.float32)
    state = torch.tensor(np.concatenate((sc['position'], sc['velocity'], acc), -1)).to(DEV)
    dest = torch.tensor(sc['destination'][f0:f0 + fc]).to(DEV)
    v0 = torch.tensor(sc['desired_speed'][f0:f0 + fc]).to(DEV)
    obs = torch.tensor(sc['obstacles']).to(DEV)
    wp, wo, ws = rnd(fc, 6, 6, seed=1), rnd(fc, 10, 6, seed=2), rnd(fc, 7, seed=3)

    sa, da, va = state.clone().requires_grad_(True), dest.clone().requires_grad_(True), v0.clone().requires_grad_(True)
    pf, of, df = ops.relative_features_packed(sa, da, obs, f0, fc)
    own = sa[f0:f0 + fc]
    sf = torch.cat((df, own[:, 2:4], own[:, 4:6], va), -1)
    g_ref = torch.autograd.grad((pf * wp).sum() + (of * wo).sum() + (sf * ws).sum(), [sa, da, va])

    sb, db_, vb = state.clone().requires_grad_(True), dest.clone().requires_grad_(True), v0.clone().requires_grad_(True)
    pf2, of2, sf2 = ops.relative_features_packed_self(sb, db_, obs, vb, f0, fc)
    g_out = torch.autograd.grad((pf2 * wp).sum() + (of2 * wo).sum() + (sf2 * ws).sum(), [sb, db_, vb])
    for a, b in ((pf2, pf), (of2, of), (sf2, sf)):
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    for a, b in zip(g_out, g_ref):
        assert torch.allclose(torch.nan_to_num(a), torch.nan_to_num(b), rtol=1e-5, atol=1e-6)
    # only the feature gradients requested (self_features unused downstream)
    (g_only,) = torch.autograd.grad((ops.relative_features_packed_self(sb, db_, obs, vb, f0, fc)[0] * wp).sum(), [sb])
    (g_only_ref,) = torch.autograd.grad((ops.relative_features_packed(sa, da, obs, f0, fc)[0] * wp).sum(), [sa])
    assert torch.allclose(torch.nan_to_num(g_only), torch.nan_to_num(g_only_ref), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('N,M,f0,fc', [(300, 200, 0, 300), (300, 200, 100, 120), (300, 200, 0, 7), (300, 200, 293, 7),
                                        (5, 3, 1, 3), (16384, 2000, 6144, 2048), (5000, 4500, 4096, 904)])
def test_relative_features_split_parts_equal_one_launch(N, M, f0, fc):
    """Agent-block sharding: the LOCAL part (own block's sources + obstacles + self features; reads no row outside the
    block) followed by the REMOTE part (the other agents, continuing the saved list) is bit-identical to the one-launch
    form -- features, indices, and the gradients of the one backward launch -- incl. blocks at either end of the scene,
    k > block size, NaN (absent) agents and the cfg4 shard shape (16384 agents, blocks of 2048)."""
    from piml_amd import ops
    from piml_amd.scenes import synthetic_gc_scene
    sc = synthetic_gc_scene(N, M, seed=9)
    rng = np.random.default_rng(2)
    acc = (rng.standard_normal((N, 2)) * 0.3).astype(np.float32)
    state = torch.tensor(np.concatenate((sc['position'], sc['velocity'], acc), -1)).to(DEV)
    if N > 100:
        state[rng.integers(0, N, 5)] = float('nan')                  # absent agents: some focal, some sources
    dest = torch.tensor(sc['destination'][f0:f0 + fc]).to(DEV)
    v0 = torch.tensor(sc['desired_speed'][f0:f0 + fc]).to(DEV)
    obs = torch.tensor(sc['obstacles']).to(DEV)
    ko = min(10, obs.shape[0])
    kp = min(6, N)
    wp, wo, ws = rnd(fc, kp, 6, seed=1), rnd(fc, ko, 6, seed=2), rnd(fc, 7, seed=3)

    sa, va = state.clone().requires_grad_(True), v0.clone().requires_grad_(True)
    ref = ops.relative_features_packed_self(sa, dest, obs, va, f0, fc, return_index=True)
    g_ref = torch.autograd.grad((torch.nan_to_num(ref[0]) * wp).sum() + (torch.nan_to_num(ref[1]) * wo).sum() +
                                (torch.nan_to_num(ref[2]) * ws).sum(), [sa, va])

    sb, vb = state.clone().requires_grad_(True), v0.clone().requires_grad_(True)
    # the LOCAL part must not depend on rows outside the block: run it on a buffer whose other rows are garbage
    scratch = torch.full_like(state, 1e30)
    scratch[f0:f0 + fc] = state[f0:f0 + fc]
    local_probe = ops.relative_features_local_part(scratch, dest, obs, v0, f0, fc)
    local = ops.relative_features_local_part(sb, dest, obs, vb, f0, fc)
    for a, b in ((local.obs_feat, local_probe.obs_feat), (local.self_features, local_probe.self_features),
                 (local.ped_idx, local_probe.ped_idx), (local.obs_idx, local_probe.obs_idx)):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    out = ops.relative_features_packed_self(sb, dest, obs, vb, f0, fc, return_index=True, local=local)
    for a, b in zip(out, ref):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    g_out = torch.autograd.grad((torch.nan_to_num(out[0]) * wp).sum() + (torch.nan_to_num(out[1]) * wo).sum() +
                                (torch.nan_to_num(out[2]) * ws).sum(), [sb, vb])
    for a, b in zip(g_out, g_ref):
        assert torch.allclose(torch.nan_to_num(a), torch.nan_to_num(b), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):                                   # a LocalFeatures of another buffer is refused
        ops.relative_features_packed_self(sa, dest, obs, va, f0, fc, local=local)


@pytest.mark.parametrize('C,N', [(4, 122), (1, 1000), (3, 5)])
def test_pinnsf_epilogue_agent_norm_matches_torch(C, N):
    """Quirk Q2: the dim=1 norm of channelled input (over the agents of a slice, per component)."""
    from piml_amd import ops
    sf = rnd(C, N, 7, seed=1)
    sf[0, :, 1] = 0.0                                   # a slice whose y-components all vanish: t = 0 -> 0.1
    acc_p, acc_o = rnd(C, N, 2, seed=2), rnd(C, N, 2, seed=3)

    def ref(ap, ao, s, tau=0.5):
        v0 = s[..., -1].unsqueeze(-1)
        t = torch.norm(s[..., :2], p=2, dim=1, keepdim=True)
        t = torch.where(t == 0, t + 0.1, t)
        return ap + ao + (v0 * (s[..., :2] / t) - s[..., 2:4]) / tau
    a = [x.clone().requires_grad_(True) for x in (acc_p, acc_o, sf)]
    w = rnd(C, N, 2, seed=4)
    out_ref = ref(*a)
    g_ref = torch.autograd.grad(out_ref, a, w)
    b = [x.clone().requires_grad_(True) for x in (acc_p, acc_o, sf)]
    out = ops.pinnsf_epilogue(b[0], b[1], b[2], 0.5, agent_norm=True)
    g_out = torch.autograd.grad(out, b, w)
    assert torch.allclose(out, out_ref, rtol=1e-5, atol=1e-5)
    for x, y in zip(g_out, g_ref):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-5), (x - y).abs().max()


@pytest.mark.parametrize('rows,cin,cout', [(40960, 128, 128), (24576, 128, 128), (16384, 64, 128), (4096, 128, 64),
                                           (4096, 64, 64), (2928, 128, 128), (40960, 6, 128), (4096, 64, 2)])
def test_chunked_weight_gradient(rows, cin, cout, monkeypatch):
    """dW as one strided-batched GEMM over 64 row chunks + the HIP chunk sum == G^T X."""
    from piml_amd import ops, tuning
    g, x = rnd(rows, cout, seed=1), rnd(rows, cin, seed=2)
    want = g.double().t().mm(x.double())
    monkeypatch.setattr(tuning, 'LOADED', True)
    got = ops._weight_grad(g, x)
    monkeypatch.setattr(tuning, 'LOADED', False)
    plain = ops._weight_grad(g, x)
    scale = float(want.abs().max())
    assert got.shape == (cout, cin)
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale      # f32 accumulation over `rows` terms
    assert float((plain.double() - want).abs().max()) <= 2e-5 * scale
    parts = rnd(7, 33, 4, seed=3)
    assert torch.allclose(ops.sum_leading(parts), parts.sum(0), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('rows,cols', [(40960, 128), (4096, 2), (300, 64), (10, 128)])
@pytest.mark.parametrize('with_parts', [True, False])
def test_layer_reduce_deferred_colsum(rows, cols, with_parts):
    """First stage of the bias-gradient sum now, its second stage together with the chunk reduction later."""
    from piml_amd import ops
    g = rnd(rows, cols, seed=5)
    y = torch.relu(rnd(rows, cols, seed=6))
    want_pre, want_db = ops.act_bwd_colsum(g, y)
    g_pre, db, pending = ops.act_bwd_colsum(g, y, defer=True)
    parts = rnd(16, 64, 12, seed=7) if with_parts else None
    out = ops.layer_reduce(parts, pending, db)
    assert torch.equal(g_pre, want_pre) and torch.equal(db, want_db)
    if with_parts:
        assert torch.allclose(out, parts.sum(0), rtol=1e-6, atol=1e-6)
    else:
        assert out is None


def test_tuned_gemm_selections_load_on_a_matching_stack():
    """piml_amd.tuning.load() must accept the committed result file whenever the software stack is the one it
    was tuned on (a silent rejection doubles the step time: default GEMM selections, single stream)."""
    from piml_amd import tuning
    file_vals = {}
    for ln in open(tuning.DEFAULT_FILE):
        f = ln.strip().split(',')
        if f[0] == 'Validator':
            file_vals[f[1]] = f[2]
    torch.cuda.tunable.enable(True)
    have = {k: v for k, v in torch.cuda.tunable.get_validators()}
    torch.cuda.tunable.enable(False)
    same_stack = all(have.get(k) == v for k, v in file_vals.items())
    ok = tuning.load()
    torch.cuda.tunable.enable(False)
    tuning.LOADED = False
    assert isinstance(ok, bool)
    if same_stack:
        assert ok, 'tuned GEMM selections were rejected on the stack they were tuned on'


def test_mlp_chain_deferred_last_bias():
    """Last layer as a plain GEMM, its bias added by the k-sum pass: same messages / pooled sums / gradients
    (incl. the deferred bias's own gradient) as the ordinary chain followed by scale_ksum."""
    from piml_amd import ops
    torch.manual_seed(5)
    lins = [torch.nn.Linear(a, b).to(DEV) for a, b in ((6, 128), (128, 128), (128, 128))]
    relus = (True, True, False)
    params = [t for lin in lins for t in (lin.weight, lin.bias)]
    x = rnd(2048, 6, 6, seed=1)
    wm, wp = rnd(2048, 6, 128, seed=2), rnd(2048, 128, seed=3)

    def run(defer):
        xa = x.clone().requires_grad_(True)
        h = ops.mlp_chain(xa, relus, *params, defer_last_bias=defer)
        m, p = ops.scale_ksum(h, 2.0, bias=lins[-1].bias if defer else None)
        g = torch.autograd.grad((m * wm).sum() + (p * wp).sum(), [xa] + params)
        return m.detach(), p.detach(), g
    m0, p0, g0 = run(False)
    m1, p1, g1 = run(True)
    assert torch.allclose(m1, m0, rtol=1e-5, atol=1e-5) and torch.allclose(p1, p0, rtol=1e-5, atol=1e-5)
    for a, b in zip(g1, g0):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 5e-5 * scale
    with pytest.raises(ValueError):
        ops.mlp_chain(x, (True, True, True), *params, defer_last_bias=True)


@pytest.mark.parametrize('use', ['pooled', 'both'])
@pytest.mark.parametrize('rows,k,width', [(2048, 6, 128), (513, 10, 64), (3, 6, 128)])
def test_encoder_pool_matches_chain_plus_ksum(use, rows, k, width):
    """encoder -> 2x -> k-sum as one node (the ksum backward also yields the last bias gradient's first stage)."""
    from piml_amd import ops
    torch.manual_seed(7)
    lins = [torch.nn.Linear(a, b).to(DEV) for a, b in ((6, width), (width, width), (width, width))]
    relus = (True, True, False)
    params = [t for lin in lins for t in (lin.weight, lin.bias)]
    x = rnd(rows, k, 6, seed=1)
    wm, wp = rnd(rows, k, width, seed=2), rnd(rows, width, seed=3)
    loss = lambda m, p: (p * wp).sum() + ((m * wm).sum() if use == 'both' else 0)
    xa = x.clone().requires_grad_(True)
    m0, p0 = ops.scale_ksum(ops.mlp_chain(xa, relus, *params), 2.0)
    g0 = torch.autograd.grad(loss(m0, p0), [xa] + params)
    xb = x.clone().requires_grad_(True)
    m1, p1 = ops.encoder_pool(xb, relus, 2.0, *params)
    g1 = torch.autograd.grad(loss(m1, p1), [xb] + params)
    assert torch.allclose(m1, m0, rtol=1e-5, atol=1e-5) and torch.allclose(p1, p0, rtol=1e-5, atol=1e-5)
    for a, b in zip(g1, g0):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 5e-5 * scale, (float((a - b).abs().max()), scale)
    with torch.no_grad():
        m2, p2 = ops.encoder_pool(x, relus, 2.0, *params)
    assert torch.allclose(m2, m0, rtol=1e-5, atol=1e-5) and torch.allclose(p2, p0, rtol=1e-5, atol=1e-5)
