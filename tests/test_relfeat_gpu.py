"""GPU parity tests of the relative-feature kernels, called through the C ABI
(piml_amd.ops -> libpiml_hip.so), against (a) the golden vectors captured from the
reference and (b) the CPU oracle on seeded synthetic scenes."""
import numpy as np
import pytest
import torch

from conftest import bits, golden, golden_names
from piml_amd.scenes import synthetic_gc_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(x):
    return torch.tensor(np.asarray(x), device=DEV)


@pytest.mark.parametrize('name', golden_names('relfeat_'))
def test_relfeat_matches_reference_golden(oracle, name):
    from piml_amd.pedestrians import Pedestrians
    g = golden(name)
    kp, ang_p, dp, ko, ang_o, do = g['params']
    v, a = dev(g['velocity']), dev(g['acceleration'])
    pf, of, df = Pedestrians().get_relative_features(
        dev(g['position']), v, a, dev(g['destination']), dev(g['obstacles']),
        int(kp), ang_p, dp, int(ko), ang_o, do)
    # bit exact against the reference's own outputs
    assert np.array_equal(bits(pf.cpu().numpy()), bits(g['ped_features']))
    assert np.array_equal(bits(of.cpu().numpy()), bits(g['obs_features']))
    assert np.array_equal(bits(df.cpu().numpy()), bits(g['dest_features']))
    # in-place NaN -> 0 side effect on the caller's tensors (data.py:483-484)
    assert np.array_equal(bits(v.cpu().numpy()), bits(g['velocity_after']))
    assert np.array_equal(bits(a.cpu().numpy()), bits(g['acceleration_after']))


@pytest.mark.parametrize('name', golden_names('relfeat_'))
def test_heading_matches_reference_golden(name):
    from piml_amd import ops
    g = golden(name)
    hd = ops.heading_direction(dev(np.nan_to_num(g['velocity'], nan=0.0)))
    assert np.array_equal(bits(hd.cpu().numpy()), bits(g['heading']))


@pytest.mark.parametrize('N,M,seed,C', [(1024, 100, 0, None), (1024, 2000, 1, None), (4096, 2000, 0, None),
                                         (300, 100, 2, 5), (5000, 4500, 3, None), (1, 0, 0, None),
                                         (63, 3, 4, 2), (4097, 1, 5, None)])
def test_relfeat_matches_oracle_synthetic(oracle, N, M, seed, C):
    """Indices, distances' order and features bit-exact against the oracle (same tie rule),
    including multi-tile sizes (> 4096 sources) and ragged tails."""
    from piml_amd import ops
    sc = synthetic_gc_scene(N, M, seed=seed, channels=C, nan_frac=0.03 if N > 1 else 0.0)
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(sc['position'].shape) * 0.3).astype(np.float32)
    args = (sc['position'], sc['velocity'], a, sc['destination'], sc['obstacles'])
    ref = oracle.relfeat_fwd(*[x[..., None, :, :] if i < 4 else x for i, x in enumerate(args)], return_index=True)
    out = ops.relative_features(*[dev(x) for x in args], return_index=True)
    for got, want in zip(out, ref[:5]):
        got = got.cpu().numpy()
        want = want.reshape(got.shape)          # the oracle carries an explicit t = 1 axis
        if got.dtype == np.int32:
            assert np.array_equal(got, want)
        else:
            assert np.array_equal(bits(got), bits(want))


def test_relfeat_focal_block_equals_full(oracle):
    """Agent-block sharding: a focal block against all sources equals the same rows of the
    full call."""
    from piml_amd import ops
    sc = synthetic_gc_scene(1000, 300, seed=11)
    args = [dev(sc[k]) for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    full = ops.relative_features(*args, return_index=True)
    for f0, fc in ((0, 250), (250, 500), (750, 250), (999, 1)):
        part = ops.relative_features(*args, focal_begin=f0, focal_count=fc, return_index=True)
        for a_, b_ in zip(part, full):
            assert torch.equal(a_, b_[f0:f0 + fc])


@pytest.mark.parametrize('N,M,C', [(512, 100, None), (2048, 2000, None), (200, 100, 3), (4096, 2000, None)])
def test_relfeat_backward_matches_oracle(oracle, N, M, C):
    from piml_amd import ops
    sc = synthetic_gc_scene(N, M, seed=21, channels=C)
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(sc['position'].shape) * 0.3).astype(np.float32)
    p, v, a_, d = [dev(x).requires_grad_(True) for x in (sc['position'], sc['velocity'], a, sc['destination'])]
    pf, of, df, pi, oi = ops.relative_features(p, v, a_, d, dev(sc['obstacles']), return_index=True)
    gp_, go_, gd_ = [torch.tensor(rng.standard_normal(t.shape).astype(np.float32), device=DEV) for t in (pf, of, df)]
    (pf * gp_).sum().add((of * go_).sum()).add((df * gd_).sum()).backward()
    want = oracle.relfeat_bwd(gp_.cpu().numpy(), go_.cpu().numpy(), gd_.cpu().numpy(), pi.cpu().numpy(),
                              oi.cpu().numpy(), sc['position'], sc['destination'])
    worst = 0.0
    for got, w in zip((p.grad, v.grad, a_.grad, d.grad), want):
        got = got.cpu().numpy()
        scale = max(1.0, np.abs(w).max())
        worst = max(worst, np.abs(got - w).max() / scale)
        assert np.abs(got - w).max() <= 1e-5 * scale
    print(f'relfeat backward N={N} M={M}: max rel err vs oracle {worst:.2e} (bar 1e-5)')
    # absent agents receive no gradient
    absent = np.isnan(sc['position'][..., 0])
    assert np.all(p.grad.cpu().numpy()[absent] == 0)


def test_relfeat_requires_gpu_tensors():
    from piml_amd import ops, _lib
    sc = synthetic_gc_scene(8, 0, seed=0)
    with pytest.raises(_lib.PimlHipError):
        ops.relative_features(*[torch.tensor(sc[k]) for k in
                                ('position', 'velocity', 'acceleration', 'destination', 'obstacles')])


def test_selection_arithmetic_is_bit_exact():
    """The distance / cosine primitives of the selection predicates, on 4M random pairs,
    against numpy restatements of the formulas pinned on PyTorch's CPU kernels."""
    from piml_amd import _lib
    n = 1 << 22
    rng = np.random.default_rng(0)
    r = ((rng.random((n, 2)) - 0.5) * 60).astype(np.float32)
    r[::5] *= np.float32(1e-3)
    h = rng.standard_normal((n, 2)).astype(np.float32)
    h[::7] = 0
    t = [dev(np.ascontiguousarray(x)) for x in (r[:, 0], r[:, 1], h[:, 0], h[:, 1])]
    dist, cosv = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    _lib.check(_lib.lib().piml_probe_arith(*[x.data_ptr() for x in t], dist.data_ptr(), cosv.data_ptr(), n,
                                           torch.cuda.current_stream().cuda_stream), 'probe')
    f64, f32 = np.float64, np.float32

    def norm(x, y):   # sqrt(fma(y, y, x*x)); the f64 detour is exact for the fma
        return np.sqrt(((x * x).astype(f32).astype(f64) + y.astype(f64) * y.astype(f64)).astype(f32))
    d = norm(r[:, 0], r[:, 1])
    n1 = np.maximum(d, f32(1e-8))
    n2 = np.maximum(norm(h[:, 0], h[:, 1]), f32(1e-8))
    with np.errstate(invalid='ignore'):
        c = ((r[:, 0] / n1) * (h[:, 0] / n2)).astype(f32) + ((r[:, 1] / n1) * (h[:, 1] / n2)).astype(f32)
    assert np.array_equal(bits(dist.cpu().numpy()), bits(d))
    assert np.array_equal(bits(cosv.cpu().numpy()), bits(c.astype(f32)))


def test_relfeat_packed_state_equals_separate(oracle):
    """Interleaved (N,6) records + focal-row destinations (the sharded layout) give the same
    features and gradients as three separate arrays."""
    from piml_amd import ops
    sc = synthetic_gc_scene(700, 300, seed=13)
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((700, 2)) * 0.3).astype(np.float32)
    p, v, a_, d = [dev(x).requires_grad_(True) for x in (sc['position'], sc['velocity'], a, sc['destination'])]
    obs = dev(sc['obstacles'])
    f0, fc = 200, 300
    ref = ops.relative_features(p, v, a_, d, obs, focal_begin=f0, focal_count=fc, return_index=True)
    state = torch.cat((p, v, a_), dim=-1).detach().requires_grad_(True)
    drow = d[f0:f0 + fc].detach().requires_grad_(True)
    out = ops.relative_features_packed(state, drow, obs, f0, fc, return_index=True)
    for x, y in zip(out, ref):
        assert torch.equal(x, y)
    w = [torch.randn_like(t) for t in out[:3]]
    sum((x * y).sum() for x, y in zip(out[:3], w)).backward()
    sum((x * y).sum() for x, y in zip(ref[:3], w)).backward()
    want = torch.cat((p.grad, v.grad, a_.grad), dim=-1)
    assert (state.grad - want).abs().max() <= 1e-5 * max(1.0, want.abs().max().item())
    assert (drow.grad - d.grad[f0:f0 + fc]).abs().max() <= 1e-6


def test_relfeat_full_size_cfg4_properties_and_oracle(oracle):
    """BASELINE.json configs[3] size (16384 agents + 2000 obstacle points, 5 LDS tiles): structural
    properties that need no reference, then the full bit-exact comparison with the oracle."""
    from piml_amd import ops
    N, M = 16384, 2000
    sc = synthetic_gc_scene(N, M, seed=9)
    args = (sc['position'], sc['velocity'], sc['acceleration'], sc['destination'], sc['obstacles'])
    pf, of, df, pi, oi = [t.cpu().numpy() for t in ops.relative_features(*[dev(x) for x in args], return_index=True)]
    p, o = sc['position'], sc['obstacles']
    for feat, idx, src, thr in ((pf, pi, p, 4.0), (of, oi, o, 4.0)):
        live = idx >= 0
        d = np.linalg.norm(feat[..., :2], axis=-1)
        assert np.all(d[live] <= thr + 1e-6)                                   # within the attention distance
        assert np.all(np.diff(np.where(live, d, 1e30), axis=-1) >= -1e-6)      # slots ascend by distance
        assert np.all(~live | (np.cumsum(~live, axis=-1) == 0))                # empty slots only at the tail
        assert np.all(feat[~live] == 0)                                        # zero padding
        rows = np.nonzero(live)
        assert np.allclose(src[idx[live]] - p[rows[0]], feat[live][:, :2], atol=0)   # gathered = source - focal
    absent = np.isnan(p[:, 0])
    assert np.all(pi[absent] == -1) and np.all(oi[absent] == -1) and np.all(df[absent] == 0)
    ref = oracle.relfeat_fwd(*[x[None] for x in args[:4]], args[4], return_index=True)
    for got, want in zip((pf, of, df, pi, oi), ref[:5]):
        want = want.reshape(got.shape)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_relfeat_cfg4_sharded_blocks_forward_and_backward(oracle):
    """BASELINE.json configs[3] in the form the 8 ranks actually launch it: 16384 agents + 2000 obstacle points,
    rank r's launch = focal rows [r*2048, (r+1)*2048) of the interleaved (N, 6) record buffer (state_ld = 6) that the
    per-step all-gather produces.  All 8 blocks: forward bit-exact against the oracle's rows, and the SUM of the
    blocks' partial d/d(state) (what the reduce-scatter adds up) against the oracle's backward of the whole scene."""
    from piml_amd import ops
    N, M, G = 16384, 2000, 8
    n = N // G
    sc = synthetic_gc_scene(N, M, seed=4)
    rng = np.random.default_rng(40)
    acc = (rng.standard_normal((N, 2)) * 0.3).astype(np.float32)
    args = (sc['position'], sc['velocity'], acc, sc['destination'], sc['obstacles'])
    ref = oracle.relfeat_fwd(*[x[None] for x in args[:4]], args[4], return_index=True)
    state = dev(np.concatenate(args[:3], axis=-1)).requires_grad_(True)
    dest, obs = dev(args[3]), dev(args[4])
    gw = [rng.standard_normal(ref[i][0].shape).astype(np.float32) for i in range(3)]
    g_dest_rows = []
    for r in range(G):
        rows = slice(r * n, (r + 1) * n)
        d_rows = dest[rows].clone().requires_grad_(True)
        out = ops.relative_features_packed(state, d_rows, obs, r * n, n, return_index=True)
        for got, want in zip(out, ref[:5]):
            got = got.detach().cpu().numpy()
            want = want[0][rows].reshape(got.shape)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f'block {r}'
        sum((o * dev(w[rows])).sum() for o, w in zip(out[:3], gw)).backward()     # accumulates into state.grad
        g_dest_rows.append(d_rows.grad)
    want = oracle.relfeat_bwd(gw[0], gw[1], gw[2], ref[3][0], ref[4][0], sc['position'], sc['destination'])
    got_state = state.grad.cpu().numpy()
    worst = 0.0
    for got, w in zip((got_state[:, 0:2], got_state[:, 2:4], got_state[:, 4:6],
                       torch.cat(g_dest_rows).cpu().numpy()), want):
        scale = max(1.0, np.abs(w).max())
        worst = max(worst, np.abs(got - w).max() / scale)
    print(f'cfg4 sharded blocks: summed d/d(state) vs oracle max rel err {worst:.2e} (bar 1e-5)')
    assert worst <= 1e-5


def test_relfeat_fuzz_small_scenes_vs_oracle(oracle):
    """200 random small scenes with random k / sight angles / thresholds, NaN agents, standing agents
    and LATTICE positions (many exact distance ties, exercising the lowest-index tie rule), all slices
    of a batch in one launch: features, indices bit-exact against the oracle."""
    from piml_amd import ops
    import os
    rng = np.random.default_rng(int(os.environ.get('PIML_FUZZ_SEED', '2024')))
    for case in range(200):
        N = int(rng.integers(1, 70))
        M = int(rng.choice([0, 1, 3, 17, 64, 130]))
        C = int(rng.choice([1, 1, 2, 5]))
        lattice = case % 3 == 0
        if lattice:
            p = rng.integers(0, 6, size=(C, N, 2)).astype(np.float32) * 0.5
            o = rng.integers(0, 6, size=(max(M, 1), 2)).astype(np.float32) * 0.5 + 0.25
        else:
            p = (rng.random((C, N, 2)) * 6).astype(np.float32)
            o = (rng.random((max(M, 1), 2)) * 6).astype(np.float32)
        o = o[:M] if M else np.zeros((0, 2), np.float32)
        v = rng.standard_normal((C, N, 2)).astype(np.float32)
        if lattice:
            v = np.round(v)                        # axis-aligned / zero headings: cos exactly on thresholds
        v[rng.random((C, N)) < 0.15] = 0
        a = rng.standard_normal((C, N, 2)).astype(np.float32)
        d = (rng.random((C, N, 2)) * 6).astype(np.float32)
        absent = rng.random((C, N)) < 0.15
        p[absent] = np.nan
        d[absent] = np.nan
        if case % 7 == 0:
            a[rng.random((C, N)) < 0.2] = np.nan
        kw = dict(topk_ped=int(rng.integers(0, 9)), topk_obs=int(rng.integers(0, 13)),
                  sight_angle_ped=float(rng.choice([0, 45, 90, 100, 180, 270])),
                  sight_angle_obs=float(rng.choice([30, 90, 120, 180])),
                  dist_threshold_ped=float(rng.choice([0, 0.5, 1.0, 2.5, 4, 100])),
                  dist_threshold_obs=float(rng.choice([0.5, 1.5, 4, 100])))
        if M == 0:
            o_oracle = np.zeros((0, 2), np.float32)
        ref = oracle.relfeat_fwd(p[:, None], v[:, None], a[:, None], d[:, None], o, return_index=True, **kw)
        out = ops.relative_features(dev(p), dev(v), dev(a), dev(d), dev(o), return_index=True, **kw)
        for got, want in zip(out, ref[:5]):
            got = got.cpu().numpy()
            want = want.reshape(got.shape)
            if got.dtype == np.int32:
                assert np.array_equal(got, want), (case, N, M, C, kw)
            else:
                assert np.array_equal(bits(got), bits(want)), (case, N, M, C, kw)


@pytest.mark.parametrize('N,M,C,packed', [(1500, 300, None, False), (300, 100, 3, False), (2048, 500, None, True)])
def test_relfeat_backward_deterministic_variant(oracle, N, M, C, packed):
    """PIML_DETERMINISTIC_BWD / ops.DETERMINISTIC_BWD: the gradient without float atomics (entries sorted by source,
    gathered in a fixed order): bit-identical between runs, equal to the atomic kernel up to summation order, and
    within 1e-5 of the oracle."""
    from piml_amd import ops
    sc = synthetic_gc_scene(N, M, seed=31, channels=C)
    rng = np.random.default_rng(7)
    acc = (rng.standard_normal(sc['position'].shape) * 0.3).astype(np.float32)
    obs = dev(sc['obstacles'])

    def run():
        if packed:
            state = dev(np.concatenate((sc['position'], sc['velocity'], acc), -1)).requires_grad_(True)
            d = dev(sc['destination']).requires_grad_(True)
            out = ops.relative_features_packed_self(state, d, obs, dev(sc['desired_speed']), 0, N, return_index=True)
            leaves = (state, d)
        else:
            leaves = tuple(dev(x).requires_grad_(True) for x in (sc['position'], sc['velocity'], acc, sc['destination']))
            out = ops.relative_features(*leaves, obs, return_index=True)
        g = torch.Generator().manual_seed(3)
        w = [torch.randn(t.shape, generator=g).to(DEV) for t in out[:3]]
        sum((o * x).sum() for o, x in zip(out[:3], w)).backward()
        return [t.grad.clone() for t in leaves], out, w
    try:
        ops.DETERMINISTIC_BWD = True
        g1, out, w = run()
        g2, _, _ = run()
        for a, b in zip(g1, g2):
            assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))          # bit-reproducible
        ops.DETERMINISTIC_BWD = False
        g0, _, _ = run()
    finally:
        ops.DETERMINISTIC_BWD = False
    worst = 0.0
    for a, b in zip(g1, g0):
        a, b = torch.nan_to_num(a), torch.nan_to_num(b)
        worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1.0)))
    print(f'deterministic vs atomic relfeat backward N={N}: max rel diff {worst:.1e}')
    assert worst <= 1e-6
    if not packed:
        want = oracle.relfeat_bwd(w[0].cpu().numpy(), w[1].cpu().numpy(), w[2].cpu().numpy(), out[3].cpu().numpy(),
                                  out[4].cpu().numpy(), sc['position'], sc['destination'])
        for got, x in zip(g1, want):
            assert np.abs(np.nan_to_num(got.cpu().numpy()) - x).max() <= 1e-5 * max(1.0, np.abs(x).max())


@pytest.mark.parametrize('lead', [(), (4,)])
def test_relative_features_self_equals_features_plus_cat(lead):
    """ops.relative_features_self (piml_relfeat_fwd_self / bwd_self): the per-frame torch.cat((dest_features, v, a, v0)) of the
    training rollout (src/models/simulators.py:778-779) inside the launch -- features bit-equal, gradients equal to the float
    atomics' order."""
    import numpy as np
    import torch
    from piml_amd import ops
    from piml_amd.scenes import synthetic_gc_scene
    N, M = 122, 100
    sc = synthetic_gc_scene(N, M, seed=5, channels=lead[0] if lead else None)
    p, v, a, d = [torch.tensor(sc[k], device='cuda').requires_grad_(True) for k in ('position', 'velocity', 'acceleration', 'destination')]
    v0 = torch.tensor(sc['desired_speed'], device='cuda').requires_grad_(True)
    o = torch.tensor(sc['obstacles'], device='cuda')
    pf, of, df = ops.relative_features(p, v, a, d, o)
    sf_want = torch.cat((df, v, a, v0), dim=-1)
    pf2, of2, sf = ops.relative_features_self(p, v, a, d, o, v0)
    assert torch.equal(pf, pf2) and torch.equal(of, of2)
    assert torch.equal(torch.nan_to_num(sf), torch.nan_to_num(sf_want)) and torch.equal(sf.isnan(), sf_want.isnan())
    g = torch.Generator().manual_seed(1)
    w1, w2, w3 = [torch.randn(t.shape, generator=g).cuda() for t in (pf, of, sf)]
    finite = ~sf_want.isnan()

    def grads(outs):
        loss = (outs[0] * w1).sum() + (outs[1] * w2).sum() + torch.where(finite, outs[2] * w3, torch.zeros_like(w3)).sum()
        return torch.autograd.grad(loss, [p, v, a, d, v0], allow_unused=True)
    want = grads((pf, of, sf_want))
    got = grads((pf2, of2, sf))
    for x, y in zip(want, got):
        assert torch.allclose(torch.nan_to_num(x), torch.nan_to_num(y), rtol=1e-5, atol=1e-6)
