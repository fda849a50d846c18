"""GPU parity tests (through the C ABI) of MLAPM.step fwd/bwd, collision detection / counts /
labels and calc_acceleration against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import golden
from piml_amd.scenes import synthetic_gc_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REL = 1e-5      # north-star tolerance on float32 forces


def dev(x):
    return torch.tensor(np.asarray(x), device=DEV)


def rel_err(got, ref, floor):
    return (np.linalg.norm(got - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), floor)).max()


def mlapm_model(g, ver):
    from piml_amd.models.mlapm import MLAPM
    tau, A, B, C, D, theta = g[f'{ver}_params']
    return MLAPM(version=ver, tau=tau, A=A, B=B, C=C, D=D, theta=theta)


@pytest.mark.parametrize('ver', ['raw', 'GC', 'UCY'])
@pytest.mark.parametrize('N', [7, 64, 1024])
def test_mlapm_step_matches_reference(ver, N):
    g = golden('mlapm')
    k = f'{ver}_N{N}'
    act = mlapm_model(g, ver).step(dev(g[k + '_p']), dev(g[k + '_v']), dev(g[k + '_v0']), dev(g[k + '_dest']),
                                   dt=0.08, radius=0.3)
    assert rel_err(act.cpu().numpy(), g[k + '_action'], 1e-3) < REL


@pytest.mark.parametrize('ver', ['raw', 'GC', 'UCY'])
@pytest.mark.parametrize('N', [7, 64, 1024])
def test_mlapm_backward_matches_reference_autograd(ver, N):
    g = golden('mlapm')
    k = f'{ver}_N{N}'
    p, v, v0, d = [dev(g[k + s]).requires_grad_(True) for s in ('_p', '_v', '_v0', '_dest')]
    act = mlapm_model(g, ver).step(p, v, v0, d, dt=0.08, radius=0.3)
    (act * dev(g[k + '_w'])).sum().backward()
    for got, name in ((p.grad, 'gp'), (v.grad, 'gv'), (v0.grad, 'gv0'), (d.grad, 'gdest')):
        ref = g[f'{k}_{name}']
        got = got.cpu().numpy().reshape(ref.shape)
        # gradients of a sum of ~N terms: tolerance relative to the largest entry
        assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), (name, np.abs(got - ref).max())


@pytest.mark.parametrize('ver', ['raw', 'GC', 'UCY'])
def test_mlapm_step_matches_oracle_4096(oracle, ver):
    """BASELINE configs[2]-sized scene against the oracle (double-accumulated pair sums)."""
    sc = synthetic_gc_scene(4096, 0, seed=3, nan_frac=0.0)
    pr = dict(raw=dict(tau=0.5, A=7.55, B=-3.0), GC=dict(tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56),
              UCY=dict(tau=5 / 6, A=10.67, B=-3.33, C=0.5, theta=20))[ver]
    from piml_amd.models.mlapm import MLAPM
    act = MLAPM(version=ver, **pr).step(dev(sc['position']), dev(sc['velocity']), dev(sc['desired_speed']),
                                        dev(sc['destination']), dt=0.08, radius=0.3)
    ref = oracle.mlapm_step(sc['position'], sc['velocity'], sc['desired_speed'], sc['destination'], 0.08, 0.3,
                            version=ver, **pr)
    got = act.cpu().numpy()
    err = np.linalg.norm(got - ref, axis=-1) / np.maximum(np.linalg.norm(ref, axis=-1), 1e-3)
    # UCY's collision flag is a hard threshold (mlapm.py:43-47): the kernel evaluates it with exactly the reference's
    # float32 operations (ucy_collision in pairwise.hip), so no pair may flip -- no outlier allowance
    print(f'MLAPM {ver} N=4096: max rel err vs oracle {err.max():.2e} (bar {REL:g})')
    assert err.max() < REL, err.max()


def test_mlapm_demo_trajectory():
    """main_mlapm.py:18-36, first 10 of the 200 Euler steps (the system is chaotic later)."""
    g = golden('mlapm')
    m = mlapm_model(g, 'GC')
    p, v = dev(g['GC_N7_p']), dev(g['GC_N7_v'])
    v0, d = dev(g['GC_N7_v0']), dev(g['GC_N7_dest'])
    for t in range(10):
        v = m.step(p, v, v0, d, dt=0.08, radius=0.3)
        p = p + v * 0.08
        assert np.abs(p.cpu().numpy() - g['demo_traj'][t + 1]).max() < 1e-4


def test_mlapm_nan_poisons_like_reference():
    from piml_amd.models.mlapm import MLAPM
    sc = synthetic_gc_scene(64, 0, seed=1, nan_frac=0.0)
    p = sc['position'].copy()
    p[5] = np.nan
    act = MLAPM(version='raw', tau=0.5, A=7.55, B=-3.0).step(dev(p), dev(sc['velocity']), dev(sc['desired_speed']),
                                                             dev(sc['destination']), dt=0.08)
    assert torch.isnan(act).all()


_LAWS = dict(raw=dict(tau=0.5, A=7.55, B=-3.0), GC=dict(tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56),
             UCY=dict(tau=5 / 6, A=10.67, B=-3.33, C=0.5, theta=20))


def _mlapm_bwd_direct(entry, sc, w, ver, workspace=None):
    """piml_mlapm_step_bwd / piml_mlapm_step_bwd_ws through the C ABI on the same inputs"""
    from piml_amd import _lib, ops
    pr = dict(C=0.0, D=0.0, theta=0.0)
    pr.update(_LAWS[ver])
    p, v, v0, d = [dev(sc[k]) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    N = p.shape[0]
    out = [torch.full((N, 2), float('nan'), device=DEV), torch.full((N, 2), float('nan'), device=DEV),
           torch.full((N,), float('nan'), device=DEV), torch.full((N, 2), float('nan'), device=DEV)]
    ptr = lambda t: t.data_ptr()
    args = [ptr(w), ptr(p), ptr(v), ptr(v0), ptr(d), N, ops.MLAPM_VARIANTS[ver], pr['tau'], pr['A'], pr['B'], pr['C'], pr['D'],
            pr['theta'], 0.3, 0.08, ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3])]
    L = _lib.lib()
    if entry == 'ws':
        need = int(L.piml_mlapm_bwd_workspace_floats(N, ops.MLAPM_VARIANTS[ver]))
        assert need > 0
        ws = torch.full((need,), float('nan'), device=DEV) if workspace is None else workspace
        rc = L.piml_mlapm_step_bwd_ws(*args, ptr(ws), ws.numel(), None)
    else:
        rc = L.piml_mlapm_step_bwd(*args, None)
    torch.cuda.synchronize()
    return rc, [o.cpu().numpy() for o in out]


@pytest.mark.parametrize('ver', ['raw', 'GC', 'UCY'])
@pytest.mark.parametrize('N', [512, 1000, 2048, 2111, 4096, 9000])
def test_mlapm_backward_once_per_pair_matches_two_role_kernel(ver, N):
    """The rotating-focal-agent backward (every ordered pair once, partial rows, fixed order) against the kernel that
    evaluates both roles in the owning wavefront: the same pair terms, sums in a different order.  (UCY: the flag-off terms
    once per pair, the flagged pairs' differences from a wavefront per agent -- the exact predicate decides in both.)"""
    sc = synthetic_gc_scene(N, 0, seed=11, nan_frac=0.0)
    w = torch.randn(N, 2, device=DEV, generator=torch.Generator(DEV).manual_seed(N))
    rc0, ref = _mlapm_bwd_direct('two_role', sc, w, ver)
    rc1, got = _mlapm_bwd_direct('ws', sc, w, ver)
    assert rc0 == 0 and rc1 == 0
    for g, r, name in zip(got, ref, ('gp', 'gv', 'gv0', 'gdest')):
        assert np.isfinite(g).all(), name
        assert np.abs(g - r).max() <= 5e-6 * max(np.abs(r).max(), 1e-3), (name, np.abs(g - r).max(), np.abs(r).max())
    rc2, again = _mlapm_bwd_direct('ws', sc, w, ver)
    assert all(np.array_equal(a, b) for a, b in zip(got, again))       # fixed order: bitwise repeatable


@pytest.mark.parametrize('ver', ['raw', 'GC'])
def test_mlapm_backward_float64_large(ver):
    """ops.mlapm_step's gradient at a size the once-per-pair backward serves (N >= 512; ragged blocks) against autograd
    through a float64 statement of mlapm.py:10-41.  (The reference's own autograd pins the same path at N = 1024:
    test_mlapm_backward_matches_reference_autograd.)"""
    from piml_amd import ops
    N = 2111
    pr = dict(C=0.0, D=0.0, theta=0.0)
    pr.update(_LAWS[ver])
    sc = synthetic_gc_scene(N, 0, seed=5, nan_frac=0.0)
    w = torch.randn(N, 2, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    leaves = [dev(sc[k]).requires_grad_(True) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    act = ops.mlapm_step(*leaves, 0.08, 0.3, version=ver, **_LAWS[ver])
    got = torch.autograd.grad(act, leaves, w)
    p, v, v0, d = [dev(sc[k]).double().requires_grad_(True) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    e = torch.nn.functional.normalize(d - p, dim=-1)
    force = (v0.reshape(-1, 1) * e - v) / pr['tau']
    vr = p[None, :, :] - p[:, None, :]                      # [focal, source]
    r = vr.norm(dim=-1, keepdim=True)
    n = torch.nn.functional.normalize(vr, dim=-1)
    view = ((v[:, None, :] * vr).sum(-1, keepdim=True) > 0).double()
    if ver == 'raw':
        term = view * pr['A'] * (pr['B'] * r).exp() * n
    else:
        vv = v[None, :, :] - v[:, None, :]
        cs = torch.nn.functional.cosine_similarity(vr, vv, dim=-1).unsqueeze(-1)
        cr = vr[..., 0] * e[:, None, 1] - vr[..., 1] * e[:, None, 0]
        th = torch.where(cr > 0, -1.0, 1.0) * (pr['theta'] / 180 * np.pi)       # -sign(cr) theta, 0 -> +theta
        c, s_ = th.cos(), th.sin()
        direc = torch.stack([c * n[..., 0] - s_ * n[..., 1], s_ * n[..., 0] + c * n[..., 1]], dim=-1)
        term = view * pr['A'] * (pr['B'] * r + pr['C'] * cs + pr['D'] * r * cs).exp() * direc
    act64 = v + (force - term.sum(dim=1)) * 0.08
    assert rel_err(act.detach().cpu().numpy(), act64.detach().cpu().numpy(), 1e-3) < REL
    ref = torch.autograd.grad(act64, [p, v, v0, d], w.double())
    for g, r_, name in zip(got, ref, ('gp', 'gv', 'gv0', 'gdest')):
        g, r_ = g.cpu().numpy().reshape(-1), r_.cpu().numpy().reshape(-1)
        assert np.abs(g - r_).max() <= 2e-5 * max(np.abs(r_).max(), 1e-3), (name, np.abs(g - r_).max(), np.abs(r_).max())


def test_mlapm_backward_workspace_contract():
    from piml_amd import _lib
    L = _lib.lib()
    assert L.piml_mlapm_bwd_workspace_floats(256, 1) == 0           # small scenes keep the two-role kernel
    assert L.piml_mlapm_bwd_workspace_floats(4096, 2) > 0
    assert L.piml_mlapm_bwd_workspace_floats(4096, 1) > 0
    sc = synthetic_gc_scene(2048, 0, seed=2, nan_frac=0.0)
    w = torch.ones(2048, 2, device=DEV)
    rc, _ = _mlapm_bwd_direct('ws', sc, w, 'GC', workspace=torch.empty(1024, device=DEV))
    assert rc != 0                                                  # a workspace that is too small is an error, not a fallback


def test_collision_detection_matches_reference():
    from piml_amd.pedestrians import Pedestrians as P
    g = golden('collision_gc')
    for thr in (0.5, 0.25, 1.5):
        assert np.array_equal(P.collision_detection(dev(g['p3']), thr).cpu().numpy(), g[f'coll3_thr{thr}'])
        assert np.array_equal(P.collision_detection(dev(g['pc']), thr).cpu().numpy(), g[f'collc_thr{thr}'])
    for thr in (0.5, 1.5):
        assert np.array_equal(P.collision_detection(dev(g['p4']), thr).cpu().numpy(), g[f'coll4_thr{thr}'])
        got = P.collision_detection(dev(g['p3']) + 0.05, thr, real_position=dev(g['p3']))
        assert np.array_equal(got.cpu().numpy(), g[f'coll3_real_thr{thr}'])
    s = golden('collision_syn')
    for thr in (0.5, 0.25):
        assert np.array_equal(P.collision_detection(dev(s['pc']), thr).cpu().numpy(), s[f'collc_thr{thr}'])


def test_collision_counts_match_matrix_sums():
    from piml_amd import ops
    g = golden('collision_gc')
    for key in ('p3', 'pc'):
        thr = (0.5, 0.25, 1.5)
        counts = ops.collision_counts(dev(g[key]), thr).cpu().numpy()
        tag = 'coll3' if key == 'p3' else 'collc'
        for h, t in enumerate(thr):
            assert np.array_equal(counts[h], g[f'{tag}_thr{t}'].sum(-1).astype(np.float32))
    s = golden('collision_syn')
    counts = ops.collision_counts(dev(s['pc']), (0.5, 0.25)).cpu().numpy()
    for h, t in enumerate((0.5, 0.25)):
        assert np.array_equal(counts[h], s[f'collc_thr{t}'].sum(-1).astype(np.float32))


def test_collision_label_matches_reference():
    from piml_amd.pedestrians import Pedestrians as P
    g = golden('collision_label')
    assert np.array_equal(P.calculate_collision_label(dev(g['feat_real'])).cpu().numpy(), g['label_real'])
    assert np.array_equal(P.calculate_collision_label(dev(g['feat_rnd'])).cpu().numpy(), g['label_rnd'])


@pytest.mark.parametrize('ver,ds', [('v0', 'gc1560'), ('v0', 'ucy'), ('v1', 'ucy'), ('v2', 'gc2344')])
def test_calc_acceleration_matches_reference(ver, ds):
    from piml_amd.utils.utils import calc_acceleration
    g = golden('calcacc')
    for tag, feat in (('real', g['feat_real'][0]), ('rnd', g['feat_rnd'])):
        out = calc_acceleration(dev(feat), ver, ds).cpu().numpy()
        ref = g[f'{tag}_{ver}_{ds}']
        assert np.allclose(out, ref, rtol=1e-5, atol=1e-6), np.abs(out - ref).max()


def test_collision_counts_fast_path_equals_general_and_oracle(oracle):
    """S <= 25 takes the streaming kernel, S > 25 the general one (friends rule active): both against
    the oracle's collision_detection(...).sum(-1), incl. a multi-tile N and thresholds on exact distances."""
    from piml_amd import ops
    rng = np.random.default_rng(5)
    for S, N in ((4, 700), (1, 5000), (25, 130), (26, 130), (40, 90)):
        p = (rng.integers(0, 40, size=(S, N, 2)) * 0.25).astype(np.float32)        # lattice: d == thr happens
        p[rng.random((S, N)) < 0.1] = np.nan
        thr = (0.5, 0.25, 1.0)
        got = ops.collision_counts(dev(p), thr).cpu().numpy()
        for h, t in enumerate(thr):
            want = oracle.collision_detection(p, t).sum(-1)
            assert np.array_equal(got[h], want), (S, N, t)


def test_mlapm_rollout_matches_host_compaction_loop(oracle):
    """The device-side simulation loop (absent agents kept as NaN rows, one captured step replayed)
    against the reference's loop structure (src/main_mlapm.py:18-36: compact the active agents on the
    host, step, Euler, arrival mask) driven by the oracle."""
    import time
    g = golden('mlapm')
    m = mlapm_model(g, 'GC')
    tau, A, B, C, D, theta = g['GC_params']
    p0, v0_, spd, dst = g['GC_N7_p'], g['GC_N7_v'], g['GC_N7_v0'], g['GC_N7_dest']
    steps, dt, radius = 200, 0.08, 0.3
    traj_p, traj_v = m.rollout(dev(p0), dev(v0_), dev(spd), dev(dst), dt, radius, steps)      # one launch per frame, 8 frames per graph
    eager_p, eager_v = m.rollout(dev(p0), dev(v0_), dev(spd), dev(dst), dt, radius, steps, use_graph=False)
    assert torch.equal(torch.nan_to_num(traj_p), torch.nan_to_num(eager_p)) and torch.equal(torch.nan_to_num(traj_v), torch.nan_to_num(eager_v))
    seq_p, seq_v = m.rollout(dev(p0), dev(v0_), dev(spd), dev(dst), dt, radius, steps, fused=False)   # MLAPM.step + torch glue per frame
    assert torch.equal(torch.isnan(traj_p), torch.isnan(seq_p))
    assert torch.equal(torch.nan_to_num(traj_p), torch.nan_to_num(seq_p)) and torch.equal(torch.nan_to_num(traj_v), torch.nan_to_num(seq_v))
    # host loop with compaction, as the reference does it
    N = p0.shape[0]
    p, v = p0.copy(), v0_.copy()
    mask = np.ones(N, bool)
    ref = np.full((steps + 1, N, 2), np.nan, np.float32)
    ref[0] = p
    for t in range(1, steps + 1):
        if not mask.any():
            break
        vn = oracle.mlapm_step(p[mask], v[mask], spd[mask], dst[mask], dt, radius, version='GC', tau=tau, A=A, B=B,
                               C=C, D=D, theta=theta)
        pn = p[mask] + vn * np.float32(dt)
        p[mask], v[mask] = pn, vn
        ref[t][mask] = pn
        mask &= ~(np.linalg.norm(p - dst, axis=-1) < radius)
    got = traj_p.cpu().numpy()
    assert np.array_equal(np.isnan(got[..., 0]), np.isnan(ref[..., 0]))       # same arrival frames
    assert np.isnan(got[-1]).all()                                            # everybody arrives within 200 steps
    assert np.nanmax(np.abs(got[:60] - ref[:60])) < 1e-4
    assert np.nanmax(np.abs(got - ref)) < 5e-3
    # throughput of the closed-form simulator at the bench scene size
    sc = synthetic_gc_scene(4096, 0, seed=0, nan_frac=0.02)
    args = [dev(sc[k]) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    m.rollout(*args, dt, radius, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.rollout(*args, dt, radius, 300)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f'MLAPM GC rollout N=4096: {300 / el:.0f} steps/s')


@pytest.mark.parametrize('ver', ['raw', 'GC', 'UCY'])
@pytest.mark.parametrize('N', [200, 4100])
def test_mlapm_fused_rollout_equals_operator_sequence(ver, N):
    """A frame as ONE launch (state read from the trajectory, arrival test on the way in, device-side frame counter,
    eight frames per captured graph) against MLAPM.step + torch glue per frame: same arrival frames, same numbers."""
    from piml_amd.models.mlapm import MLAPM
    sc = synthetic_gc_scene(N, 0, seed=4, nan_frac=0.03)
    args = [dev(sc[k]) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    args[3] = args[0] + (args[3] - args[0]) * 0.02          # destinations close by: agents arrive within the rollout
    args[3] = torch.where(torch.isnan(args[3]), dev(sc['destination']), args[3])
    m = MLAPM(version=ver, **_LAWS[ver])
    steps = 37                                               # 1 + 4 graphs of 8 + 4 single frames
    fp, fv = m.rollout(*args, 0.08, 0.3, steps)
    sp, sv = m.rollout(*args, 0.08, 0.3, steps, fused=False)
    assert torch.equal(torch.isnan(fp), torch.isnan(sp)) and torch.equal(torch.isnan(fv), torch.isnan(sv))
    gone = torch.isnan(fp[-1, :, 0]) & ~torch.isnan(fp[0, :, 0])
    assert int(gone.sum()) > 0                               # somebody did arrive
    assert torch.equal(torch.nan_to_num(fp), torch.nan_to_num(sp)) and torch.equal(torch.nan_to_num(fv), torch.nan_to_num(sv))


def test_collision_counts_grid_form_equals_sweeps_and_oracle(oracle, monkeypatch):
    """The per-frame cell-grid form of the many-slice path (piml_collision_counts_grid) against the two-sweep form and the
    oracle: pairs that stay together for more than 25 frames (friends rule), distances exactly on a threshold, absent
    agents, coordinates that alias on the 32 x 32 torus (a 60 m hall at 0.5 m cells), negative coordinates, and a frame
    with a coordinate beyond the grid's trusted range (that workgroup walks all pairs)."""
    from piml_amd import ops
    rng = np.random.default_rng(11)
    S, N = 60, 300
    base = (rng.integers(-120, 120, size=(1, N, 2)) * 0.25).astype(np.float32)           # lattice over 60 m, negative too
    p = base + (rng.integers(-1, 2, size=(S, N, 2)) * 0.25).astype(np.float32)            # jitter by lattice steps per frame
    p[:, 10] = p[:, 11] + np.float32(0.25)                                                # a pair together in all 60 frames
    p[:30, 20] = p[:30, 21]                                                               # ... in exactly 30
    p[:20, 30] = p[:20, 31] + np.float32(0.5)                                             # exactly on the 0.5 threshold, 20 frames
    p[rng.random((S, N)) < 0.05] = np.nan
    p[7, 5] = (3.0e7, -2.0e7)                                                             # beyond 1e5 cells
    thr = (0.5, 0.25)
    want = [oracle.collision_detection(p, t).sum(-1) for t in thr]
    for grid in (True, False):
        monkeypatch.setattr(ops, 'COLLISION_GRID', grid)
        got = ops.collision_counts(dev(p), thr).cpu().numpy()
        for h in range(len(thr)):
            assert np.array_equal(got[h], want[h]), (grid, thr[h], np.abs(got[h] - want[h]).max())
    assert want[0].sum() > 0


@pytest.mark.parametrize('T,S,N', [(5, 4, 122), (1, 1, 7), (32, 3, 300), (6, 25, 1100)])
def test_collision_counts_frames_equal_per_frame_calls(T, S, N):
    """The frames of a training window counted in ONE launch: record f is bitwise what a launch on frame f alone writes."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + N)
    frames = []
    for _ in range(T):
        p = (torch.rand(S, N, 2, generator=g) * 8.0).to(DEV)
        p[:, ::9] = float('nan')
        frames.append(p)
    got = ops.collision_counts_frames(frames, (0.5, 0.25))
    assert len(got) == T
    for f, rec in zip(frames, got):
        assert torch.equal(rec, ops.collision_counts(f, (0.5, 0.25)))
