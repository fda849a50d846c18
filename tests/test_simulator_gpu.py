"""GPU parity of the rollout loops (HOT LOOP B / C of SURVEY.md section 3) against the reference's own
`BaseSimulator` outputs captured in tests/golden/rollout.npz: same data, same weights."""
import types

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# Bounds of the differentiable-rollout parity tests against the reference's goldens (BPTT through 5-10 frames of
# closed loop; float32 round-off compounds through the frames): loss scalars relative, gradients relative to the
# largest entry of each tensor.  The measured values are printed by every test.
SCALAR_TOL = 1e-4
GRAD_TOL = 1e-4


def sim_args(**kw):
    a = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16,
        decoder_hidden_layers=2, dropout=0.5, activation='relu', dataset_name='gc1560', res_hidden_layers=3,
        model='pinnsf_m', device=DEV, gpus='3', learning_rate=0.002, weight_decay=5e-4, batch_size=3,
        topk_ped=6, topk_obs=10, sight_angle_ped=90, sight_angle_obs=90, dist_threshold_ped=4,
        dist_threshold_obs=4, num_history_velocity=1, skip_frames=25, valid_steps=5, time_decay=1,
        reg_weight=0., collision_threshold=0.5, collision_loss_weight=10, val_coll_weight=30,
        hard_collision_penalty=10, teacher_weight=0, collision_pred_weight=10, collision_focus_weight=10,
        new_collision_loss_flag=0, collision_loss_version='v0', finetune_lr_decay=1, finetune_wd_aug=1,
        ft_lr_decay2=0., exp_name='golden', model_name_suffix='x', epochs=1, patience=1, ft_patience=5,
        pinnsf_interaction='sim', iter_flag=0, true_label_weight=0)
    a.__dict__.update(kw)
    return a


def load_data(g, prefix):
    d = types.SimpleNamespace()
    for k in g.files:
        if k.startswith(prefix + '/') and k.count('/') == 1:
            name = k.split('/')[1]
            v = g[k]
            if name in ('time_unit', 'num_frames'):
                setattr(d, name, v.item())
            elif not name.startswith('out_') and name not in ('scalars', 'counts'):
                setattr(d, name, torch.tensor(v, device=DEV))
    d.num_frames = int(d.num_frames)
    return d


def make_sim(g, args, sd_prefix):
    from piml_amd.models.simulators import BaseSimulator
    sim = BaseSimulator(args)
    sd = {k[len(sd_prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(sd_prefix)}
    sim.model.load_state_dict(sd, strict=True)
    sim.model.eval()
    return sim


def test_inference_rollout_matches_reference():
    g = golden('rollout')
    sim = make_sim(g, sim_args(), 'sd_m/')
    data = load_data(g, 'roll')
    with torch.no_grad():
        res = sim.get_multiple_rollouts(data, t_start=0, load_model=False)            # fused + HIP-graph replay
        eager = sim.get_multiple_rollouts(data, t_start=0, load_model=False, use_graph=False, fused=False)
        fused = sim.get_multiple_rollouts(data, t_start=0, load_model=False, use_graph=False, fused=True)
    # the fused integrator kernel and the captured step reproduce the torch-op step bit for bit
    for other in (eager, fused):
        for k in ('position', 'velocity', 'acceleration', 'mask_p'):
            assert torch.equal(torch.nan_to_num(getattr(res, k)), torch.nan_to_num(getattr(other, k))), k
    p, ref = res.position.cpu().numpy(), g['roll/out_position']
    m, mref = res.mask_p.cpu().numpy(), g['roll/out_mask_p']
    # short horizon: tight; the first steps must agree to float32 round-off
    for horizon, tol in ((3, 2e-6), (10, 1e-5), (40, 1e-3)):
        a, b = p[:horizon], ref[:horizon]
        assert np.array_equal(np.isnan(a), np.isnan(b)), horizon
        assert np.nanmax(np.abs(a - b)) <= tol, (horizon, np.nanmax(np.abs(a - b)))
        assert np.array_equal(m[:horizon], mref[:horizon])
    # long horizon (160 frames): neighbour-set flips make trajectories diverge chaotically, so compare
    # statistically: who is in the scene, and the typical displacement error
    both = ~np.isnan(p[..., 0]) & ~np.isnan(ref[..., 0])
    assert (np.isnan(p[..., 0]) != np.isnan(ref[..., 0])).mean() < 0.01
    err = np.linalg.norm(p - ref, axis=-1)[both]
    assert np.median(err) < 1e-3 and err.mean() < 0.05


@pytest.mark.parametrize('model_name', ['pinnsf_m', 'pinnsf_bm'])
def test_training_rollout_matches_reference(model_name):
    g = golden('rollout')
    tag = f'train_{model_name}'
    sim = make_sim(g, sim_args(model=model_name), f'{tag}/sd/')
    data = load_data(g, tag)
    labels_before = data.labels.clone()
    out = sim.test_multiple_rollouts_for_training(data)
    out[0].backward()
    got = np.array([float(x.detach()) for x in out])
    ref = g[f'{tag}/scalars']
    scal = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)))
    assert [sim.collision_count, sim.hard_collision_count] == list(g[f'{tag}/counts'])
    assert torch.equal(torch.nan_to_num(data.labels), torch.nan_to_num(labels_before))   # caller's data untouched
    worst = 0.0
    for k, p in sim.model.named_parameters():
        ref_g = g[f'{tag}/grad/{k}']
        got_g = np.zeros_like(ref_g) if p.grad is None else p.grad.cpu().numpy()
        scale = max(np.abs(ref_g).max(), 1e-6)
        worst = max(worst, np.abs(got_g - ref_g).max() / scale)
    print(f'training rollout {model_name}: scalars max rel err {scal:.2e} (bar {SCALAR_TOL:g}), '
          f'gradients max err / max|g| {worst:.2e} (bar {GRAD_TOL:g})')
    assert scal <= SCALAR_TOL, (got, ref)
    assert worst <= GRAD_TOL, worst


def test_train_batch_runs_and_learns():
    """A few Adam steps on the channelled batch reduce its loss (the fine-tune loop end to end)."""
    g = golden('rollout')
    sim = make_sim(g, sim_args(learning_rate=1e-3, weight_decay=0.0), 'train_pinnsf_m/sd/')
    data = load_data(g, 'train_pinnsf_m')
    first = sim.train_batch(data)['loss']
    for _ in range(15):
        last = sim.train_batch(data)['loss']
    assert last < first


def test_rollout_speed_report():
    """Not a parity test: prints simulated steps/s of the inference rollout on the real GC clip
    (reference on 8 CPU cores: 95 steps/s)."""
    import time
    g = golden('rollout')
    sim = make_sim(g, sim_args(), 'sd_m/')
    data = load_data(g, 'roll')
    with torch.no_grad():
        for graph, fused in ((False, False), (True, False), (True, True)):
            sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph, fused=fused)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph, fused=fused)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'rollout N=122 M=100, {data.num_frames} frames, graph={graph} fused={fused}: '
                  f'{data.num_frames / dt:.0f} steps/s')


def test_two_stream_forward_equals_single_stream():
    """The optional side stream for the obstacle branch changes scheduling, not results."""
    g = golden('rollout')
    sim = make_sim(g, sim_args(), 'sd_m/')
    data = load_data(g, 'train_pinnsf_m')
    args = (data.ped_features[:, 0], data.obs_features[:, 0], data.self_features[:, 0])
    with torch.no_grad():
        ref = sim.model(*args)
        sim.model.obs_stream = torch.cuda.Stream()
        got = sim.model(*args)
        torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert (a is None) == (b is None)          # (the simulator's model does not materialise messages nobody reads)
        if a is not None:
            assert torch.equal(a, b)


def test_graphed_train_step_equals_eager(monkeypatch):
    """The captured whole fine-tuning step (rollout + losses + backward + Adam) reproduces the eager
    step sequence: same losses, same weights after several updates.  Run with the atomics-free relfeat backward
    (ops.DETERMINISTIC_BWD): every other kernel of the step is bit-reproducible, so graph and eager agree EXACTLY
    (measured: 0.0 in four runs); with the default float-atomic backward two runs differ through the order of the
    atomics alone and 26 Adam updates amplify that to 1e-4 .. 2e-3 on single weights, whatever the launch mode."""
    import time
    from piml_amd import ops
    monkeypatch.setattr(ops, 'DETERMINISTIC_BWD', True)
    g = golden('rollout')
    data = load_data(g, 'train_pinnsf_m')
    runs = {}
    for graph in (False, True):
        sim = make_sim(g, sim_args(learning_rate=1e-3, hip_graph=graph), 'train_pinnsf_m/sd/')
        losses = [sim.train_batch(data)['loss'] for _ in range(6)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            sim.train_batch(data)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f'fine-tune step C=4 T=5 N=122, graph={graph}: {dt * 1e3:.2f} ms/step')
        runs[graph] = (losses, [p.detach().clone() for p in sim.model.parameters()], sim.collision_count)
    assert np.allclose(runs[True][0], runs[False][0], rtol=1e-6), (runs[True][0], runs[False][0])
    assert runs[True][2] == runs[False][2]
    worst = max(float(((a - b).abs() / (b.abs() + 1e-2 * b.abs().max())).max()) for a, b in zip(runs[True][1], runs[False][1]))
    print(f'weights after 26 updates, graph vs eager: max |diff| / (|w| + 1 % of the tensor max) = {worst:.1e}')
    for a, b in zip(runs[True][1], runs[False][1]):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)


FLAGS_BM = dict(new_collision_loss_flag=1, teacher_weight=0.5, reg_weight=1e-3)
FLAGS_M = dict(new_collision_loss_flag=1, teacher_weight=0.25, reg_weight=1e-4)
# the loss switches of the shipped UCY experiment (src/configs/exp_configs/piml-ucydata.yaml)
FLAGS_UCY_EXP = dict(valid_steps=10, collision_loss_version='v2', time_decay=0.9, reg_weight=1e-2,
                     collision_pred_weight=5e-2, collision_focus_weight=1, collision_loss_weight=40,
                     hard_collision_penalty=1)


@pytest.mark.parametrize('fixture,tag,model_name,ds,finetune,extra', [
    ('rollout_more', 'ucy_m', 'pinnsf_m', 'ucy', False, {}),
    ('rollout_more', 'gc_res', 'pinnsf_res', 'gc1560', True, {}),
    ('rollout_flags', 'gc_flags_bm', 'pinnsf_bm', 'gc1560', False, FLAGS_BM),
    ('rollout_flags', 'gc_flags_m', 'pinnsf_m', 'gc1560', False, FLAGS_M),
    ('rollout_flags', 'ucy_exp_bm', 'pinnsf_bm', 'ucy', False, FLAGS_UCY_EXP),
    ('rollout_flags', 'gc_exp_bm', 'pinnsf_bm', 'gc2344', False,
     dict(FLAGS_UCY_EXP, collision_loss_weight=200, hard_collision_penalty=2))])
def test_training_rollout_more_configs(fixture, tag, model_name, ds, finetune, extra):
    """UCY configuration (tau = 5/6, 2-point obstacle placeholder, k_o = 2), the residual fine-tune
    network of `--model pinnsf_res`, and the non-default loss switches (label-collision masking, teacher
    acceleration loss, message regulariser, bottleneck collision head), against the reference's scalars
    and gradients."""
    from piml_amd.models.simulators import BaseSimulator
    g = golden(fixture)
    args = sim_args(model=model_name, dataset_name=ds, **{'valid_steps': 6, **extra})
    sim = BaseSimulator(args)
    if finetune:
        sim.set_ft_model(args)
    sd = {k[len(tag) + 4:]: torch.tensor(g[k]) for k in g.files if k.startswith(f'{tag}/sd/')}
    sim.model.load_state_dict(sd, strict=True)
    sim.model.eval()
    data = load_data(g, tag)
    out = sim.test_multiple_rollouts_for_training(data)
    out[0].backward()
    got = np.array([float(x.detach()) for x in out])
    ref = g[f'{tag}/scalars']
    scal = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)))
    assert [sim.collision_count, sim.hard_collision_count] == list(g[f'{tag}/counts'])
    named = dict(sim.model.named_parameters())
    worst = 0.0
    for k in g.files:
        if k.startswith(f'{tag}/grad/'):
            ref_g = g[k]
            p = named[k[len(tag) + 6:]]
            got_g = np.zeros_like(ref_g) if p.grad is None else p.grad.cpu().numpy()
            worst = max(worst, float(np.abs(got_g - ref_g).max() / max(np.abs(ref_g).max(), 1e-6)))
    total = sum(float(p.grad.abs().sum()) for p in named.values() if p.grad is not None)
    tot_err = abs(total - float(g[f'{tag}/grad_abs_sum'])) / float(g[f'{tag}/grad_abs_sum'])
    print(f'training rollout {tag}: scalars max rel err {scal:.2e} (bar {SCALAR_TOL:g}), gradients max err / max|g| '
          f'{worst:.2e} (bar {GRAD_TOL:g}), sum|g| rel err {tot_err:.2e}')
    assert scal <= SCALAR_TOL, (got, ref)
    assert worst <= GRAD_TOL, worst
    assert tot_err <= GRAD_TOL


@pytest.mark.parametrize('model_name', ['pinnsf_m', 'pinnsf_bm'])
def test_fused_train_step_equals_torch_ops(model_name):
    """ops.train_rollout_step (integrator + waypoint switch + injection + NaN flag as one launch each way)
    against the torch-op expression of the same frame step inside the fine-tuning rollout."""
    g = golden('rollout')
    tag = f'train_{model_name}'
    res = {}
    for fused in (True, False):
        sim = make_sim(g, sim_args(model=model_name), f'{tag}/sd/')
        sim.fused_train_step = fused
        out = sim.test_multiple_rollouts_for_training(load_data(g, tag))
        out[0].backward()
        res[fused] = ([float(x.detach()) for x in out], [None if p.grad is None else p.grad.clone()
                                                          for p in sim.model.parameters()],
                      (sim.collision_count, sim.hard_collision_count))
    assert np.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-7), (res[True][0], res[False][0])
    assert res[True][2] == res[False][2]
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            scale = max(float(b.abs().max()), 1e-6)
            assert float((a - b).abs().max()) <= 1e-4 * scale


def test_train_rollout_step_op():
    """The op alone against torch ops, including the gradient cut at re-initialised agents."""
    from piml_amd import ops
    from piml_amd.models.simulators import _gather_waypoints
    C, T, N, D, dt = 3, 4, 50, 3, 0.08
    gen = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    series = [r(C, T, N, 2) for _ in range(4)] + [torch.randint(0, D, (C, T, N), generator=gen).to(DEV)]
    p, v, a, ap = r(C, N, 2), r(C, N, 2), r(C, N, 2), r(C, N, 2)
    p[0, :5] = float('nan')
    waypoints = r(C, D, N, 2)
    dest_idx = torch.randint(0, D, (C, N), generator=gen).to(DEV)
    dest = _gather_waypoints(waypoints, dest_idx).clone()
    dest[:, ::3] = p[:, ::3] + 0.1                      # some agents within 0.5 m of their waypoint
    dest_num = torch.randint(1, D + 1, (N,), generator=gen).to(DEV)
    dest_idx = torch.minimum(dest_idx, (dest_num - 1).expand(C, N))
    new_flag = (torch.rand(C, T, N, generator=gen) < 0.2).to(DEV)
    for t_next in (1, T - 1, T):
        la = [x.clone().requires_grad_(True) for x in (p, v, a, ap)]
        vn, pn = la[1] + la[2] * dt, la[0] + la[1] * dt
        near = torch.norm(la[0] - dest, p=2, dim=-1) < 0.5
        idx = dest_idx + near.long()
        idx = idx - (idx > dest_num - 1).long()
        dn = _gather_waypoints(waypoints, idx)
        an = la[3]
        if t_next < T:
            m = new_flag[:, t_next, :]
            m2 = m.unsqueeze(-1)
            pn, vn, an, dn = (torch.where(m2, s[:, t_next], x) for s, x in zip(series[:4], (pn, vn, an, dn)))
            idx = torch.where(m, series[4][:, t_next], idx)
        w = [r(C, N, 2) for _ in range(3)]
        loss = sum((torch.nan_to_num(x) * y).sum() for x, y in zip((pn, vn, an), w))
        g_ref = torch.autograd.grad(loss, la)
        lb = [x.clone().requires_grad_(True) for x in (p, v, a, ap)]
        flag = torch.zeros((), device=DEV, dtype=torch.int32)
        out = ops.train_rollout_step(lb[0], lb[1], lb[2], lb[3], dest, dest_idx, waypoints, dest_num, dt,
                                     new_flag=new_flag, series=series, t_next=t_next, nan_flag=flag)
        for got, want in zip(out, (pn, vn, an, dn, idx)):
            assert torch.equal(torch.nan_to_num(got.float()), torch.nan_to_num(want.detach().float()))
        loss2 = sum((torch.nan_to_num(x) * y).sum() for x, y in zip(out[:3], w))
        g_out = torch.autograd.grad(loss2, lb)
        for x, y in zip(g_out, g_ref):
            assert torch.allclose(x, y, rtol=1e-6, atol=1e-6)
        assert int(flag) == 0
    ap_nan = ap.clone(); ap_nan[1, 7, 0] = float('nan')
    flag = torch.zeros((), device=DEV, dtype=torch.int32)
    ops.train_rollout_step(p, v, a, ap_nan, dest, dest_idx, waypoints, dest_num, dt, nan_flag=flag)
    assert int(flag) == 1


def test_train_rollout_step_zero_nan():
    """zero_nan = the in-place NaN -> 0 of the next get_relative_features call (data.py:483-484) folded into
    the step, with its gradient cut."""
    from piml_amd import ops
    C, N, D, dt = 2, 40, 2, 0.08
    gen = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    p, v, a, ap = r(C, N, 2), r(C, N, 2), r(C, N, 2), r(C, N, 2)
    v[0, 3, 0] = float('nan'); a[1, 5, 1] = float('nan'); ap[0, 7, :] = float('nan')
    waypoints = r(D, N, 2)
    dest_idx = torch.zeros(C, N, dtype=torch.int64, device=DEV)
    dest = waypoints[0].expand(C, N, 2).contiguous()
    dest_num = torch.full((N,), D, dtype=torch.int64, device=DEV)
    la = [x.clone().requires_grad_(True) for x in (p, v, a, ap)]
    vn, pn, an = la[1] + la[2] * dt, la[0] + la[1] * dt, la[3] * 1.0
    vn = vn.masked_fill(vn.isnan(), 0)
    an = an.masked_fill(an.isnan(), 0)
    w = [r(C, N, 2) for _ in range(3)]
    g_ref = torch.autograd.grad(sum((torch.nan_to_num(x) * y).sum() for x, y in zip((pn, vn, an), w)), la)
    lb = [x.clone().requires_grad_(True) for x in (p, v, a, ap)]
    out = ops.train_rollout_step(lb[0], lb[1], lb[2], lb[3], dest, dest_idx, waypoints, dest_num, dt, zero_nan=True)
    assert not out[1].isnan().any() and not out[2].isnan().any()
    for got, want in zip(out[:3], (pn, vn, an)):
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want.detach()))
    g_out = torch.autograd.grad(sum((torch.nan_to_num(x) * y).sum() for x, y in zip(out[:3], w)), lb)
    for x, y in zip(g_out, g_ref):
        assert torch.allclose(torch.nan_to_num(x), torch.nan_to_num(y), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('N,M', [(4096, 2000), (2500, 700)])
def test_pooled_inference_forward_matches_message_path(N, M):
    """Inference frames sum the neighbour axis BEFORE the encoders' last layer (PIML_POOL_H2: layer 2 with exchanged operands,
    the agents' sums as additions between registers, the last layer folded into the decoder's first).  Same accelerations as
    the message path to float32 rounding, for tiles whose agents start at every offset (k = 6: three, k = 10: five) and a
    row count that is not a multiple of 32; and the rollout that uses it equals the rollout that does not."""
    from piml_amd.scenes import synthetic_rollout_data
    from piml_amd.models.simulators import BaseSimulator
    import piml_amd.models.model as MODEL
    data = synthetic_rollout_data(N, M, 12, DEV)
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args())
    sim.model.eval()
    pf, of, sf = data.ped_features[0], data.obs_features[0], data.self_features[0]
    with torch.no_grad():
        ref = sim.model(pf, of, sf)[0]
        sim.model.predictions_only = True
        try:
            with sim.model.packed_weights():
                got = sim.model(pf, of, sf)
                assert sim.model._ph2 is not None                      # the pooled path did serve the call
                again = sim.model(pf, of, sf)[0]
            assert got[1] is None and got[-1] is None
        finally:
            sim.model.predictions_only = False
        assert sim.model._ph2 is None
        scale = ref.abs().max()
        assert (got[0] - ref).abs().max() <= 1e-5 * scale, ((got[0] - ref).abs().max(), scale)
        assert torch.equal(got[0], again)
        # the whole rollout: pooled inference frames against message frames
        a = sim.get_multiple_rollouts(data, 0, load_model=False)
        keep = MODEL.POOLED_INFERENCE
        MODEL.POOLED_INFERENCE = False
        try:
            b = sim.get_multiple_rollouts(data, 0, load_model=False)
        finally:
            MODEL.POOLED_INFERENCE = keep
        pa, pb = torch.nan_to_num(a.position), torch.nan_to_num(b.position)
        assert (pa - pb).abs().max() <= 1e-4, (pa - pb).abs().max()


@pytest.mark.parametrize('model_name', ['pinnsf_bm', 'pinnsf_bottleneck'])
def test_bottleneck_inference_frames_equal_operator_sequence(model_name):
    """Inference frames of the bottleneck variants: no auxiliary collision head, and the network's epilogue (neighbour-axis sums +
    desired force) inside the integrator's launch (piml_rollout_step_ksum) -- bitwise the trajectory of the operator sequence
    (ops.pinnsf_epilogue_ksum, then ops.rollout_step), eager and replayed."""
    from piml_amd.scenes import synthetic_rollout_data
    from piml_amd.models.simulators import BaseSimulator
    import piml_amd.models.simulators as SIM
    data = synthetic_rollout_data(300, 120, 14, DEV)
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args(model=model_name))
    sim.model.eval()
    res = {}
    with torch.no_grad():
        for merged in (True, False):
            keep = SIM.STEP_WITH_KSUM
            SIM.STEP_WITH_KSUM = merged
            try:
                for graph in (False, True):
                    out = sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph)
                    res[(merged, graph)] = (torch.nan_to_num(out.position).clone(), torch.nan_to_num(out.acceleration).clone())
            finally:
                SIM.STEP_WITH_KSUM = keep
    ref = res[(False, False)]
    assert float(ref[1].abs().max()) > 0
    for key, got in res.items():
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), key


@pytest.mark.parametrize('model', ['pinnsf_m', 'pinnsf_bm'])
def test_pointwise_training_is_bitwise_the_same_under_pytorchs_adam(model, monkeypatch):
    """HOT LOOP A for forty captured steps (dropout 0.5, weight decay, reg_weight, the collision head's BCE for pinnsf_bm) with
    piml_amd.optim.Adam (one launch, piml_adam_step) and with torch.optim.Adam(fused=True, capturable=True): the pointwise step has
    no float atomics, so every parameter and every logged loss must agree BITWISE -- the one-launch optimiser is PyTorch's arithmetic."""
    from piml_amd.models.simulators import BaseSimulator
    from piml_amd import ops
    runs = []
    for which in ('piml', 'torch'):
        monkeypatch.setenv('PIML_ADAM', which)
        torch.manual_seed(666)
        ops.dropout_seed(666, DEV)
        sim = BaseSimulator(sim_args(model=model, learning_rate=2e-4, reg_weight=1e-2, collision_pred_weight=5e-2, hip_graph=True))
        assert type(sim.optimizer).__module__.startswith('piml_amd' if which == 'piml' else 'torch')
        sim.model.train()
        g = torch.Generator().manual_seed(5)
        logs = []
        for it in range(40):
            rows = 128
            batch = (torch.randn(rows, 6, 6, generator=g).to(DEV), torch.randn(rows, 10, 6, generator=g).to(DEV),
                     torch.randn(rows, 7, generator=g).to(DEV),
                     torch.cat((torch.randn(rows, 6, generator=g), (torch.rand(rows, 6, generator=g) < 0.2).float()), 1).to(DEV))
            logs.append(sim.train_batch(batch))
        runs.append(([p.detach().clone() for p in sim.model.parameters()], logs))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    for la, lb in zip(runs[0][1], runs[1][1]):
        assert la == lb


@pytest.mark.parametrize('model_name', ['pinnsf_m', 'pinnsf_bm'])
def test_fused_rollout_losses_and_tail_in_step_equal_their_torch_forms(model_name):
    """The fine-tuning rollout with this round's one-launch forms -- the collision-prediction loss (ops.collision_pred_loss: labels,
    gates, BCE, accuracy), the rollout losses, the model's agent-norm tail inside the frame step's launches (ops.rollout_frame tail=) --
    against the same rollout on their torch / separate-launch forms: every returned scalar (incl. the prediction accuracy) and every
    parameter gradient."""
    g = golden('rollout')
    tag = f'train_{model_name}'
    res = {}
    for fused in (True, False):
        sim = make_sim(g, sim_args(model=model_name, collision_pred_weight=5e-2), f'{tag}/sd/')
        sim.fused_rollout_losses = fused
        sim.tail_in_step = fused
        out = sim.test_multiple_rollouts_for_training(load_data(g, tag))
        out[0].backward()
        res[fused] = ([float(x.detach()) for x in out], [None if p.grad is None else p.grad.clone() for p in sim.model.parameters()])
    assert np.allclose(res[True][0], res[False][0], rtol=2e-5, atol=1e-7), (res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            scale = max(float(b.abs().max()), 1e-6)
            assert float((a - b).abs().max()) <= 1e-4 * scale
