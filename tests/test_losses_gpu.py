"""GPU: ops.rollout_losses (piml_amd/csrc/losses.hip) against the torch-operator expression of the same losses --
BaseSimulator.multiple_rollout_mse_loss / multiple_rollout_collision_loss on the masked, gated positions, exactly as
_training_rollout_frames assembles them (reference src/models/simulators.py:172-249, 790-819) -- values and the gradient
with respect to the predicted positions.  Tolerance 1e-5 relative to each tensor's largest magnitude; measured error printed."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def torch_losses(sim, p, labels, mask_pred, gates, coll, hard, abn, decay):
    keep = (mask_pred != 0).unsqueeze(-1)
    gate4 = gates.view(1, -1, 1, 1)
    p_res = torch.where(keep & gate4, p, torch.zeros_like(p))
    lab = torch.where(keep, labels, torch.zeros_like(labels))[..., :2]
    out = [sim.multiple_rollout_mse_loss(p_res, lab, decay, reduction='sum')]
    for c in (coll, hard):
        out.append(sim.multiple_rollout_collision_loss(p_res, lab, decay, 10, c, reduction='sum', abnormal_mask=abn))
    return torch.stack(out)


@pytest.mark.parametrize('C,T,N,decay,with_abn', [(4, 5, 122, 1.0, False), (4, 10, 37, 0.9, True), (1, 2, 1, 0.5, False),
                                                 (16, 5, 976, 0.9, True), (3, 7, 30000, 1.0, False)])
def test_rollout_losses_match_the_torch_expression(C, T, N, decay, with_abn):
    from piml_amd import ops
    from piml_amd.models.simulators import BaseSimulator
    sim = BaseSimulator.__new__(BaseSimulator)          # the loss methods use no state
    g = torch.Generator().manual_seed(C * 1000 + N)
    p = (torch.randn(C, T, N, 2, generator=g) * 3).to(DEV)
    labels = (torch.randn(C, T, N, 7, generator=g) * 3).to(DEV)
    mask_pred = (torch.rand(C, T, N, generator=g) < 0.7).long().to(DEV) * 3
    mask_pred[:, T - 1] = 0 if T > 2 else mask_pred[:, T - 1]           # a frame nobody is predicted in: gate closed
    gates = mask_pred.sum(dim=(0, 2)) > 0
    absent = (torch.rand(C, T, N, generator=g) < 0.1).to(DEV) & (mask_pred == 0)
    p = torch.where(absent.unsqueeze(-1), torch.full_like(p, float('nan')), p)            # absent agents: NaN, masked out
    labels = torch.where(absent.unsqueeze(-1), torch.full_like(labels, float('nan')), labels)
    coll = (torch.rand(C, T, N, generator=g) < 0.05).float().to(DEV) * 2
    hard = (torch.rand(C, T, N, generator=g) < 0.02).float().to(DEV)
    abn = (torch.rand(N, generator=g) < 0.8).float().to(DEV) if with_abn else None
    w = torch.tensor([1.0, 10.0, 100.0], device=DEV)

    p1 = p.clone().requires_grad_(True)
    got = torch.stack(ops.rollout_losses(p1, labels, mask_pred, gates, coll, hard, abn, decay))
    (got * w).sum().backward()
    p2 = p.clone().requires_grad_(True)
    want = torch_losses(sim, p2, labels, mask_pred, gates, coll, hard, abn, decay)
    (want * w).sum().backward()
    err = float(((got.double() - want.double()).abs() / want.double().abs().clamp_min(1e-30)).max())
    g2 = torch.nan_to_num(p2.grad)
    gerr = float((p1.grad.double() - g2.double()).abs().max() / g2.double().abs().max().clamp_min(1e-30))
    print(f'rollout losses C={C} T={T} N={N} decay={decay}: sums {err:.1e}, gradient {gerr:.1e}')
    assert torch.isfinite(got).all() and torch.isfinite(p1.grad).all()
    assert err <= 1e-5 and gerr <= 1e-5
    # without collision records only the squared error is asked for
    only = torch.stack(ops.rollout_losses(p, labels, mask_pred, gates, None, None, None, decay))
    assert float((only[0] - got[0]).abs()) <= 1e-6 * float(got[0].abs()) and float(only[1]) == 0.0 and float(only[2]) == 0.0


def test_rollout_losses_are_deterministic():
    from piml_amd import ops
    g = torch.Generator().manual_seed(3)
    C, T, N = 8, 5, 5000
    p = torch.randn(C, T, N, 2, generator=g).to(DEV)
    labels = torch.randn(C, T, N, 6, generator=g).to(DEV)
    mask_pred = (torch.rand(C, T, N, generator=g) < 0.8).long().to(DEV)
    gates = mask_pred.sum(dim=(0, 2)) > 0
    coll = (torch.rand(C, T, N, generator=g) < 0.1).float().to(DEV)
    first = torch.stack(ops.rollout_losses(p, labels, mask_pred, gates, coll, coll, None, 0.9))
    for _ in range(10):
        assert torch.equal(first, torch.stack(ops.rollout_losses(p, labels, mask_pred, gates, coll, coll, None, 0.9)))


@pytest.mark.parametrize('model', ['pinnsf_m', 'pinnsf_bm'])
def test_param_grad_sink_equals_autograd_accumulation(model):
    """ops.ParamGradSink: the weight gradients of several backward passes through the same network inside one optimiser step,
    summed by the slot-sum launches themselves (PIML_ACCUMULATE / piml_*_bwd_acc), are BITWISE what autograd's per-tensor
    accumulation gives (same sums, same order), for the fused network (`pinnsf_m`) and the encoder / row decoder / head64
    operators of `pinnsf_bm`."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_simulator_gpu import sim_args
    from piml_amd import ops
    from piml_amd.models.simulators import BaseSimulator
    torch.manual_seed(5)
    sim = BaseSimulator(sim_args(model=model, dropout=0.0, learning_rate=1e-3))
    net = sim.model
    net.train(False)
    # (the message path: on the sums path -- messages_wanted = False -- the sink accumulates the FOLDED layers' gradients and
    # unfolds their running sums, which is the same arithmetic in another order: test_param_grad_sink_on_the_sums_path)
    net.messages_wanted = True
    g = torch.Generator().manual_seed(11)
    frames = [(torch.randn(4, 122, 6, 6, generator=g).to(DEV), torch.randn(4, 122, 10, 6, generator=g).to(DEV),
               torch.randn(4, 122, 7, generator=g).to(DEV)) for _ in range(3)]

    def run(use_sink):
        for p in net.parameters():
            p.grad = None
        import contextlib
        sink = ops.ParamGradSink()
        with (sink.step() if use_sink else contextlib.nullcontext()):
            loss = 0
            for pf, of, sf in frames:
                out = net(pf, of, sf)
                loss = loss + out[0].square().sum() + (out[-1].sum() if model == 'pinnsf_bm' else 0)
            loss.backward()
        return [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
    want, got = run(False), run(True)
    assert sum(w is not None for w in want) >= 24
    for (name, _), w, gt in zip(net.named_parameters(), want, got):
        assert (w is None) == (gt is None), name
        if w is not None:
            assert torch.equal(w, gt), name


def test_param_grad_sink_on_the_sums_path():
    """ops.ParamGradSink with the network on the agents' sums of h2 (model.messages_wanted = False): the folded layers' gradients
    accumulate folded and are unfolded from their running sums -- equal to autograd's accumulation of the unfolded gradients to
    float32 rounding (1e-6 of the tensor's largest entry), deterministic, and every parameter gets its gradient."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_simulator_gpu import sim_args
    from piml_amd import ops
    from piml_amd.models.simulators import BaseSimulator
    import contextlib
    torch.manual_seed(5)
    sim = BaseSimulator(sim_args(model='pinnsf_m', dropout=0.0, learning_rate=1e-3))
    net = sim.model
    net.train(False)
    assert net.messages_wanted is False
    g = torch.Generator().manual_seed(11)
    frames = [(torch.randn(4, 122, 6, 6, generator=g).to(DEV), torch.randn(4, 122, 10, 6, generator=g).to(DEV),
               torch.randn(4, 122, 7, generator=g).to(DEV)) for _ in range(3)]

    def run(use_sink):
        for p in net.parameters():
            p.grad = None
        sink = ops.ParamGradSink()
        with (sink.step() if use_sink else contextlib.nullcontext()):
            loss = 0
            for pf, of, sf in frames:
                out = net(pf, of, sf)
                assert out[1] is None and out[2] is None
                loss = loss + out[0].square().sum()
            loss.backward()
        return [None if p.grad is None else p.grad.detach().clone() for p in net.parameters()]
    want, got, again = run(False), run(True), run(True)
    assert sum(w is not None for w in want) >= 24
    worst = 0.0
    for (name, _), w, gt, g2 in zip(net.named_parameters(), want, got, again):
        assert (w is None) == (gt is None), name
        if w is not None:
            assert torch.equal(gt, g2), name
            worst = max(worst, float((w - gt).abs().max() / w.abs().max()))
    print(f'ParamGradSink on the sums path vs autograd accumulation: max rel diff {worst:.1e}')
    assert worst <= 2e-6


def test_multi_copy_one_launch_for_a_batch_of_tensors():
    """ops.multi_copy (piml_multi_copy): the training batch into the captured step's static inputs -- mixed dtypes, odd byte
    counts, an empty tensor, more pairs than one launch holds; anything non-contiguous falls back to torch._foreach_copy_."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(4, 5, 122, 6, 6), (4, 5, 122, 7), (4, 5, 122), (3,), (0, 2), (1,), (17, 13)] * 5       # 35 pairs
    dtypes = [torch.float32, torch.float32, torch.int64, torch.uint8, torch.float32, torch.bool, torch.int32] * 5
    srcs = [(torch.rand(*s, generator=g) * 100).to(dt).cuda() for s, dt in zip(shapes, dtypes)]
    dsts = [torch.zeros_like(s) for s in srcs]
    ops.multi_copy(dsts, srcs)
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    # unaligned views (odd byte offsets) and a non-contiguous pair
    base_s, base_d = torch.arange(1000, dtype=torch.uint8).cuda(), torch.zeros(1000, dtype=torch.uint8).cuda()
    ops.multi_copy([base_d[3:500]], [base_s[5:502]])
    assert torch.equal(base_d[3:500], base_s[5:502]) and int(base_d[:3].sum()) == 0 and int(base_d[500:].sum()) == 0
    a, b = torch.rand(8, 8).cuda(), torch.zeros(8, 8).cuda()
    ops.multi_copy([b.t()], [a.t()])
    assert torch.equal(a, b)


def test_rollout_prologue_and_frame_node_equal_the_torch_expressions():
    """ops.rollout_prologue (piml_rollout_prologue) against the statements it replaces (src/models/simulators.py:672-697, :707),
    and ops.rollout_frame (one autograd node: frame step + features of its result) against the two-operator composition:
    outputs bit-equal, gradients equal up to the order of the float atomics."""
    import types
    from piml_amd import ops
    from piml_amd.scenes import synthetic_gc_scene
    g = torch.Generator().manual_seed(9)
    C, T, N, M = 3, 5, 97, 60
    sc = synthetic_gc_scene(N, M, seed=2, channels=C * T)
    f = lambda k: torch.tensor(sc[k], device='cuda').view(C, T, N, -1).contiguous()
    data = types.SimpleNamespace(position=f('position'), velocity=f('velocity'), acceleration=f('acceleration'), destination=f('destination'))
    data.dest_idx = torch.randint(0, 3, (C, T, N), generator=g).cuda()
    data.mask_p = (torch.rand(C, T, N, generator=g) > 0.3).float().cuda()
    data.mask_p_pred = (data.mask_p * (torch.rand(C, T, N, generator=g) > 0.4).float().cuda()).contiguous()
    data.mask_p_pred[:, 3] = 0                      # a frame nobody is predicted in: its gate is closed
    data.self_features = torch.randn(C, T, N, 7, generator=g).cuda()
    for t0 in (0, 2):
        pro = ops.rollout_prologue(data, t0)
        assert pro is not None
        assert torch.equal(pro['mask_pred'], data.mask_p_pred.long())
        assert torch.equal(pro['new_flag_u8'].bool(), (data.mask_p - data.mask_p_pred).long() == 1)
        gates = data.mask_p_pred.long().sum(dim=(0, 2)) > 0
        assert torch.equal(pro['gates'], gates) and torch.equal(pro['gates_f'], gates.float()) and int(pro['nan_flag']) == 0
        for k, src in (('p', data.position), ('v', data.velocity), ('a', data.acceleration), ('dest', data.destination)):
            a, b = pro[k], src[:, t0]
            assert torch.equal(a.isnan(), b.isnan()) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        assert torch.equal(pro['dest_idx'], data.dest_idx[:, t0]) and torch.equal(pro['speed'], data.self_features[:, t0, :, 6:])
    # the frame node
    pro = ops.rollout_prologue(data, 0)
    obstacles = torch.tensor(sc['obstacles'], device='cuda')
    waypoints = torch.rand(4, N, 2, generator=g).cuda() * 30
    dest_num = torch.full((N,), 4, dtype=torch.int64, device='cuda')
    series = (data.position, data.velocity, data.acceleration, data.destination, data.dest_idx)
    res = []
    for fused in (True, False):
        leaves = [torch.nan_to_num(pro[k]).clone().requires_grad_(True) for k in ('p', 'v', 'a')]
        a_pred = torch.randn(C, N, 2, generator=torch.Generator().manual_seed(4)).cuda().requires_grad_(True)
        nan_flag = torch.zeros((), dtype=torch.int32, device='cuda')
        if fused:
            out = ops.rollout_frame(*leaves, a_pred, pro['dest'], pro['dest_idx'], waypoints, dest_num, 0.08, pro['new_flag_u8'], series, 1,
                                    nan_flag, obstacles, pro['speed'])
        else:
            st = ops.train_rollout_step(*leaves, a_pred, pro['dest'], pro['dest_idx'], waypoints, dest_num, 0.08, new_flag=pro['new_flag_u8'],
                                        series=series, t_next=1, nan_flag=nan_flag, zero_nan=True)
            out = (*st, *ops.relative_features_self(st[0], st[1], st[2], st[3], obstacles, pro['speed']))
        ws = [torch.randn(o.shape, generator=torch.Generator().manual_seed(20 + i)).cuda() for i, o in enumerate(out)]
        loss = sum((torch.nan_to_num(o) * w).sum() for o, w in zip(out, ws) if o.dtype == torch.float32 and o.requires_grad)
        grads = torch.autograd.grad(loss, leaves + [a_pred])
        res.append((out, grads))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('focus', [True, False])
@pytest.mark.parametrize('C,T,N,t_start', [(4, 5, 122, 0), (2, 7, 1500, 2)])
def test_rollout_losses_frames_equals_the_stacked_form(C, T, N, t_start, focus):
    """ops.rollout_losses_frames (count records per frame, gated inside, weights and statistics inside) against
    ops.rollout_losses on the stacked / gated tensors + the torch expressions around it (src/models/simulators.py:708-728,
    790-819): every output, the statistics and the gradient."""
    from piml_amd import ops
    g = torch.Generator().manual_seed(C * 100 + T)
    p = torch.randn(C, T, N, 2, generator=g).cuda().requires_grad_(True)
    labels = torch.randn(C, T, N, 12, generator=g).cuda()
    mask = (torch.rand(C, T, N, generator=g) > 0.3).long().cuda()
    mask[:, T - 2] = 0                                        # a closed gate
    gates = mask.sum(dim=(0, 2)) > 0
    recs = [None] * t_start + [(torch.rand(2, C, N, generator=g) > 0.9).float().cuda() for _ in range(T - t_start)]
    abn = (torch.rand(N, generator=g) > 0.1).float().cuda()
    w_c, w_h = 0.7, 0.7 * 3.0
    total, mse, cw, hw, stats = ops.rollout_losses_frames(p, labels, mask, gates, recs, focus, abn, 0.9, w_c if focus else 0.0,
                                                          w_h if focus else 0.0)
    gp, = torch.autograd.grad(total, p)
    stack = torch.stack([torch.zeros(2, C, N, device='cuda') if r is None else r for r in recs], dim=2) * gates.float().view(1, 1, -1, 1)
    coll, hard = stack[0], stack[1]
    sums = ops.rollout_losses(p, labels, mask, gates, coll if focus else None, hard if focus else None, abn, 0.9)
    want_total = sums[0] + (sums[1] * w_c + sums[2] * w_h if focus else 0.0)
    gw, = torch.autograd.grad(want_total, p)
    close = lambda a, b: torch.allclose(a, b, rtol=2e-6, atol=1e-6)
    assert close(total, want_total) and close(mse, sums[0])
    if focus:
        assert close(cw, sums[1] * w_c) and close(hw, sums[2] * w_h)
    assert float(stats[0]) == float(coll.sum()) and float(stats[1]) == float(hard.sum()) and float(stats[2]) == float((mask == 1).sum())
    assert torch.allclose(gp, gw, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('C,T,N,k,t_start', [(4, 5, 122, 6, 0), (2, 7, 33, 6, 2), (4, 5, 976, 6, 0), (1, 3, 5, 4, 1)])
def test_collision_pred_loss_equals_the_torch_expression(C, T, N, k, t_start):
    """ops.collision_pred_loss (one launch each way) against the statements it replaces -- per frame the model's last output and
    calculate_collision_label of the frame's pedestrian features (src/models/simulators.py:731-733), behind the loop two stacks
    with zero frames before t_start, the gates, F.binary_cross_entropy(reduction='sum') * weight, the accuracy (:826-830) -- and
    their autograd; incl. a gated-off frame, predictions at 0.5 (round half to even), near 0 and near 1 (clamped logarithms)."""
    import torch.nn.functional as F
    from piml_amd import ops
    g = torch.Generator().manual_seed(5)
    nfr = T - t_start
    gates_f = torch.ones(T)
    gates_f[t_start + nfr // 2] = 0.0                                          # a frame nobody is predicted in
    gates_f = gates_f.to(DEV)
    preds, feats = [], []
    for f in range(nfr):
        p = torch.rand(C, N, k, generator=g)
        p.view(-1)[:4] = torch.tensor([0.5, 1e-30, 1.0 - 1e-7, 0.5000001])
        preds.append(p.to(DEV).requires_grad_(True))
        x = torch.randn(C, N, k, 6, generator=g) * torch.tensor([0.6, 0.6, 1.0, 1.0, 1.0, 1.0])
        x.view(-1, 6)[0] = 0.0                                                 # distance exactly zero: no collision (d != 0)
        feats.append(x.to(DEV))
    weight = 5e-2
    got_loss, got_acc = ops.collision_pred_loss(preds, feats, gates_f, t_start, T, weight)
    (got_loss * 1.7).backward()
    got_grads = [p.grad.clone() for p in preds]
    # the reference's statements
    ref_p = [p.detach().clone().requires_grad_(True) for p in preds]
    gk = gates_f.view(1, -1, 1, 1)
    pad = [torch.zeros(C, N, k, device=DEV)] * t_start
    pc = torch.stack(pad + ref_p, dim=1) * gk
    tc = torch.stack(pad + [ops.collision_label(x) for x in feats], dim=1) * gk
    want_loss = F.binary_cross_entropy(pc, tc, reduction='sum') * weight
    want_acc = torch.sum(torch.round(pc) == tc) / tc.numel()
    (want_loss * 1.7).backward()
    assert abs(float(got_loss) - float(want_loss)) <= 2e-6 * abs(float(want_loss))
    assert float(got_acc) == pytest.approx(float(want_acc), abs=1e-6)
    for a, b in zip(got_grads, ref_p):
        assert torch.allclose(a, b.grad, rtol=2e-6, atol=1e-7)
    # deterministic
    l2, _ = ops.collision_pred_loss([p.detach() for p in preds], feats, gates_f, t_start, T, weight)
    assert torch.equal(l2, got_loss.detach())


@pytest.mark.parametrize('rows,k,with_reg,with_cp', [(128, 6, True, True), (1024, 6, False, True), (4096, 6, True, False), (37, 6, False, False)])
def test_pointwise_losses_equal_the_torch_expressions(rows, k, with_reg, with_cp):
    """ops.pointwise_losses (one launch) against src/models/simulators.py:333-352: mse_loss(sum) on labels[:, 4:6], the L1 regulariser
    of the messages, binary_cross_entropy(sum) on labels[:, 6:], their sum in the reference's order, and the gradients."""
    import torch.nn.functional as F
    from piml_amd import ops
    g = torch.Generator().manual_seed(9)
    pred = torch.randn(rows, 2, generator=g).to(DEV).requires_grad_(True)
    msgs = (torch.randn(rows, k, 2, generator=g)).to(DEV).requires_grad_(True)
    with torch.no_grad():
        msgs.view(-1)[:3] = torch.tensor([0.0, -0.0, 1e-30], device=DEV)
    coll = torch.rand(rows, k, generator=g)
    coll.view(-1)[:3] = torch.tensor([0.5, 1e-30, 1.0 - 1e-7])
    coll = coll.to(DEV).requires_grad_(True)
    labels = torch.cat((torch.randn(rows, 6, generator=g), (torch.rand(rows, k, generator=g) < 0.2).float()), 1).to(DEV)
    w = 1e-2
    got = ops.pointwise_losses(pred, labels, w if with_reg else 0.0, msgs if with_reg else None, coll if with_cp else None)
    (got[0] * 0.7).backward()
    got_g = [t.grad.clone() if t.grad is not None else None for t in (pred, msgs, coll)]
    for t in (pred, msgs, coll):
        t.grad = None
    mse = F.mse_loss(pred, labels[:, 4:6], reduction='sum')
    loss = mse
    reg = cp = None
    if with_reg:
        reg = torch.sum(w * torch.abs(msgs))
        loss = loss + reg
    if with_cp:
        cp = F.binary_cross_entropy(coll, labels[:, 6:], reduction='sum')
        loss = loss + cp
    (loss * 0.7).backward()
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)
    assert rel(got[0], loss) <= 2e-6 and rel(got[1], mse) <= 2e-6
    if with_reg:
        assert rel(got[2], reg) <= 2e-6
    if with_cp:
        assert rel(got[3], cp) <= 2e-6
    for a, t in zip(got_g, (pred, msgs, coll)):
        if t.grad is None:
            assert a is None
        else:
            assert torch.allclose(a, t.grad, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('kind', ['sum', 'sum_obs', 'ksum'])
def test_frame_node_with_the_models_tail_equals_the_separate_launches(kind):
    """ops.rollout_frame(tail=...) (piml_train_step_tail_fwd / bwd: the model's tail under the agent-axis norm, quirk Q2, inside the
    frame step's launches) against ops.pinnsf_epilogue[_ksum](agent_norm=True) followed by ops.rollout_frame(a_pred): every output
    and every gradient BITWISE (same reductions, same arithmetic), at a slice size above one pass of the workgroup (N = 300)."""
    import types
    from piml_amd import ops
    from piml_amd.scenes import synthetic_gc_scene
    g = torch.Generator().manual_seed(19)
    C, T, N, M = 3, 4, 300, 40
    sc = synthetic_gc_scene(N, M, seed=4, channels=C * T)
    f = lambda k: torch.tensor(sc[k], device='cuda').view(C, T, N, -1).contiguous()
    data = types.SimpleNamespace(position=f('position'), velocity=f('velocity'), acceleration=f('acceleration'), destination=f('destination'))
    data.dest_idx = torch.randint(0, 3, (C, T, N), generator=g).cuda()
    data.mask_p = (torch.rand(C, T, N, generator=g) > 0.3).float().cuda()
    data.mask_p_pred = (data.mask_p * (torch.rand(C, T, N, generator=g) > 0.4).float().cuda()).contiguous()
    data.self_features = torch.randn(C, T, N, 7, generator=g).cuda()
    pro = ops.rollout_prologue(data, 0)
    obstacles = torch.tensor(sc['obstacles'], device='cuda')
    waypoints = torch.rand(4, N, 2, generator=g).cuda() * 30
    dest_num = torch.full((N,), 4, dtype=torch.int64, device='cuda')
    series = (data.position, data.velocity, data.acceleration, data.destination, data.dest_idx)
    kp, ko = (6, 10) if kind == 'ksum' else (1, 1)
    res = []
    for fused in (True, False):
        gg = torch.Generator().manual_seed(4)
        leaves = [torch.nan_to_num(pro[k]).clone().requires_grad_(True) for k in ('p', 'v', 'a')]
        acc_p = torch.randn(*((C, N, kp, 2) if kind == 'ksum' else (C, N, 2)), generator=gg).cuda().requires_grad_(True)
        acc_o = None if kind == 'sum' else torch.randn(*((C, N, ko, 2) if kind == 'ksum' else (C, N, 2)), generator=gg).cuda().requires_grad_(True)
        sf = torch.randn(C, N, 7, generator=gg).cuda()
        sf[1, :, 0] = 0.0                             # a slice whose x components are all zero: the norm's 0 -> 0.1 branch
        sf.requires_grad_(True)
        nan_flag = torch.zeros((), dtype=torch.int32, device='cuda')
        tail = (acc_p, acc_o, sf, 2.0)
        if fused:
            out = ops.rollout_frame(*leaves, None, pro['dest'], pro['dest_idx'], waypoints, dest_num, 0.08, pro['new_flag_u8'], series, 1,
                                    nan_flag, obstacles, pro['speed'], tail=tail)
        else:
            a_pred = ops.pinnsf_epilogue_ksum(acc_p, acc_o, sf, 2.0, agent_norm=True) if kind == 'ksum' else \
                ops.pinnsf_epilogue(acc_p, acc_o, sf, 2.0, agent_norm=True)
            out = ops.rollout_frame(*leaves, a_pred, pro['dest'], pro['dest_idx'], waypoints, dest_num, 0.08, pro['new_flag_u8'], series, 1,
                                    nan_flag, obstacles, pro['speed'])
        ws = [torch.randn(o.shape, generator=torch.Generator().manual_seed(20 + i)).cuda() for i, o in enumerate(out)]
        loss = sum((torch.nan_to_num(o) * w).sum() for o, w in zip(out, ws) if o.dtype == torch.float32 and o.requires_grad)
        grads = torch.autograd.grad(loss, leaves + [acc_p, sf] + ([acc_o] if acc_o is not None else []))
        res.append((out, grads))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))
    # the features' backward adds with float atomics in BOTH forms, and the tail's gradients take sums over a slice's agents of what
    # those atomics produced: the separate launches differ from THEMSELVES run to run by up to 1.2e-7 of a tensor's largest entry
    # (measured; an elementwise relative bound trips on entries that cancel).  Bound: 2e-6 of the largest entry.
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


@pytest.mark.parametrize('wd', [0.0, 1e-4])
def test_one_launch_adam_is_bitwise_pytorchs_fused_adam(wd):
    """piml_amd.optim.Adam (piml_adam_step: every parameter of a group and its step counter in ONE launch) against
    torch.optim.Adam(fused=True, capturable=True) -- the optimiser of both training loops (src/models/simulators.py:69-71) -- over eight
    steps on PINNSF's parameter shapes plus odd sizes: parameters and optimiser state bitwise, state_dict()s interchangeable."""
    from piml_amd.optim import Adam
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 6), (128,), (128, 128), (128,), (128, 128), (128,), (64, 128), (64,), (64, 64), (64,), (2, 64), (2,), (1,), (7, 3), (1025,)]
    mine = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.2).to(DEV)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    kw = dict(lr=2e-4, weight_decay=wd, capturable=True, fused=True)
    oa, ob = Adam(mine, **kw), torch.optim.Adam(ref, **kw)
    for it in range(8):
        for p, q in zip(mine, ref):
            gr = (torch.randn(*p.shape, generator=g) * (10.0 ** (it % 3 - 2))).to(DEV)
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        for p, q in zip(mine, ref):
            assert torch.equal(p, q), (it, tuple(p.shape))
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    for k in sb:
        for name in ('step', 'exp_avg', 'exp_avg_sq'):
            assert torch.equal(sa[k][name], sb[k][name]), (k, name)
    ob.load_state_dict(oa.state_dict())        # interchangeable
