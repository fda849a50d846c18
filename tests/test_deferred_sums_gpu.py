"""GPU: the weight-gradient slot sums of the fused network riding in the relfeat backward's launch (ops.deferred_slot_sums,
PIML_DEFER_SLOT_SUMS of piml_pinnsf_bwd, piml_amd/csrc/reduce.hpp) -- the step of src/models/simulators.py:699-779 (network
backward, then the features' backward) with one launch less.  The sums are the same kernel code on the same slots: every
gradient must be BITWISE what the stand-alone launch gives, whether the relfeat backward takes them, the block's exit does, a
second deferral does, or nothing is deferred because a parameter already holds a gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _scene(N=600, M=300, seed=3):
    from piml_amd.scenes import synthetic_gc_scene
    sc = synthetic_gc_scene(N, M, seed=seed)
    state = torch.tensor(np.concatenate([sc['position'], sc['velocity'], sc['acceleration']], axis=-1), device=DEV)
    return state, torch.tensor(sc['destination'], device=DEV), torch.tensor(sc['obstacles'], device=DEV), torch.tensor(sc['desired_speed'], device=DEV)


def _model(cls='PINNSF_multitask'):
    import piml_amd.models.model as MODEL
    from test_mlpglue_gpu import model_args
    torch.manual_seed(5)
    return getattr(MODEL, cls)(model_args()).to(DEV).eval()


def _step(model, scene, defer, through_features=True, pre_grad=False, twice=False):
    from piml_amd import ops
    import contextlib
    state, dest, obs, v0 = scene
    st = state.clone().requires_grad_(True)
    for p in model.parameters():
        p.grad = torch.zeros_like(p) if pre_grad else None
    N = st.shape[0]
    ctx = ops.deferred_slot_sums() if defer else contextlib.nullcontext()
    outs = []
    with ctx:
        for _ in range(2 if twice else 1):
            pf, of, sf = ops.relative_features_packed_self(st, dest, obs, v0, 0, N)
            if not through_features:
                pf, of, sf = pf.detach(), of.detach(), sf.detach()
            acc = model(pf, of, sf)[0]
            acc.backward(torch.ones_like(acc))
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters() if p.grad is not None] + ([st.grad.clone()] if through_features else [])


@pytest.mark.parametrize('case', ['relfeat_takes_them', 'exit_flushes', 'param_has_grad', 'two_passes'])
def test_deferred_slot_sums_are_bitwise_the_standalone_launch(case):
    model, scene = _model(), _scene()
    kw = dict(through_features=case != 'exit_flushes', pre_grad=case == 'param_has_grad', twice=case == 'two_passes')
    want = _step(model, scene, False, **kw)
    got = _step(model, scene, True, **kw)
    assert len(want) == len(got)
    for i, (a, b) in enumerate(zip(want, got)):
        if i == len(want) - 1 and kw['through_features']:      # d/d(state): float atomics, order not fixed
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(a, b), f'gradient {i} differs by {(a - b).abs().max().item()}'
    assert any(float(g.abs().max()) > 0 for g in got[:-1])


@pytest.mark.parametrize('cls', ['PINNSF_bottleneck_multitask', 'PINNSF_bottleneck'])
@pytest.mark.parametrize('case', ['relfeat_takes_them', 'exit_flushes', 'param_has_grad', 'two_passes'])
def test_deferred_slot_sums_of_the_bottleneck_variants(cls, case):
    """The per-row decoders and the encoders of the bottleneck variants are two operators with a slot-sum launch each
    (piml_rowdecoder_bwd_acc, piml_encoder_bwd_acc): inside the block both leave their sums (PIML_DEFER_SLOT_SUMS of the two
    entries), the library merges the two descriptions (network.hip: pending_slot_sums_leave) and the relfeat backward's launch runs
    them -- three launches become one, every gradient bitwise the stand-alone launches'.  Reference: src/models/model.py:1116-1134,
    :1192-1218 under the step of src/models/simulators.py:699-779."""
    model, scene = _model(cls), _scene()
    kw = dict(through_features=case != 'exit_flushes', pre_grad=case == 'param_has_grad', twice=case == 'two_passes')
    want = _step(model, scene, False, **kw)
    got = _step(model, scene, True, **kw)
    assert len(want) == len(got) and len(want) > 10
    for i, (a, b) in enumerate(zip(want, got)):
        if i == len(want) - 1 and kw['through_features']:      # d/d(state): float atomics, order not fixed
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(a, b), f'gradient {i} differs by {(a - b).abs().max().item()}'


def test_deferred_slot_sums_inside_a_captured_graph():
    from piml_amd import ops
    import piml_amd
    if not piml_amd.hip_graphs_safe():
        pytest.skip('HIP graphs not trusted in this process')
    model, scene = _model(), _scene()
    want = _step(model, scene, False)
    state, dest, obs, v0 = scene
    st = state.clone().requires_grad_(True)
    N = st.shape[0]

    def body():
        st.grad = None
        for p in model.parameters():
            p.grad = None
        pf, of, sf = ops.relative_features_packed_self(st, dest, obs, v0, 0, N)
        acc = model(pf, of, sf)[0]
        with ops.deferred_slot_sums():
            acc.backward(torch.ones_like(acc))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    have = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(have) == len(want) - 1
    for a, b in zip(want, have):
        assert torch.equal(a, b)


@pytest.mark.parametrize('relfeat_between', [True, False])
def test_deferred_weight_pack_is_bitwise_the_standalone_launch(relfeat_between, monkeypatch):
    """PIML_DEFER_PACK (model.packed_weights() by default): the pack rides as trailing workgroups of the next relfeat forward
    launch, or -- when no such launch comes before the network -- is launched by the network's forward itself; either way the
    outputs and gradients are bitwise those of the pack as a launch of its own, also when the weights changed since the
    previous pack (a stale image would show)."""
    from piml_amd import ops
    model, scene = _model(), _scene()
    state, dest, obs, v0 = scene
    N = state.shape[0]
    base = [p.detach().clone() for p in model.parameters()]
    feats0 = ops.relative_features_packed_self(state, dest, obs, v0, 0, N)

    def one_pass(factor, mode):
        with torch.no_grad():
            for p, b in zip(model.parameters(), base):
                p.copy_(b * factor)
                p.grad = None
        import contextlib
        monkeypatch.setattr(ops, 'DEFER_PACK', mode == 'deferred')
        with (model.packed_weights() if mode != 'self' else contextlib.nullcontext()):
            feats = ops.relative_features_packed_self(state, dest, obs, v0, 0, N) if relfeat_between else feats0
            acc = model(*feats)[0]
            acc.backward(torch.ones_like(acc))
        torch.cuda.synchronize()
        return [acc.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    for factor in (1.0, 1.25, 0.5):
        want = one_pass(factor, 'self')              # no packed_weights block: the forward packs for itself
        for mode in ('standalone', 'deferred'):
            got = one_pass(factor, mode)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert torch.equal(a, b), (factor, mode)
