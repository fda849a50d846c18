"""GPU: `--model pinnsf_pb` / `pinnsf_pbc` (reference src/models/model.py:1307-1540) and the hand-written
collision post-correction against goldens captured from the reference
(tests/golden/model_polar.npz) and against the CPU oracle."""
import types

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def model_args(**kw):
    a = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3,
        processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5, activation='relu',
        dataset_name='ucy', res_hidden_layers=3, correction_hidden_layers=1, time_unit=0.08,
        collision_threshold=0.5)
    a.__dict__.update(kw)
    return a


def load(name):
    import piml_amd.models.model as MODEL
    g = golden('model_polar')
    cls = {'pinnsf_pb': MODEL.PINNSF_polar_bottleneck, 'pinnsf_pbc': MODEL.PINNSF_polar_bottleneck_collision}[name]
    m = cls(model_args()).eval()
    sd = {k[len(name) + 4:]: torch.tensor(g[k]) for k in g.files if k.startswith(name + '/sd/')}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), g


@pytest.mark.parametrize('tag', ['n', 'c', 'd'])
def test_collision_post_correction_matches_reference_and_oracle(tag, oracle):
    from piml_amd import ops
    g = golden('model_polar')
    ped, sf = g[f'in_{tag}/ped'], g[f'in_{tag}/selff']
    pre, ref = g[f'pinnsf_pbc/pre_{tag}'], g[f'pinnsf_pbc/out_{tag}0']
    got = ops.collision_post_correction(torch.tensor(pre, device=DEV), torch.tensor(ped, device=DEV),
                                        torch.tensor(np.ascontiguousarray(sf[..., 2:4]), device=DEV), 0.5, 0.08).cpu().numpy()
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() <= 1e-5 * scale
    want = oracle.collision_post_correction(pre, ped, sf[..., 2:4], 0.5, 0.08)
    assert np.abs(got - want).max() <= 1e-5 * scale


def torch_correction(P, ped, vi, thr=0.5, dt=0.08):
    """torch-op restatement of model.py:1383-1444 (autograd reference for the analytic backward)."""
    R = thr + 1.34 * 2 * dt
    pji = torch.nan_to_num(ped[..., :2], nan=0.0)
    norm = torch.norm(pji, p=2, dim=-1) + 1e-6
    nji = pji / norm.unsqueeze(-1)
    vji = ped[..., 2:4]
    vik = vi.unsqueeze(-2).expand_as(vji)
    vj = vji + vik
    coll = ((R >= norm) & (norm > 1e-4)).float()
    inter = ((vik * pji).sum(-1) * (vj * (-pji)).sum(-1)).detach()
    inter = (torch.nan_to_num(inter) > 0).float()
    enc, chase = coll * inter, coll * (1 - inter)

    def nearest(flag):
        d = (norm * flag).detach()
        d = torch.where(d < 1e-4, d + 100, d)
        idx = d.min(dim=-1).indices[..., None, None].expand(*d.shape[:-1], 1, 2)
        return torch.gather(nji, -2, idx).squeeze(-2), torch.gather(vji, -2, idx).squeeze(-2)
    n, _ = nearest(enc)
    m = (enc.sum(-1, keepdim=True) > 0).float()
    a = -(vi * n).sum(-1, keepdim=True) * n / dt * m
    P_ = P * m
    s = (P_ * n).sum(-1, keepdim=True)
    s = s * (s > 0)
    P = P + (P_ - s * n + a)
    n, w = nearest(chase)
    m = (chase.sum(-1, keepdim=True) > 0).float()
    q = (w * n).sum(-1, keepdim=True)
    a = q * (q < 0) * n / dt * m
    P_ = P * m
    s = (P_ * n).sum(-1, keepdim=True)
    s = s * (s > 0) * (q < 0)
    return P + (P_ - s * n + a)


@pytest.mark.parametrize('tag', ['c', 'd'])
def test_collision_post_correction_backward(tag):
    from piml_amd import ops
    g = golden('model_polar')
    ped = torch.tensor(g[f'in_{tag}/ped'], device=DEV)
    vi = torch.tensor(np.ascontiguousarray(g[f'in_{tag}/selff'][..., 2:4]), device=DEV)
    pre = torch.tensor(g[f'pinnsf_pbc/pre_{tag}'], device=DEV)
    w = torch.linspace(-1, 1, pre.numel(), device=DEV).view_as(pre)
    a = [x.clone().requires_grad_(True) for x in (pre, ped, vi)]
    ref = torch_correction(*a)
    g_ref = torch.autograd.grad((ref * w).sum(), a)
    b = [x.clone().requires_grad_(True) for x in (pre, ped, vi)]
    out = ops.collision_post_correction(*b, 0.5, 0.08)
    g_out = torch.autograd.grad((out * w).sum(), b)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    for x, y in zip(g_out, g_ref):
        scale = max(1.0, float(y.abs().max()))
        assert float((x - y).abs().max()) <= 1e-4 * scale, (float((x - y).abs().max()), scale)


@pytest.mark.parametrize('name', ['pinnsf_pb', 'pinnsf_pbc'])
@pytest.mark.parametrize('tag', ['n', 'c', 'd'])
def test_polar_models_match_reference(name, tag):
    m, g = load(name)
    with torch.no_grad():
        outs = m(*[torch.tensor(g[f'in_{tag}/{k}'], device=DEV) for k in ('ped', 'obs', 'selff')])
    q = 0
    while f'{name}/out_{tag}{q}' in g.files:
        ref = g[f'{name}/out_{tag}{q}']
        got = outs[q].cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max()), (q, np.abs(got - ref).max())
        q += 1
    assert q == len(outs)


@pytest.mark.parametrize('name', ['pinnsf_pb', 'pinnsf_pbc'])
def test_polar_models_gradients_match_reference(name):
    """d(sum w * acceleration)/d(inputs, weights) on the dense scene against the reference's autograd."""
    m, g = load(name)
    ins = [torch.tensor(g[f'in_d/{k}'], device=DEV).requires_grad_(True) for k in ('ped', 'obs', 'selff')]
    res = m(*ins)
    w = torch.linspace(-1.0, 1.0, res[0].numel(), device=DEV).view_as(res[0])
    (res[0] * w).sum().backward()
    for k, x in zip(('ped', 'obs', 'selff'), ins):
        ref = g[f'{name}/grad_d/{k}']
        got = torch.nan_to_num(x.grad).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), (k, np.abs(got - ref).max(), np.abs(ref).max())
    named = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(f'{name}/grad_d/param/'):
            ref = g[k]
            got = named[k[len(name) + 14:]].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), k


def test_simulator_builds_polar_models():
    from piml_amd.models.simulators import BaseSimulator
    from test_simulator_gpu import sim_args
    for name, cls in (('pinnsf_pb', 'PINNSF_polar_bottleneck'), ('pinnsf_pbc', 'PINNSF_polar_bottleneck_collision')):
        sim = BaseSimulator(sim_args(model=name, time_unit=0.08))
        assert type(sim.model).__name__ == cls
