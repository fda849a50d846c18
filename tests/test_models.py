"""CPU-only: the PINNSF mirrors load the reference's state_dicts (same keys/shapes) and
reproduce the reference's eval-mode outputs for (N,.) and channelled (C,N,.) inputs."""
import types

import numpy as np
import pytest
import torch

from conftest import golden

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


@pytest.mark.parametrize('name', sorted(CASES))
def test_model_matches_reference(name):
    import piml_amd.models.model as MODEL
    g = golden('model')
    cls, kw = CASES[name]
    m = getattr(MODEL, cls)(model_args(**kw)).eval()
    sd = {k[len(name) + 4:]: torch.tensor(g[k]) for k in g.files if k.startswith(name + '/sd/')}
    assert set(sd) == set(m.state_dict()), set(sd) ^ set(m.state_dict())
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        for tag, keys in (('n', ('ped', 'obs', 'selff')), ('c', ('pedc', 'obsc', 'selfc'))):
            outs = m(*[torch.tensor(g[k]) for k in keys])
            q = 0
            while f'{name}/out_{tag}{q}' in g.files:
                ref = g[f'{name}/out_{tag}{q}']
                got = outs[q].numpy()
                assert got.shape == ref.shape
                scale = max(1.0, np.abs(ref).max())
                assert np.abs(got - ref).max() <= 2e-5 * scale, (tag, q, np.abs(got - ref).max())
                q += 1
            assert q == len(outs)


def test_param_count_default():
    import piml_amd.models.model as MODEL
    m = MODEL.PINNSF_multitask(model_args())
    assert sum(p.numel() for p in m.parameters()) == 134277


@pytest.mark.parametrize('tag', ['n', 'd'])
def test_polar_bottleneck_matches_reference(tag):
    """`--model pinnsf_pb` (model.py:1447-1535) on (N, .) inputs; channelled input needs the GPU heading fill."""
    import piml_amd.models.model as MODEL
    g = golden('model_polar')
    m = MODEL.PINNSF_polar_bottleneck(model_args(time_unit=0.08, collision_threshold=0.5)).eval()
    sd = {k[len('pinnsf_pb/sd/'):]: torch.tensor(g[k]) for k in g.files if k.startswith('pinnsf_pb/sd/')}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        outs = m(*[torch.tensor(g[f'in_{tag}/{k}']) for k in ('ped', 'obs', 'selff')])
    for q, o in enumerate(outs):
        ref = g[f'pinnsf_pb/out_{tag}{q}']
        assert np.abs(o.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('k,scale', [(6, 2.0), (10, 1.0)])
def test_pooled_inference_fold_is_the_reference_algebra(k, scale):
    """The inference forward sums the neighbour axis BEFORE the encoder's last layer and folds that layer into the decoder's first
    (ops.pooled_h2_decoder_weights, PIML_POOL_H2).  On the reference's own modules (model.py:1271-1283: encoder -> processor
    scale -> sum over k -> decoder): decoder_1(sum_r scale (W3 h2_r + b3)) == W' sum_r h2_r + b', in float64 to 1e-12."""
    import piml_amd.models.model as M
    from piml_amd import ops
    torch.manual_seed(3)
    net = getattr(M, 'PINNSF_multitask')(model_args()).double()
    enc = [t for lin in net.obs_encoder.mlp[0::2] for t in (lin.weight, lin.bias)]
    dec = [t for lin in net.obs_decoder.mlp[0::2] for t in (lin.weight, lin.bias)] + [net.obs_predictor.mlp[0].weight,
                                                                                       net.obs_predictor.mlp[0].bias]
    h2 = torch.randn(37, k, 128, dtype=torch.float64).relu()
    msgs = scale * (h2 @ enc[4].T + enc[5])                         # processor output per row (eval mode: no dropout)
    want = msgs.sum(dim=-2) @ dec[0].T + dec[1]                     # the decoder's first layer on the pooled messages
    w1c, b1c, *rest = ops.pooled_h2_decoder_weights(enc, dec, scale, k)
    assert w1c.dtype == torch.float32 and tuple(w1c.shape) == (64, 128) and tuple(b1c.shape) == (64,)
    assert all(a is b for a, b in zip(rest, dec[2:]))               # everything behind the first layer is untouched
    w64 = scale * (dec[0] @ enc[4])
    b64 = dec[1] + scale * k * (dec[0] @ enc[5])
    got = h2.sum(dim=-2) @ w64.T + b64
    assert (got - want).abs().max() <= 1e-12 * want.abs().max()
    assert (w1c.double() - w64).abs().max() <= 6e-8 * w64.abs().max() and (b1c.double() - b64).abs().max() <= 6e-8 * b64.abs().max()
