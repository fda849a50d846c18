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
