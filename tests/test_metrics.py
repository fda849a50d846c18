"""Evaluation metrics (reference src/functions/metrics.py:16-273): batched Sinkhorn OT / MMD / MAE
against values computed by the reference's own per-frame implementations."""
import numpy as np
import pytest
import torch

from conftest import golden


def _check(dev):
    from piml_amd.functions import metrics as M
    g = golden('metrics')
    p, q, mask = [torch.tensor(g[k], device=dev) for k in ('p', 'q', 'mask')]
    for red in ('sum', 'mean'):
        assert np.isclose(M.ot_with_time_mask(p, q, mask, reduction=red), float(g[f'ot_{red}']), rtol=1e-4)
        assert np.isclose(M.mmd_with_time_mask(p, q, mask, reduction=red), float(g[f'mmd_{red}']), rtol=1e-3)
        assert np.isclose(M.mae_with_time_mask(p, q, mask, reduction=red), float(g[f'mae_{red}']), rtol=1e-5)
    return M, g, q


def test_ot_mmd_mae_match_reference_cpu():
    _check('cpu')


@pytest.mark.gpu
def test_metrics_on_gpu_incl_collision_count():
    M, g, q = _check('cuda:0')
    for red in ('sum', 'mean'):
        assert np.isclose(M.collision_count(q, 0.6, reduction=red), float(g[f'coll_{red}']), rtol=1e-6)


def test_fde_matches_bruteforce():
    from piml_amd.functions import metrics as M
    g = golden('metrics')
    p, q, mask = [torch.tensor(g[k]) for k in ('p', 'q', 'mask')]
    want = []
    for i in range(mask.shape[1]):
        ts = np.nonzero(g['mask'][:, i] == 1)[0]
        if len(ts):
            want.append(np.linalg.norm(g['p'][ts[-1], i] - g['q'][ts[-1], i]))
    assert np.isclose(M.fde_with_time_mask(p, q, mask, 'mean'), np.mean(want), rtol=1e-5)
    assert np.isclose(M.fde_with_time_mask(p, q, mask, 'sum'), np.sum(want), rtol=1e-5)
