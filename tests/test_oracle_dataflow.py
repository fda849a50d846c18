"""CPU: the reference-dataflow restatement (oracle/dataflow.py, dense N x M tensors + full sort) against the C
oracle on seeded scenes -- the second form of bench.py's cpu_baseline must compute the same step."""
import numpy as np
import pytest
import torch

from piml_amd.scenes import synthetic_gc_scene


@pytest.mark.parametrize('n,m,seed', [(200, 100, 0), (333, 500, 1), (64, 2, 2)])
def test_dataflow_equals_oracle(oracle, n, m, seed):
    from oracle import dataflow
    sc = synthetic_gc_scene(n, m, seed=seed)
    rng = np.random.default_rng(seed)
    acc = (rng.standard_normal((n, 2)) * 0.2).astype(np.float32)
    t = lambda x: torch.tensor(x)
    pf, of, df = dataflow.relative_features(t(sc['position']), t(sc['velocity']), t(acc), t(sc['destination']),
                                            t(sc['obstacles']))
    ref = oracle.relfeat_fwd(sc['position'][None], sc['velocity'][None], acc[None], sc['destination'][None],
                             sc['obstacles'])
    for got, want in zip((pf, of, df), ref[:3]):
        got, want = got.numpy(), want[0]
        assert got.shape == want.shape
        # identical arithmetic; rows may swap only between neighbours at bit-equal distance (torch.sort is not stable)
        same = np.isclose(got, want, rtol=0, atol=0) | (np.isnan(got) & np.isnan(want))
        assert same.mean() > 0.999, same.mean()
        bad_rows = ~same.reshape(same.shape[0], -1).all(-1)
        for r in np.nonzero(bad_rows)[0]:
            a = np.sort(np.linalg.norm(got[r].reshape(-1, got.shape[-1])[:, :2], axis=-1))
            b = np.sort(np.linalg.norm(want[r].reshape(-1, want.shape[-1])[:, :2], axis=-1))
            assert np.allclose(a, b, rtol=1e-6, atol=1e-6)
