"""CPU-only, build container only: fuzz the oracle against the LIVE reference (imported from
/root/reference when that directory exists; skipped elsewhere, e.g. on the GPU box).  Widens the
pinning of the oracle beyond the committed golden vectors: random k, sight angles, thresholds."""
import os
import sys
import types

import numpy as np
import pytest

from conftest import bits

REF = '/root/reference/src'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')


@pytest.fixture(scope='module')
def ref_pedestrians():
    import torch  # noqa: F401
    sys.dont_write_bytecode = True
    sys.modules.setdefault('setproctitle', types.SimpleNamespace(setproctitle=lambda *_: None))
    sys.path.insert(0, REF)
    try:
        import data.data as DATA
        yield DATA.Pedestrians()
    finally:
        sys.path.remove(REF)


def test_relfeat_fuzz_oracle_equals_live_reference(oracle, ref_pedestrians):
    import torch
    rng = np.random.default_rng(7)
    for case in range(120):
        N = int(rng.integers(2, 60))
        M = int(rng.choice([2, 5, 40, 150]))
        T = int(rng.choice([1, 1, 1, 4]))
        p = (rng.random((T, N, 2)) * 7).astype(np.float32)
        v = rng.standard_normal((T, N, 2)).astype(np.float32)
        v[rng.random((T, N)) < 0.2] = 0                       # exercises the temporal heading fill when T > 1
        a = rng.standard_normal((T, N, 2)).astype(np.float32)
        d = (rng.random((T, N, 2)) * 7).astype(np.float32)
        absent = rng.random((T, N)) < 0.15
        p[absent] = np.nan
        d[absent] = np.nan
        o = (rng.random((M, 2)) * 7).astype(np.float32)
        kp, ko = int(rng.integers(1, 9)), int(rng.integers(1, 12))
        ang_p, ang_o = float(rng.choice([30, 60, 90, 100, 150, 180])), float(rng.choice([45, 90, 135]))
        dp, do = float(rng.choice([0.5, 1.5, 4, 50])), float(rng.choice([1.0, 4, 50]))
        t = [torch.tensor(x.copy()) for x in (p, v, a, d, o)]
        rpf, rof, rdf = ref_pedestrians.get_relative_features(*t, kp, ang_p, dp, ko, ang_o, do)
        pf, of, df = oracle.relfeat_fwd(p, v, a, d, o, kp, ang_p, dp, ko, ang_o, do)
        # random continuous positions: no exact distance ties, so even the slot order must agree
        assert np.array_equal(bits(pf), bits(rpf.numpy())), (case, N, M, T, kp, ang_p, dp)
        assert np.array_equal(bits(of), bits(rof.numpy())), (case, N, M, T, ko, ang_o, do)
        assert np.array_equal(bits(df), bits(rdf.numpy()))


def test_rollout_losses_equal_live_reference():
    """The rollout losses of BaseSimulator (pure torch, host side) against the reference's methods."""
    import torch
    sys.path.insert(0, REF)
    try:
        import models.simulators as RSIM
    finally:
        sys.path.remove(REF)
    from piml_amd.models.simulators import BaseSimulator as Mine
    ref = RSIM.BaseSimulator.__new__(RSIM.BaseSimulator)
    mine = Mine.__new__(Mine)
    g = torch.Generator().manual_seed(0)
    pred, lab = torch.randn(3, 6, 11, 2, generator=g), torch.randn(3, 6, 11, 2, generator=g)
    coll = (torch.rand(3, 6, 11, generator=g) > 0.8).float()
    am = (torch.rand(11, generator=g) > 0.3).float()
    for decay in (1.0, 0.9):
        for rev in (False, True):
            a = mine.multiple_rollout_mse_loss(pred, lab, decay, reduction='sum', reverse=rev)
            b = ref.multiple_rollout_mse_loss(pred, lab, decay, reduction='sum', reverse=rev)
            assert torch.allclose(a, b, rtol=1e-6)
        a = mine.multiple_rollout_collision_avoidance_loss(pred, lab, decay, reduction='none')
        b = ref.multiple_rollout_collision_avoidance_loss(pred, lab, decay, reduction='none')
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        for mask in (None, am):
            a = mine.multiple_rollout_collision_loss(pred, lab, decay, 10, coll.clone(), reduction='sum', abnormal_mask=mask)
            b = ref.multiple_rollout_collision_loss(pred, lab, decay, 10, coll.clone(), reduction='sum', abnormal_mask=mask)
            assert torch.allclose(a, b, rtol=1e-5)
    emb = torch.randn(5, 7, generator=g)
    assert torch.allclose(mine.l1_reg_loss(emb, 0.01, 'sum'), ref.l1_reg_loss(emb, 0.01, 'sum'))


def test_data_loader_equals_live_reference():
    """Same batches as the reference's loader for pointwise data under the same numpy seed."""
    import torch
    sys.path.insert(0, REF)
    try:
        import data.data as RDATA
        import utils.data_loader as RLOAD
    finally:
        sys.path.remove(REF)
    from piml_amd.data.data import PointwisePedData
    from piml_amd.utils import data_loader as LOAD
    n = 23
    feats = dict(ped_features=torch.arange(n * 2 * 6.).reshape(n, 2, 6), obs_features=torch.zeros(n, 3, 6),
                 self_features=torch.arange(n * 7.).reshape(n, 7), labels=torch.arange(n * 12.).reshape(n, 12))
    mine = PointwisePedData()
    for k, v in feats.items():
        setattr(mine, k, v)
    mine.dataset_len = n
    ref = RDATA.PointwisePedData(**feats)
    np.random.seed(5)
    a = LOAD.data_loader(mine, 4, seed=0)
    np.random.seed(5)
    b = RLOAD.data_loader(ref, 4, seed=0)
    assert len(a) == len(b) == n // 4
    for x, y in zip(a, b):
        for u, w in zip(x, y):
            assert torch.equal(u, w)
