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
