import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')      # see piml_amd/__init__.py (before any GPU call)
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


def bits(x):
    """Bit pattern of a float32 array (for bit-exact comparisons; NaN-safe)."""
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def canon_idx(idx, dist, thr=None):
    """Neighbour indices with dead slots set to -1 and groups of exactly tied distances
    sorted by index, so that two selections can be compared irrespective of how an
    unstable sort ordered exact ties."""
    idx = np.array(idx, dtype=np.int64)
    dist = np.array(dist, dtype=np.float32)
    dead = (idx < 0) | ~np.isfinite(dist) if thr is None else ~(dist <= thr)
    idx[dead] = -1
    dist = np.where(dead, np.inf, dist)
    flat_i = idx.reshape(-1, idx.shape[-1])
    flat_d = dist.reshape(-1, idx.shape[-1])
    out = np.empty_like(flat_i)
    for r in range(flat_i.shape[0]):
        order = np.lexsort((flat_i[r], flat_d[r]))
        out[r] = flat_i[r][order]
    return out.reshape(idx.shape)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.build()
    return O
