"""Batching (src/utils/data_loader.py:14-53).  Like the reference, pointwise batches are drawn
from one global-RNG permutation (`np.random.permutation`; the `seed` / `shuffle` arguments are
accepted and ignored there, quirk Q9) and channelled data is cut into consecutive slices."""
import numpy as np

from ..data.data import ChanneledTimeIndexedPedData, PointwisePedData


def make_batch(train_ids, batch_size, seed, shuffle=True, drop_last=True):
    n = len(train_ids)
    if shuffle:
        train_ids = train_ids[np.random.permutation(n)]
    batches = [train_ids[i * batch_size:(i + 1) * batch_size] for i in range(n // batch_size)]
    if not drop_last:
        batches.append(train_ids[n - n % batch_size:])
    return batches


def data_loader(data, batch_size, seed, shuffle=True, drop_last=True):
    if isinstance(data, PointwisePedData):
        return [data[idx] for idx in make_batch(np.arange(len(data)), batch_size, seed)]
    if isinstance(data, list):
        loaders = []
        for d in data:
            for i in range(d.dataset_len // batch_size):
                loaders.append(ChanneledTimeIndexedPedData.slice(d, list(range(i * batch_size, (i + 1) * batch_size))))
        return loaders
    raise NotImplementedError
