"""numpy restatement of piml_dropout_keep_bits (piml_amd/csrc/dropout.hip) for the tests: Philox4x32-10 (Salmon, Moraes,
Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter = (offset lo, offset hi, row, c >> 2),
key = (seed lo, seed hi); feature c takes output word c & 3 and is kept iff word >= round(p * 2^32).
tests/test_dropout.py pins `philox4x32_10` on the known-answer vectors of the Random123 distribution."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xffffffff)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """uint32 arrays (broadcastable) -> four uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return [x.astype(np.uint32) for x in c]


def keep_mask(seed, offset, rows, cols, p):
    """bool (rows, cols): the mask call number `offset` of piml_dropout_keep_bits draws."""
    thresh = min(int(np.float64(np.float32(p)) * 4294967296.0 + 0.5), 1 << 32)
    row = np.arange(rows, dtype=np.uint64)[:, None]
    grp = np.arange((cols + 3) // 4, dtype=np.uint64)[None, :]
    zero = np.zeros((rows, grp.shape[1]), dtype=np.uint64)
    out = philox4x32_10(zero + np.uint64(offset & 0xffffffff), zero + np.uint64(offset >> 32), zero + row, zero + grp,
                        seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    u = np.stack(out, -1).reshape(rows, -1)[:, :cols].astype(np.uint64)
    return u >= np.uint64(thresh)


def keep_bits(seed, offset, rows, cols, p):
    """int32 (rows, ceil(cols / 32)) in the kernel's layout."""
    k = keep_mask(seed, offset, rows, cols, p)
    words = (cols + 31) // 32
    pad = np.zeros((rows, words * 32 - cols), dtype=bool)
    k = np.concatenate([k, pad], 1).reshape(rows, words, 32).astype(np.uint64)
    w = (k << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    return w.view(np.int32)
