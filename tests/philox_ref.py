"""numpy restatement of the dropout keep-mask stream (piml_amd/csrc/philox.hpp, dropout.hip, encoder_x3.hip) for the
tests: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11),
counter = (offset lo, offset hi, row, (stream << 16) | sub), key = (seed lo, seed hi); p = 0.5: keep word w = output
word w & 3 of call sub = 0xFFFF - (w >> 2); other p: 16 bits per feature, kept iff >= round(p * 65536).
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


def keep_bits(seed, offset, rows, cols, p, stream=0):
    """int32 (rows, ceil(cols / 32)): the mask draw number `offset` of piml_dropout_keep_bits yields (philox.hpp)."""
    words = (cols + 31) // 32
    row = np.arange(rows, dtype=np.uint64)[:, None]
    k0, k1 = seed & 0xffffffff, (seed >> 32) & 0xffffffff
    o0, o1 = np.uint64(offset & 0xffffffff), np.uint64(offset >> 32)
    if np.float32(p) == np.float32(0.5):                       # fair bits: one call per 128 features
        calls = (words + 3) // 4
        sub = (np.uint64(stream << 16) | (np.uint64(0xFFFF) - np.arange(calls, dtype=np.uint64)))[None, :]
        zero = np.zeros((rows, calls), dtype=np.uint64)
        out = philox4x32_10(zero + o0, zero + o1, zero + row, zero + sub, k0, k1)
        w = np.stack(out, -1).reshape(rows, -1)[:, :words].astype(np.uint32)
    else:                                                      # 16 bits per feature
        thresh = int(float(np.float32(p)) * 65536.0 + 0.5)
        calls = words * 4
        sub = (np.uint64(stream << 16) | np.arange(calls, dtype=np.uint64))[None, :]
        zero = np.zeros((rows, calls), dtype=np.uint64)
        out = philox4x32_10(zero + o0, zero + o1, zero + row, zero + sub, k0, k1)
        u32 = np.stack(out, -1).astype(np.uint64)                                  # (rows, calls, 4)
        u16 = np.stack([u32 & np.uint64(0xFFFF), u32 >> np.uint64(16)], -1)        # (rows, calls, 4, 2): feature 8 call + 2 word + half
        keep = (u16.reshape(rows, -1) >= np.uint64(thresh)).astype(np.uint64).reshape(rows, words, 32)
        w = (keep << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    left = cols - 32 * np.arange(words)
    mask = np.where(left >= 32, 0xFFFFFFFF, (1 << np.clip(left, 0, 31)) - 1).astype(np.uint32)
    return (w & mask[None, :]).view(np.int32)


def keep_mask(seed, offset, rows, cols, p, stream=0):
    """bool (rows, cols)."""
    b = keep_bits(seed, offset, rows, cols, p, stream).view(np.uint32).astype(np.uint64)
    k = (b[:, :, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)
    return k.reshape(rows, -1)[:, :cols].astype(bool)
