"""Oracle (TEST INFRASTRUCTURE): Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011 - "Parallel random numbers: as easy as 1, 2, 3";
constants M0 = 0xD2511F53, M1 = 0xCD9E8D57, W0 = 0x9E3779B9, W1 = 0xBB67AE85) in numpy, and the dropout mask the HIP training step
derives from it (csrc/train.hip dropout_kernel): element e = m*K + k of stream `stream_id` is kept iff word (e & 3) of
philox(counter = (e >> 2 lo, e >> 2 hi, stream_id, 0), key = (seed lo, seed hi)) >= p * 2^32.
Pinned against the published known-answer vectors of the Random123 distribution (tests/test_oracle_golden.py)."""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter arrays (uint64 arrays holding 32-bit values); keys are python ints. Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def dropout_keep(M: int, K: int, p: float, seed: int, stream_id: int) -> np.ndarray:
    """bool [M, K]: the keep mask of mc_dropout_bf16."""
    e = np.arange(M * K, dtype=np.uint64)
    e4 = e >> np.uint64(2)
    w = philox4x32_10(e4 & MASK, e4 >> np.uint64(32), np.full_like(e4, stream_id), np.zeros_like(e4), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=1)                                   # [n, 4]
    r = words[np.arange(M * K), (e & np.uint64(3)).astype(np.int64)]
    thr = min(int(p * 4294967296.0), 4294967295)
    return (r >= np.uint32(thr)).reshape(M, K)
