"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the product's dropout mask
(`dfd-clip_amd/csrc/dropout.hpp`): Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
SC'11 — the published constants and round function) keyed on (seed, site), counter (group, step); eight 16-bit
draws per block; element e is kept iff draw(e) >= round(p * 65536); kept values are scaled by
65536 / (65536 - round(p * 65536)).

The reference itself uses `torch.nn.Dropout` (src/models.py:163, :294, :304, :804-912), whose masks come from
torch's global generator and are not reproducible across implementations, so parity of the
train-mode path is checked GIVEN the mask: tests regenerate the product's mask here and feed it to the oracle.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint64 arrays holding 32-bit values); returns four uint64 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & U32 for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & U32
        hi1, lo1 = p1 >> np.uint64(32), p1 & U32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & U32, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & U32, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def threshold(p):
    return min(65535, int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))


def multiplier(n, p, seed, step, site):
    """float32 array [n]: 0 where element e is dropped, 65536/(65536 - thr) where it is kept."""
    thr = threshold(p)
    if thr == 0:
        return np.ones(n, dtype=np.float32)
    groups = (n + 7) // 8
    g = np.arange(groups, dtype=np.uint64)
    seed, step = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFFFFFFFFFF
    k0 = (seed & 0xFFFFFFFF) ^ ((int(site) * 0x9E3779B9) & 0xFFFFFFFF)
    k1 = seed >> 32
    w = philox4x32_10(g & U32, g >> np.uint64(32), np.full(groups, step & 0xFFFFFFFF, dtype=np.uint64),
                      np.full(groups, step >> 32, dtype=np.uint64), k0, k1)
    draws = np.empty((groups, 8), dtype=np.uint64)
    for j in range(4):
        draws[:, 2 * j] = w[j] & np.uint64(0xFFFF)
        draws[:, 2 * j + 1] = w[j] >> np.uint64(16)
    keep = draws.reshape(-1)[:n] >= np.uint64(thr)
    return np.where(keep, np.float32(65536.0) / np.float32(65536 - thr), np.float32(0.0)).astype(np.float32)


def make_dropper(p, seed, step):
    """-> drop(site, tensor, div=1): tensor * mask of that site (probability p / div), flat element order."""
    import torch

    def drop(site, t, div=1):
        q = p / div
        if q <= 0:
            return t
        m = torch.from_numpy(multiplier(t.numel(), q, seed, step, site)).view(t.shape)
        return t * m.to(t.dtype)
    return drop
