"""Host-side checks of round-3 device logic that need no GPU: the counter-based dropout mixer (realise_amd/csrc/common.h rng_hash4 /
drop_mult) restated in numpy - keep rate, uniformity of the 16-bit lanes, independence along a row, down a column, across quads and
across seeds (the reference uses torch's Philox dropout, transformers/modeling_bert.py:190,253,276,342; only the statistics can match)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def rng_hash4(seed, quad):
    seed, quad = np.uint64(seed), quad.astype(np.uint64)
    m1 = ((quad ^ seed) & M32) * np.uint64(0x9E3779B1)
    x = ((m1 & M32) ^ (m1 >> np.uint64(32)) ^ ((seed * np.uint64(0x632BE5AB)) & M32)) & M32
    m2 = x * np.uint64(0x85EBCA77)
    m3 = ((x ^ np.uint64(0x27D4EB2F)) & M32) * np.uint64(0xC2B2AE3D)
    return ((m2 >> np.uint64(32)) ^ (m3 & M32)) & M32, ((m3 >> np.uint64(32)) ^ (m2 & M32)) & M32


def lanes16(seed, n):
    o0, o1 = rng_hash4(seed, np.arange(n // 4, dtype=np.uint64))
    return np.stack([o0 & np.uint64(0xFFFF), o0 >> np.uint64(16), o1 & np.uint64(0xFFFF), o1 >> np.uint64(16)], 1).reshape(-1).astype(np.int64)


def test_dropout_mixer_statistics():
    n = 1 << 21
    t16 = int(0.1 * 2 ** 32) >> 16
    for seed in (1, 77, 0xDEADBEEF, 20240917):
        v = lanes16(seed, n)
        keep = (v >= t16).astype(np.float64)
        assert abs(keep.mean() - 0.9) < 1.5e-3
        for b in (v >> 8, v & 255):                                   # 255 degrees of freedom: mean 255, sigma 22.6
            c = np.bincount(b, minlength=256)
            assert ((c - n / 256) ** 2 / (n / 256)).sum() < 255 + 5 * 22.6
        for lag in (1, 2, 3, 4, 5, 128, 768, 16384):                  # neighbours in a quad, next quad, next row of S / H, far
            assert abs(np.corrcoef(keep[:-lag], keep[lag:])[0, 1]) < 4e-3
    a, b = (lanes16(100, n) >= t16).astype(float), (lanes16(101, n) >= t16).astype(float)
    assert abs(np.corrcoef(a, b)[0, 1]) < 4e-3                         # consecutive step seeds
    k = (lanes16(5, 128 * 8192) >= t16).reshape(8192, 128)
    assert k.mean(0).min() > 0.885 and k.mean(0).max() < 0.915        # every key column of an attention row keeps ~0.9
