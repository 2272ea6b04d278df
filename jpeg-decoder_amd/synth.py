"""Synthetic quantised zigzag coefficients: counter-based splitmix64, block-indexed,
integer-only so that every host and every rank produces the same bytes.

DC ~ U[-64, 64]; AC[k] is zero with probability min(0.95, k/48), else a two-sided geometric
(discrete Laplace) magnitude with scale ~ 24/(1+k).  kind 1: uniform +-1024; kind 2: full-range
int16 (exercises the iquant wrap).  tests/test_synth.py pins this against the oracle's C generator.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synth_blocks(n, seed, first_block=0, kind=0, chunk=1 << 16):
    """(n, 64) int16 array of blocl16_t records."""
    out = np.empty((n, 64), dtype=np.int16)
    k = np.arange(64, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3)
    pz = np.minimum((np.arange(64) * 256) // 48, 243).astype(np.uint64)[None, :]
    scale = (4259 // (1 + np.arange(64))).astype(np.uint64)[None, :]
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        gb = (np.arange(s, e, dtype=np.uint64) + np.uint64(first_block))[:, None]
        with np.errstate(over="ignore"):
            h = _splitmix64(base + gb * np.uint64(64) + k)
        if kind == 2:
            v = (h & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16).astype(np.int64)
        elif kind == 1:
            v = (h % np.uint64(2049)).astype(np.int64) - 1024
        else:
            u = ((h >> np.uint64(32)) | np.uint64(1)).astype(np.uint64)
            # count leading zeros of the 32-bit value u, exactly (frexp exponent of an exact double)
            g = (32 - np.frexp(u.astype(np.float64))[1]).astype(np.uint64)
            frac = (h >> np.uint64(8)) & np.uint64(0xFF)
            mag = (((g * np.uint64(256) + frac) * scale) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)
            neg = (h & np.uint64(1 << 16)) != 0
            v = np.where(neg, -mag.astype(np.int64), mag.astype(np.int64))
            v = np.where((h & np.uint64(0xFF)) < pz, 0, v)
            dc = (h[:, 0] % np.uint64(129)).astype(np.int64) - 64
            v[:, 0] = dc
        out[s:e] = v.astype(np.int16)
    return out


# shaun_the_sheep.jpeg's luma / chroma quantisation tables in zigzag order: the data the reference's
# header test pins (test/entete_test.c:201-218) and tests/golden/shaun_the_sheep_planes.npz holds;
# the bench's qtables.
SHAUN_QT_LUMA = np.array([
    10, 7, 7, 9, 7, 6, 10, 9, 8, 9, 11, 11, 10, 12, 15, 25, 16, 15, 14, 14, 15, 30, 22, 23, 18, 25, 36, 32, 38, 37, 35, 32,
    35, 34, 40, 45, 57, 48, 40, 42, 54, 43, 34, 35, 50, 68, 50, 54, 59, 61, 64, 64, 64, 38, 48, 70, 75, 69, 62, 74, 57, 63,
    64, 61], dtype=np.uint16)
SHAUN_QT_CHROMA = np.array([11, 11, 11, 15, 13, 15, 29, 16, 16, 29, 61, 41, 35, 41] + [61] * 50, dtype=np.uint16)


def shaun_qtables():
    return [SHAUN_QT_LUMA, SHAUN_QT_CHROMA, None, None]
