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


# ---- the same generator on a torch device (bench.py: gigabytes of coefficients without minutes of numpy) -------------
def _lsr(x, n):
    """logical right shift of an int64 tensor holding a uint64 bit pattern"""
    return (x >> n) & ((1 << (64 - n)) - 1)


def _splitmix64_t(x):
    x = x + (0x9E3779B97F4A7C15 - (1 << 64))           # int64 arithmetic wraps like uint64
    x = (x ^ _lsr(x, 30)) * (0xBF58476D1CE4E5B9 - (1 << 64))
    x = (x ^ _lsr(x, 27)) * (0x94D049BB133111EB - (1 << 64))
    return x ^ _lsr(x, 31)


def synth_blocks_torch(n, seed, first_block=0, kind=0, device="cuda", chunk=1 << 20):
    """synth_blocks() computed on `device`: an (n, 64) int16 tensor with the same bytes (tests/test_synth.py)."""
    import torch
    out = torch.empty((n, 64), dtype=torch.int16, device=device)
    k = torch.arange(64, dtype=torch.int64, device=device)[None, :]
    base = (int(seed) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    if base >= 1 << 63:
        base -= 1 << 64
    pz = torch.clamp((torch.arange(64, dtype=torch.int64, device=device) * 256) // 48, max=243)[None, :]
    scale = (4259 // (1 + torch.arange(64, dtype=torch.int64, device=device)))[None, :]
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        gb = (torch.arange(s, e, dtype=torch.int64, device=device) + int(first_block))[:, None]
        h = _splitmix64_t(base + gb * 64 + k)
        if kind == 2:
            v = (h & 0xFFFF).to(torch.int64)
            v = torch.where(v >= 0x8000, v - 0x10000, v)
        elif kind == 1:
            hi, lo = _lsr(h, 32), h & 0xFFFFFFFF
            v = ((hi % 2049) * ((1 << 32) % 2049) + lo % 2049) % 2049 - 1024
        else:
            u = _lsr(h, 32) | 1
            g = 32 - torch.frexp(u.to(torch.float64))[1].to(torch.int64)      # leading zeros of the 32-bit value
            frac = _lsr(h, 8) & 0xFF
            mag = (((g * 256 + frac) * scale) & 0xFFFFFFFF) >> 16
            v = torch.where((h & (1 << 16)) != 0, -mag, mag)
            v = torch.where((h & 0xFF) < pz, torch.zeros_like(v), v)
            h0 = h[:, 0]
            hi, lo = _lsr(h0, 32), h0 & 0xFFFFFFFF
            v[:, 0] = ((hi % 129) * ((1 << 32) % 129) + lo % 129) % 129 - 64
        out[s:e] = v.to(torch.int16)
    return out
