"""Shared helpers for the parity tests: frame construction and synthetic coefficient planes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402

# shaun_the_sheep.jpeg's two quantisation tables (zigzag order); the same values the reference's
# own header test pins as literals (test/entete_test.c:201-218).  Loaded from the golden fixture.
GOLDEN = os.path.join(ROOT, "tests", "golden")


def qtables_ramp():
    qa = np.array([2 + i for i in range(64)], dtype=np.uint16)
    qb = np.array([3 + 2 * i for i in range(64)], dtype=np.uint16)
    qc = np.array([(40503 * (i + 1)) & 0xFFFF for i in range(64)], dtype=np.uint16)  # wraps int16 products
    return [qa, qb, qc, None]


def oracle_desc(width, height, sampling, idq=None, apply_iquant=True):
    """FrameDesc built WITHOUT the product library (oracle geometry), usable on CPU-only hosts."""
    d = pyoracle.FrameDesc()
    d.width, d.height, d.nbcomp = width, height, len(sampling)
    d.apply_iquant = 1 if apply_iquant else 0
    for c, (hs, vs) in enumerate(sampling[:3]):
        d.comp[c].hs, d.comp[c].vs = hs, vs
        d.comp[c].idq = idq[c] if idq is not None else (0 if c == 0 else 1)
    rc, d = pyoracle.frame_geometry(d)
    assert rc == 0, rc
    return d


def to_gpu_desc(jg, d):
    return jg.FrameDesc.from_buffer_copy(bytes(d))


def plane_shape(d, c):
    return (d.nbmcuV * d.comp[c].vs * d.nbmcuH * d.comp[c].hs, 64)


def synth_planes(d, seed, kind=0):
    return [pyoracle.synth_blocks(plane_shape(d, c)[0], seed + c, 0, kind) for c in range(d.nbcomp)]


def sha256_hex(a):
    """SHA-256 of an array's bytes (the digest the golden metadata stores)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kat_frame(blocks, layout):
    """(desc sampling, width, height, planes) of the frame that carries the adversarial stage blocks (impulses at each of
    the 64 positions x +-{1, 1024, 2047}, all-32767 / all--32768, synthetic kinds 0-2) through the FUSED path: 640 blocks
    per chroma-sized plane (the 612 vectors + zero blocks), 64 x 10 MCUs, so that 4:4:4 takes whole 64-MCU tiles and 4:2:0
    whole 32-MCU tiles.  Cb / Cr = the same blocks rotated by 211 / 422 places, Y of 4:2:0 = four rotations."""
    b = np.concatenate([blocks, np.zeros((640 - blocks.shape[0], 64), np.int16)])
    if layout == "gray":
        return ((1, 1),), 512, 80, [b]
    if layout == "444":
        return ((1, 1),) * 3, 512, 80, [b, np.roll(b, 211, 0), np.roll(b, 422, 0)]
    if layout == "420":
        y = np.concatenate([b, np.roll(b, 101, 0), np.roll(b, 307, 0), np.roll(b, 499, 0)])
        return ((2, 2), (1, 1), (1, 1)), 1024, 160, [y, np.roll(b, 211, 0), np.roll(b, 422, 0)]
    raise ValueError(layout)
