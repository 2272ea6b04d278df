"""CPU suite: pins the oracle (oracle/jpeg_oracle.c) against
  * the reference's own known-answer tests (test/iqzz_test.c, test/idct_opt_test.c), restated,
  * the reference's golden files valid_out/invader.pgm and valid_out/shaun_the_sheep.ppm,
  * vectors generated from the compiled reference (oracle/make_golden.py -> tests/golden/).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers
from oracle import pyoracle

orc = pyoracle.oracle
G = helpers.GOLDEN
QT = helpers.qtables_ramp()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# literal table of test/iqzz_test.c:21-29 (ZZ[row][col] = zigzag index)
ZZ_TABLE = np.array([
    [0, 1, 5, 6, 14, 15, 27, 28], [2, 4, 7, 13, 16, 26, 29, 42], [3, 8, 12, 17, 25, 30, 41, 43],
    [9, 11, 18, 24, 31, 40, 44, 53], [10, 19, 23, 32, 39, 45, 52, 54], [20, 22, 33, 38, 46, 51, 55, 60],
    [21, 34, 37, 47, 50, 56, 59, 61], [35, 36, 48, 49, 57, 58, 62, 63]], dtype=np.int16)


def test_izz_known_answer():
    # test/iqzz_test.c:15-45: line[i] = i, res->data[j][i] == ref[i][j]
    res = orc.izz_blocks(np.arange(64, dtype=np.int16)[None, :])[0]
    assert np.array_equal(res.T, ZZ_TABLE)


def test_iquant_identity():
    # test/iqzz_test.c:48-64: q[i] = i, in[i] = 2(i+1)
    q = np.arange(64, dtype=np.uint16)
    line = ((np.arange(64) + 1) * 2).astype(np.int16)
    res = orc.iquant_blocks(line[None, :], 0, 63, q)[0]
    assert np.array_equal(res.astype(np.int32), q.astype(np.int32) * line)


def test_iquant_wraps_int16():
    blk = np.full((1, 64), 300, dtype=np.int16)
    q = np.full(64, 300, dtype=np.uint16)
    res = orc.iquant_blocks(blk, 0, 63, q)[0]
    assert np.all(res == np.int16((300 * 300) & 0xFFFF if (300 * 300) & 0x8000 == 0 else ((300 * 300) & 0xFFFF) - 65536))


def slow_idct(blocks):
    """Naive float IDCT of src/idct.c:50-71 restated in numpy (blocks[n][col][row])."""
    n = np.arange(8)
    cosm = np.cos((2 * n[:, None] + 1) * n[None, :] * np.pi / 16)  # [x][u]
    cu = np.where(n == 0, 1 / np.sqrt(2), 1.0)
    a = blocks.astype(np.float64) * cu[None, :, None] * cu[None, None, :]
    out = 0.25 * np.einsum("xu,yv,nuv->nxy", cosm, cosm, a) + 128
    return np.clip(out, 0, 255)


def test_idct_opt_within_one_of_slow_idct():
    # test/idct_opt_test.c:13-49: 10 000 random blocks of values 0..255, tolerance 1
    rng = np.random.default_rng(1)
    blocks = rng.integers(0, 256, size=(10000, 8, 8)).astype(np.int16)
    fast = orc.idct_opt_blocks(blocks).astype(np.float64)
    slow = slow_idct(blocks)
    assert np.max(np.abs(fast - np.floor(slow))) <= 1


def test_stage_vectors_from_reference():
    v = np.load(os.path.join(G, "stage_vectors.npz"))
    for name in ("ramp", "wrap"):
        dq = orc.iquant_blocks(v["blocks"], 0, 63, v[f"q_{name}"])
        assert np.array_equal(dq, v[f"iquant_{name}"])
        zz = orc.izz_blocks(dq)
        assert np.array_equal(zz, v[f"izz_{name}"])
        assert np.array_equal(orc.idct_opt_blocks(zz), v[f"idct_{name}"])
    assert np.array_equal(orc.iquant_blocks(v["blocks"], 5, 20, v["q_ramp"]), v["iquant_band_5_20"])


def test_ycc2rgb_every_input():
    meta = json.load(open(os.path.join(G, "meta.json")))
    idx = np.arange(1 << 24, dtype=np.uint32)
    rgb = orc.ycc2rgb_pixels(idx & 255, (idx >> 8) & 255, (idx >> 16) & 255)
    assert sha(rgb) == meta["ycc2rgb_all_sha256"]
    v = np.load(os.path.join(G, "ycc2rgb_vectors.npz"))
    assert np.array_equal(rgb[v["index"]], v["rgb"])


def frame_cases():
    meta = json.load(open(os.path.join(G, "meta.json")))
    return sorted(meta["frames"].items())


@pytest.mark.parametrize("key,info", frame_cases())
def test_frame_vectors_from_reference(key, info):
    d = helpers.oracle_desc(info["width"], info["height"], [tuple(s) for s in info["sampling"]], idq=info["idq"],
                            apply_iquant=bool(info["apply_iquant"]))
    planes = helpers.synth_planes(d, info["seed"], info["kind"])
    out = orc.backend_run(d, planes, QT)
    assert out.shape[0] == info["rows"]
    assert sha(out) == info["sha256"]
    small = np.load(os.path.join(G, "frame_vectors_small.npz"))
    if key in small.files:
        assert np.array_equal(out, small[key])


@pytest.mark.parametrize("stem,gold", [("invader_normal", "invader.pgm"), ("shaun_the_sheep", "shaun_the_sheep.ppm")])
def test_reference_golden_files(stem, gold):
    """valid_out/<gold> byte for byte from the coefficient planes the reference's VLD decodes."""
    z = np.load(os.path.join(G, stem + "_planes.npz"))
    d = pyoracle.FrameDesc.from_buffer_copy(z["desc"].tobytes())
    planes = [z[f"plane{c}"] for c in range(d.nbcomp)]
    qts = [z["qtables"][i] if z["qt_present"][i] else None for i in range(4)]
    out = orc.backend_run(d, planes, qts)
    hdr = f"P{6 if d.nbcomp == 3 else 5}\n{d.width} {d.height}\n255\n".encode()
    assert hdr + out.tobytes() == open(os.path.join(G, gold), "rb").read()
    if stem == "shaun_the_sheep":
        assert (d.nbmcuH, d.nbmcuV) == (19, 15)          # test/entete_test.c:278-280
        assert out.shape == (240, 900)                   # padded rows: 15 MCU rows x 16, not 225


def test_oracle_matches_reference_objects_when_present():
    """In the build container the real reference objects are available: compare directly."""
    ref = pyoracle.load_ref("strict")
    if ref is None:
        pytest.skip("oracle/_ref not built here")
    blk = np.concatenate([pyoracle.synth_blocks(20000, 3, 0, k) for k in (0, 1, 2)])
    a, b = orc.iquant_blocks(blk, 0, 63, QT[2]), ref.iquant_blocks(blk, 0, 63, QT[2])
    assert np.array_equal(a, b)
    assert np.array_equal(orc.idct_opt_blocks(orc.izz_blocks(a)), ref.idct_opt_blocks(ref.izz_blocks(b)))
    d = helpers.oracle_desc(301, 97, ((2, 2), (1, 1), (1, 1)))
    planes = helpers.synth_planes(d, 9)
    assert np.array_equal(orc.backend_run(d, planes, QT), ref.backend_run(d, planes, QT))


@pytest.mark.parametrize("name", ["c1_512x512_gray", "c5_1920x1080_420", "c2_4096x4096_444", "c4_4096x4096_422", "c3_8192x8192_420"])
def test_oracle_whole_rasters_at_baseline_sizes(name):
    """The restatement reproduces the compiled reference's whole raster (SHA-256 from oracle/make_golden.py) at every
    BASELINE.json frame size -- the hashes the GPU suite compares the HIP path with."""
    import json
    gold = json.load(open(os.path.join(helpers.GOLDEN, "meta.json")))["baseline_sizes"][name]
    d = helpers.oracle_desc(gold["width"], gold["height"], [tuple(s) for s in gold["sampling"]])
    planes = helpers.synth_planes(d, gold["seed"])
    out = pyoracle.oracle.backend_run(d, planes, helpers.qtables_ramp())
    band = d.max_v * 8
    assert out.shape[0] == gold["rows"]
    assert hashlib.sha256(np.ascontiguousarray(out[:band]).tobytes()).hexdigest() == gold["sha256_first_band"]
    assert hashlib.sha256(np.ascontiguousarray(out[-band:]).tobytes()).hexdigest() == gold["sha256_last_band"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == gold["sha256"]


def test_kat_frames_through_the_oracle():
    """The adversarial stage blocks (64 impulse positions x +-{1, 1024, 2047}, all-32767, all--32768, full-range int16) as
    whole gray / 4:4:4 / 4:2:0 frames, with the ramp table, the int16-wrapping table and without iquant: oracle raster ==
    the hash the compiled reference produced (oracle/make_golden.py --kat-only)."""
    import json
    meta = json.load(open(os.path.join(helpers.GOLDEN, "meta.json")))["kat_frames"]
    blocks = np.load(os.path.join(helpers.GOLDEN, "stage_vectors.npz"))["blocks"]
    assert len(meta) == 9
    for key, info in sorted(meta.items()):
        samp, w, h, planes = helpers.kat_frame(blocks, info["layout"])
        d = helpers.oracle_desc(w, h, samp, idq=info["idq"][:len(samp)], apply_iquant=bool(info["apply_iquant"]))
        out = pyoracle.oracle.backend_run(d, planes, helpers.qtables_ramp())
        assert out.shape[0] == info["rows"] and helpers.sha256_hex(out) == info["sha256"], key
