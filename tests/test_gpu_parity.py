"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden
fixtures.  Bar: bit-exact (integer/byte outputs)."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers
from oracle import pyoracle

pytestmark = pytest.mark.gpu
orc = pyoracle.oracle
G = helpers.GOLDEN
QT = helpers.qtables_ramp()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---- stage-level contracts ---------------------------------------------------------------------

def test_izz_known_answer(gpu):
    from test_oracle_golden import ZZ_TABLE
    assert np.array_equal(gpu.izz_blocks(np.arange(64, dtype=np.int16)[None, :])[0].T, ZZ_TABLE)


def test_iquant_identity_and_band(gpu):
    q = np.arange(64, dtype=np.uint16)
    line = ((np.arange(64) + 1) * 2).astype(np.int16)
    assert np.array_equal(gpu.iquant_blocks(line[None, :], 0, 63, q)[0].astype(np.int32), q.astype(np.int32) * line)
    v = np.load(os.path.join(G, "stage_vectors.npz"))
    assert np.array_equal(gpu.iquant_blocks(v["blocks"], 5, 20, v["q_ramp"]), v["iquant_band_5_20"])
    assert np.array_equal(gpu.iquant_blocks(v["blocks"], 63, 63, v["q_wrap"]), orc.iquant_blocks(v["blocks"], 63, 63, v["q_wrap"]))


def test_stage_vectors_from_reference(gpu):
    v = np.load(os.path.join(G, "stage_vectors.npz"))
    for name in ("ramp", "wrap"):
        dq = gpu.iquant_blocks(v["blocks"], 0, 63, v[f"q_{name}"])
        assert np.array_equal(dq, v[f"iquant_{name}"])
        zz = gpu.izz_blocks(dq)
        assert np.array_equal(zz, v[f"izz_{name}"])
        assert np.array_equal(gpu.idct_opt_blocks(zz), v[f"idct_{name}"])


@pytest.mark.parametrize("kind,n", [(0, 300000), (1, 200000), (2, 200000)])
def test_idct_random_blocks_vs_oracle(gpu, kind, n):
    blk = pyoracle.synth_blocks(n, 0xC0FFEE + kind, 0, kind)
    zz = orc.izz_blocks(orc.iquant_blocks(blk, 0, 63, QT[2 if kind == 2 else 0]))
    assert np.array_equal(gpu.idct_opt_blocks(zz), orc.idct_opt_blocks(zz))


def test_idct_property_within_one_of_slow_idct(gpu):
    from test_oracle_golden import slow_idct
    rng = np.random.default_rng(2)
    blocks = rng.integers(0, 256, size=(10000, 8, 8)).astype(np.int16)   # test/idct_opt_test.c:26
    assert np.max(np.abs(gpu.idct_opt_blocks(blocks).astype(np.float64) - np.floor(slow_idct(blocks)))) <= 1


def test_ycc2rgb_every_input(gpu):
    meta = json.load(open(os.path.join(G, "meta.json")))
    idx = np.arange(1 << 24, dtype=np.uint32)
    rgb = gpu.ycc2rgb_pixels(idx & 255, (idx >> 8) & 255, (idx >> 16) & 255)
    assert sha(rgb) == meta["ycc2rgb_all_sha256"]


# ---- frame-level back end -------------------------------------------------------------------------

def frame_cases():
    meta = json.load(open(os.path.join(G, "meta.json")))
    return sorted(meta["frames"].items())


@pytest.mark.parametrize("key,info", frame_cases())
def test_frame_vectors_from_reference(gpu, key, info):
    d = helpers.oracle_desc(info["width"], info["height"], [tuple(s) for s in info["sampling"]], idq=info["idq"],
                            apply_iquant=bool(info["apply_iquant"]))
    planes = helpers.synth_planes(d, info["seed"], info["kind"])
    out = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
    assert out.shape[0] == info["rows"]
    assert sha(out) == info["sha256"]


@pytest.mark.parametrize("stem,gold", [("invader_normal", "invader.pgm"), ("shaun_the_sheep", "shaun_the_sheep.ppm")])
def test_reference_golden_files(gpu, stem, gold):
    z = np.load(os.path.join(G, stem + "_planes.npz"))
    d = gpu.FrameDesc.from_buffer_copy(z["desc"].tobytes())
    planes = [z[f"plane{c}"] for c in range(d.nbcomp)]
    qts = [z["qtables"][i] if z["qt_present"][i] else None for i in range(4)]
    out = gpu.backend_run(d, planes, qts)
    hdr = f"P{6 if d.nbcomp == 3 else 5}\n{d.width} {d.height}\n255\n".encode()
    assert hdr + out.tobytes() == open(os.path.join(G, gold), "rb").read()


WIDTHS = [(1, 1), (7, 9), (8, 8), (9, 17), (63, 16), (64, 64), (65, 33), (255, 31), (257, 8), (1023, 24), (1025, 16), (2049, 9), (4099, 17)]
SAMPLINGS = [((1, 1),), ((1, 1),) * 3, ((2, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)),
             ((4, 1), (1, 1), (1, 1)), ((2, 2), (2, 1), (1, 2)), ((4, 2), (2, 2), (1, 1))]


@pytest.mark.parametrize("samp", SAMPLINGS, ids=lambda s: "_".join(f"{a}{b}" for a, b in s))
def test_ragged_sizes_vs_oracle(gpu, samp):
    """Edge tiles, width crop, padded rows, every alignment of the raster row start."""
    for i, (w, h) in enumerate(WIDTHS):
        for aq in (True, False):
            d = helpers.oracle_desc(w, h, samp, apply_iquant=aq)
            planes = helpers.synth_planes(d, 500 + i, kind=0 if aq else 1)
            want = orc.backend_run(d, planes, QT)
            got = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
            assert np.array_equal(want, got), (samp, w, h, aq)


@pytest.mark.parametrize("samp", [((2, 2),), ((1, 2),), ((4, 1),), ((1, 4),), ((4, 4),)], ids=lambda s: f"{s[0][0]}x{s[0][1]}")
def test_gray_with_sampling_above_one(gpu, samp):
    """One component with sampling factors > 1: save_mcu_ligne_bw (src/decoder_utils.c:55-58) only reads the first
    block row of each MCU row, rows 8.. of a band repeat rows 0..7.  Oracle pinned on the reference objects
    (oracle/check_oracle_vs_ref.c); fast (power-of-two tile) and generic gray paths."""
    for (w, h) in ((61, 37), (300, 70), (2048, 64), (4099, 33)):
        d = helpers.oracle_desc(w, h, samp)
        planes = helpers.synth_planes(d, 600 + w)
        want = orc.backend_run(d, planes, QT)
        assert np.array_equal(gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT), want), (samp, w, h)
        if samp[0][1] > 1:
            band = want[:8 * samp[0][1]]
            assert np.array_equal(band[8:16], band[0:8])


def test_out_stride_and_untouched_padding(gpu):
    d = helpers.oracle_desc(301, 50, ((2, 2), (1, 1), (1, 1)))
    planes = helpers.synth_planes(d, 42)
    want = orc.backend_run(d, planes, QT)
    stride = 301 * 3 + 13
    buf = np.full((want.shape[0], stride), 0xAB, dtype=np.uint8)
    got = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT, out=buf, out_stride=stride)
    assert np.array_equal(got[:, :903], want)
    assert np.all(got[:, 903:] == 0xAB)


def test_out_stride_limit(gpu):
    d = helpers.to_gpu_desc(gpu, helpers.oracle_desc(64, 64, ((1, 1),) * 3))
    with pytest.raises(Exception, match="out_stride"):
        gpu.backend_run_device(d, [4096, 4096, 4096], QT, 4096, out_stride=(1 << 26) + 4)   # rejected before any access


@pytest.mark.parametrize("env", [{"JPEGGPU_NO_UNIFORM": "1"}, {"JPEGGPU_TILE_MCUS": "21"}, {"JPEGGPU_TILE_MCUS": "5"},
                                 {"JPEGGPU_TARGET_BLOCKS": "128"}], ids=lambda e: "_".join(f"{k[8:]}{v}" for k, v in e.items()))
def test_other_tilings_and_prologues(gpu, env):
    """The per-lane prologue (components sharing a wave, tables through LDS) and tiles that are not a power of
    two wide (generic pixel phase) stay bit-exact: the tuning knobs select them."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    gpu.reload_tuning()          # the knobs are read from the environment once, not per launch
    try:
        for samp in (((1, 1),) * 3, ((2, 2), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 1),)):
            d = helpers.oracle_desc(1500 + 8 * len(samp), 70, samp)
            planes = helpers.synth_planes(d, 99, kind=1)
            assert np.array_equal(orc.backend_run(d, planes, QT), gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)), samp
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        gpu.reload_tuning()


@pytest.mark.parametrize("samp", [((1, 1),) * 3, ((2, 2), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1))],
                         ids=["444", "420", "422", "440"])
def test_linear_tiling(gpu, samp):
    """Frames whose MCU rows are not a whole number of 64-MCU tiles are tiled in raster order (a tile runs on into the
    next MCU row; the last tile of the frame is partly empty).  Widths that crop the last MCU, padded rows, several
    frames per launch; the knob JPEGGPU_NO_LINEAR gives the per-row tiling, which must produce the same bytes."""
    import torch
    mw = samp[0][0] * 8
    for (w, h) in ((65 * mw, 40), (65 * mw - 5, 8 * samp[0][1] * 3), (120 * mw, 8 * samp[0][1] * 5), (200 * mw - mw + 1, 70), (129 * mw - 3, 8 * samp[0][1])):
        d = helpers.oracle_desc(w, h, samp)
        assert d.nbmcuH > 64 and d.nbmcuH % 64
        planes = helpers.synth_planes(d, 800 + w % 97)
        want = orc.backend_run(d, planes, QT)
        gd = helpers.to_gpu_desc(gpu, d)
        assert np.array_equal(gpu.backend_run(gd, planes, QT), want), (samp, w, h)
        stride = want.shape[1] + 29
        buf = np.full((want.shape[0], stride), 0xEE, dtype=np.uint8)
        got = gpu.backend_run(gd, planes, QT, out=buf, out_stride=stride)
        assert np.array_equal(got[:, :want.shape[1]], want) and np.all(buf[:, want.shape[1]:] == 0xEE)
    # three frames in one launch, and the same with the per-row tiling
    nb = [helpers.plane_shape(d, c)[0] for c in range(3)]
    host = [pyoracle.synth_blocks(nb[c] * 3, 850 + c) for c in range(3)]
    dev = [torch.from_numpy(a).cuda() for a in host]
    rows, rb = gpu.raster_rows(gd), gpu.raster_row_bytes(gd)
    outs = []
    for knob in (None, "1"):
        if knob:
            os.environ["JPEGGPU_NO_LINEAR"] = knob
        gpu.reload_tuning()
        try:
            out = torch.zeros(3 * rows * rb, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            gpu.backend_run_batch_device(gd, 3, [t.data_ptr() for t in dev], None, QT, out.data_ptr())
            gpu.synchronize()
            outs.append(out.cpu().numpy().reshape(3, rows, rb))
        finally:
            os.environ.pop("JPEGGPU_NO_LINEAR", None)
            gpu.reload_tuning()
    assert np.array_equal(outs[0], outs[1])
    for f in range(3):
        assert np.array_equal(outs[0][f], orc.backend_run(d, [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)], QT)), f


@pytest.mark.parametrize("knob", [None, "JPEGGPU_NO_RESTAGE", "JPEGGPU_NO_UNIFORM"], ids=["default", "no_restage", "per_lane_prologue"])
def test_kat_frames_through_the_fused_kernels(gpu, knob):
    """SURVEY 8d's adversarial set -- impulses at each of the 64 positions x +-{1, 1024, 2047}, all-32767 / all--32768 blocks,
    full-range int16 under the wrapping table (test/iqzz_test.c:21-29,48-64 are the reference's own KATs for the two stages) --
    through jpeggpu_backend_run, i.e. through izz_rp + idct_block_rp with their packed op_sel/neg instructions (the stage
    entry jpeggpu_idct_opt_blocks runs the scalar idct_block instead): gray, 4:4:4 (backend_kernel_rs444) and 4:2:0
    (backend_kernel_rs420) frames, each also through the general kernel and its per-lane prologue.  Against the oracle byte
    for byte and against the hash of the compiled reference (tests/golden/meta.json, oracle/make_golden.py --kat-only)."""
    meta = json.load(open(os.path.join(G, "meta.json")))["kat_frames"]
    blocks = np.load(os.path.join(G, "stage_vectors.npz"))["blocks"]
    if knob:
        os.environ[knob] = "1"
    gpu.reload_tuning()
    try:
        for key, info in sorted(meta.items()):
            samp, w, h, planes = helpers.kat_frame(blocks, info["layout"])
            d = helpers.oracle_desc(w, h, samp, idq=info["idq"][:len(samp)], apply_iquant=bool(info["apply_iquant"]))
            got = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
            assert got.shape[0] == info["rows"] and sha(got) == info["sha256"], key
            assert np.array_equal(got, orc.backend_run(d, planes, QT)), key
    finally:
        if knob:
            os.environ.pop(knob, None)
        gpu.reload_tuning()


RS_CASES = {
    # (width, height, frames, row padding, apply_iquant)
    "444": (((1, 1),) * 3, [(512, 8, 1, 0, True), (1024, 24, 3, 16, False), (1536, 13, 2, 48, True), (2048, 40, 1, 7, True), (4096, 16, 2, 0, False),
            # linear tiling (an even number of MCUs per row above 64: backend_kernel_rs444_lin); 520 = 65 MCUs stays on the general kernel
            (528, 40, 2, 0, True), (1040, 24, 3, 16, False), (1920, 88, 1, 48, True), (1936, 17, 2, 0, True), (520, 16, 1, 0, True)]),
    "420": (((2, 2), (1, 1), (1, 1)), [(512, 16, 1, 0, True), (1024, 40, 3, 16, False), (1920, 1080, 2, 0, True), (1040, 50, 2, 32, True),
                                     (528, 33, 3, 0, False), (2048, 64, 1, 7, True), (8192, 32, 1, 0, True), (496, 16, 1, 0, True)]),
}


@pytest.mark.parametrize("layout", sorted(RS_CASES))
def test_restaged_kernels_with_16_byte_stores(gpu, layout):
    """4:4:4 frames made of whole 64-MCU tiles take backend_kernel_rs444 (other uncropped widths with an even number of MCUs
    per row: backend_kernel_rs444_lin, linear tiling), uncropped 4:2:0 frames backend_kernel_rs420 (32-MCU
    tiles, Cb and Cr sharing a wave, linear tiling when an MCU row is not a whole number of tiles) -- both with the RGB restaged
    inside the wave and 16-byte stores, and only when the raster is 16-byte aligned; anything else, and JPEGGPU_NO_RESTAGE, the
    general kernel.  Both against the oracle: several frames per launch, padded rows, the special chroma pair, both iquant modes."""
    import torch
    samp, cases = RS_CASES[layout]
    for (w, h, nf, pad, aq) in cases:
        d = helpers.oracle_desc(w, h, samp, apply_iquant=aq)
        gd = helpers.to_gpu_desc(gpu, d)
        nb = [helpers.plane_shape(d, c)[0] for c in range(3)]
        host = [pyoracle.synth_blocks(nb[c] * nf, 4400 + 3 * w + c, 0, 0 if aq else 1) for c in range(3)]
        if w in (1024, 1040):      # the special pair (cb, cr) = (28, 228) under a ramp of luma values
            for c in range(3):
                host[c][:nb[c], :] = 0
            host[0][:nb[0], 0] = ((np.arange(nb[0]) % 256) - 128) * 8
            host[1][:nb[1], 0] = (28 - 128) * 8
            host[2][:nb[2], 0] = (228 - 128) * 8
        dev = [torch.from_numpy(a).cuda() for a in host]
        rows, rb = gpu.raster_rows(gd), gpu.raster_row_bytes(gd)
        stride = rb + pad
        outs = []
        for knob in (None, "1"):
            if knob:
                os.environ["JPEGGPU_NO_RESTAGE"] = knob
            gpu.reload_tuning()
            try:
                out = torch.full((nf * rows * stride + 32,), 0x5A, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                gpu.backend_run_batch_device(gd, nf, [t.data_ptr() for t in dev], None, QT, out.data_ptr(), stride, 0)
                gpu.synchronize()
                outs.append(out.cpu().numpy())
            finally:
                os.environ.pop("JPEGGPU_NO_RESTAGE", None)
                gpu.reload_tuning()
        assert np.array_equal(outs[0], outs[1]), (w, h)
        got = outs[0][:nf * rows * stride].reshape(nf, rows, stride)
        assert np.all(got[:, :, rb:] == 0x5A) and np.all(outs[0][nf * rows * stride:] == 0x5A)
        for f in range(nf):
            want = orc.backend_run(d, [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)], QT)
            assert np.array_equal(got[f, :, :rb], want), (w, h, f)


def test_full_int16_range_and_wrap(gpu):
    d = helpers.oracle_desc(640, 64, ((2, 2), (1, 1), (1, 1)), idq=[2, 2, 2])
    planes = helpers.synth_planes(d, 7, kind=2)
    assert np.array_equal(orc.backend_run(d, planes, QT), gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT))


def test_zero_and_dc_only_frames(gpu):
    d = helpers.oracle_desc(128, 32, ((2, 1), (1, 1), (1, 1)))
    planes = [np.zeros(helpers.plane_shape(d, c), np.int16) for c in range(3)]
    out = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
    assert np.array_equal(out, orc.backend_run(d, planes, QT)) and np.all(out == 128)
    for c in range(3):
        planes[c][:, 0] = np.arange(planes[c].shape[0]) % 257 - 128
    assert np.array_equal(gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT), orc.backend_run(d, planes, QT))


def test_special_chroma_pair_reaches_the_fused_kernel(gpu):
    """(cb, cr) = (28, 228): the one chroma pair whose green offset depends on y."""
    d = helpers.oracle_desc(256, 8, ((1, 1),) * 3, apply_iquant=False)
    planes = [np.zeros(helpers.plane_shape(d, c), np.int16) for c in range(3)]
    planes[0][:, 0] = (np.arange(32) * 8 - 128) * 8     # DC -> flat blocks of luma 0, 8, 16, ...
    planes[1][:, 0] = (28 - 128) * 8
    planes[2][:, 0] = (228 - 128) * 8
    want = orc.backend_run(d, planes, QT)
    assert np.array_equal(gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT), want)


@pytest.mark.parametrize("samp", [((1, 1),) * 3, ((2, 2), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1))],
                         ids=["444", "420", "422", "440"])
def test_every_chroma_pair_through_the_fused_kernel(gpu, samp):
    """All 65 536 (cb, cr) pairs, one per MCU as flat chroma blocks, under textured luma: the packed colour
    path of each sub-sampling-specific pixel phase against the oracle (the special pair included)."""
    hs, vs = samp[0]
    d = helpers.oracle_desc(256 * 8 * hs, 256 * 8 * vs, samp, apply_iquant=False)
    assert d.nbmcuH == 256 and d.nbmcuV == 256
    planes = [pyoracle.synth_blocks(helpers.plane_shape(d, 0)[0], 1234, 0, 0)]
    planes[0][:, 0] = np.clip(planes[0][:, 0] * 12, -1100, 1100)          # luma spread over the whole range
    mcu = np.arange(65536)
    for c, val in ((1, mcu % 256), (2, mcu // 256)):                      # MCU (row j, column i): cb = i, cr = j
        pl = np.zeros(helpers.plane_shape(d, c), np.int16)
        pl[:, 0] = (val - 128) * 8 + 4                                    # flat block of exactly `val`
        planes.append(pl)
    want = orc.backend_run(d, planes, QT)
    got = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
    assert np.array_equal(got, want)
    # the chroma blocks really are flat and cover 0..255: check through a grey-luma rendering of one MCU row
    flat = [np.zeros_like(planes[0])] + planes[1:]
    row = orc.backend_run(d, flat, QT, mcu_rows=(128, 129))[128 * 8 * vs]
    assert len(np.unique(row.reshape(-1, 3)[:, 2])) >= 140                # blue follows cb across the row (clamped at both ends)


# ---- BASELINE.json sizes: stripes vs the oracle + size-independent properties -------------------

BIG = {"c2_4096x4096_444": (4096, 4096, ((1, 1),) * 3), "c3_8192x8192_420": (8192, 8192, ((2, 2), (1, 1), (1, 1))),
       "c4_4096x4096_422": (4096, 4096, ((2, 1), (1, 1), (1, 1))), "c1_512x512_gray": (512, 512, ((1, 1),)),
       "c5_1920x1080_420": (1920, 1080, ((2, 2), (1, 1), (1, 1)))}


@pytest.mark.parametrize("name", sorted(BIG))
def test_baseline_sizes(gpu, name):
    w, h, samp = BIG[name]
    d = helpers.oracle_desc(w, h, samp)
    planes = helpers.synth_planes(d, 0x5EED0000 + len(name))
    gd = helpers.to_gpu_desc(gpu, d)
    full = gpu.backend_run(gd, planes, QT)
    # (1) the WHOLE raster: against the hash the compiled reference produced for this frame (oracle/make_golden.py)
    #     and byte for byte against the oracle
    gold = json.load(open(os.path.join(G, "meta.json")))["baseline_sizes"][name]
    assert full.shape[0] == gold["rows"] and sha(full) == gold["sha256"], name
    assert np.array_equal(full, orc.backend_run(d, planes, QT)), name
    # (2) stripe decomposition == whole frame (the multi-GPU sharding property): run the two halves
    #     as independent frames and concatenate
    half = d.nbmcuV // 2
    parts = []
    for (m0, m1) in ((0, half), (half, d.nbmcuV)):
        sd = helpers.to_gpu_desc(gpu, d)
        sd.nbmcuV = m1 - m0
        sub = [planes[c][m0 * d.comp[c].vs * d.nbmcuH * d.comp[c].hs: m1 * d.comp[c].vs * d.nbmcuH * d.comp[c].hs] for c in range(d.nbcomp)]
        parts.append(gpu.backend_run(sd, sub, QT))
    assert np.array_equal(np.concatenate(parts), full)
    # (3) idempotence / determinism: a second run gives the same bytes
    assert sha(gpu.backend_run(gd, planes, QT)) == sha(full)


def test_batch_equals_frame_by_frame(gpu):
    """jpeggpu_backend_run_batch_device over N frames == N single-frame calls (C5's unit of work)."""
    import torch
    d = helpers.oracle_desc(1920, 1080, ((2, 2), (1, 1), (1, 1)))
    gd = helpers.to_gpu_desc(gpu, d)
    nf = 5
    nb = [helpers.plane_shape(d, c)[0] for c in range(3)]
    host = [pyoracle.synth_blocks(nb[c] * nf, 900 + c) for c in range(3)]
    dev = [torch.from_numpy(a).cuda() for a in host]
    rows, rb = gpu.raster_rows(gd), gpu.raster_row_bytes(gd)
    out = torch.zeros(nf * rows * rb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    gpu.backend_run_batch_device(gd, nf, [t.data_ptr() for t in dev], None, QT, out.data_ptr(), 0, 0, st.cuda_stream)
    st.synchronize()
    got = out.cpu().numpy().reshape(nf, rows, rb)
    assert rows == 1088                                   # padded rows quirk, src/decoder_utils.c:68
    for f in (0, 2, 4):
        planes = [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)]
        assert np.array_equal(got[f], gpu.backend_run(gd, planes, QT))
    f = 3
    planes = [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)]
    want = orc.backend_run(d, planes, QT, mcu_rows=(10, 12))
    assert np.array_equal(got[f][160:192], want[160:192])


def test_c5_batch_of_8192_frames_beyond_4GiB_offsets(gpu):
    """BASELINE.json configs[4] at its stated size: 8192 frames of 1920x1080 4:2:0 in ONE launch (51 GB of planes,
    51 GB of rasters, frame offsets far beyond 4 GiB).  Eight distinct frames are replicated on the device; every
    frame f must equal frame f mod 8, and those eight equal the oracle."""
    import torch
    free, _total = torch.cuda.mem_get_info()
    d = helpers.oracle_desc(1920, 1080, ((2, 2), (1, 1), (1, 1)))
    gd = helpers.to_gpu_desc(gpu, d)
    K, NF = 8, 8192
    nb = [helpers.plane_shape(d, c)[0] for c in range(3)]
    rows, rb = gpu.raster_rows(gd), gpu.raster_row_bytes(gd)
    fb = rows * rb
    need = NF * (sum(nb) * 128 + fb)
    if free < need + (8 << 30):
        pytest.skip(f"needs {need >> 30} GiB of free device memory, {free >> 30} GiB available")
    host = [pyoracle.synth_blocks(nb[c] * K, 0xC5 + c) for c in range(3)]
    dev = [torch.from_numpy(a).cuda().repeat(NF // K, 1) for a in host]     # frame f = frame f mod K
    assert all(t.shape[0] == NF * nb[c] for c, t in enumerate(dev))
    out = torch.empty(NF * fb, dtype=torch.uint8, device="cuda")
    assert (NF - 1) * fb > (1 << 32) and (NF - 1) * nb[1] * 128 > (1 << 32)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    gpu.backend_run_batch_device(gd, NF, [t.data_ptr() for t in dev], None, QT, out.data_ptr(), 0, 0, st.cuda_stream)
    st.synchronize()
    first = out[:K * fb]
    G = 64                                                  # compare 64 groups of K frames at a time
    v = out.view(NF // K, K * fb)
    for g0 in range(0, NF // K, G):
        assert bool((v[g0:g0 + G] == first).all()), f"a frame in groups {g0}..{g0 + G - 1} differs from its frame mod {K}"
    got = first.cpu().numpy().reshape(K, rows, rb)
    for f in range(K):
        planes = [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)]
        assert np.array_equal(got[f], orc.backend_run(d, planes, QT)), f
    del dev, out, v, first
    torch.cuda.empty_cache()


def test_bench_line_contract(gpu):
    """bench.py prints ONE JSON line with the driver's keys plus roofline and cpu_baseline (a short run)."""
    import json
    import subprocess
    import sys
    root = helpers.ROOT
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--prewarm-ms", "50", "--workload", "c5", "--frames", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in res, k
    assert res["unit"] == "MPix/s" and res["n_gpus"] == 1 and res["steps"] == 3 and res["scaling"] == "weak" and res["vs_baseline"] is None
    assert "workload" in res["config"] and res["value"] > 0
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_randomised_geometries_vs_oracle(gpu):
    """700 random frames -- every sampling layout the writers accept, widths that cut MCUs, tiles that wrap (linear tiling),
    batches, padded rows and frame strides, both iquant modes -- through the device-pointer entry, byte for byte vs the oracle."""
    import torch
    rng = np.random.default_rng(20251111)
    layouts = SAMPLINGS + [((2, 1), (2, 1), (1, 1)), ((1, 1), (1, 1), (1, 1)), ((2, 2), (1, 2), (1, 2)), ((4, 4), (2, 2), (2, 2)), ((3, 1), (1, 1), (1, 1))]
    for case in range(700):
        samp = layouts[int(rng.integers(len(layouts)))]
        mw, mh = max(s[0] for s in samp) * 8, max(s[1] for s in samp) * 8
        nmh = int(rng.choice([1, 2, 7, 31, 63, 64, 65, 66, 100, 127, 129, 200, 257]))
        w = nmh * mw - int(rng.integers(0, mw))
        h = int(rng.integers(1, 4)) * mh - int(rng.integers(0, mh))
        if w < 1 or h < 1 or w > 65535:
            continue
        aq = bool(rng.integers(2))
        d = helpers.oracle_desc(w, h, samp, apply_iquant=aq)
        gd = helpers.to_gpu_desc(gpu, d)
        nf = int(rng.choice([1, 1, 2, 3]))
        nc = len(samp)
        nb = [helpers.plane_shape(d, c)[0] for c in range(nc)]
        pad_blocks = int(rng.integers(0, 3))                       # frames need not be packed back to back
        host = [pyoracle.synth_blocks((nb[c] + pad_blocks) * nf, 7000 + 3 * case + c, 0, int(rng.integers(0, 2))) for c in range(nc)]
        dev = [torch.from_numpy(a).cuda() for a in host]
        rows, rb = gpu.raster_rows(gd), gpu.raster_row_bytes(gd)
        stride = rb + int(rng.choice([0, 0, 1, 5, 64]))
        fstride = stride * rows + int(rng.choice([0, 0, 3, 128]))
        out = torch.full((nf * fstride + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu.backend_run_batch_device(gd, nf, [t.data_ptr() for t in dev], [(nb[c] + pad_blocks) * 128 for c in range(nc)], QT,
                                     out.data_ptr(), stride, fstride)
        gpu.synchronize()
        got = out.cpu().numpy()
        for f in range(nf):
            planes = [host[c][f * (nb[c] + pad_blocks):f * (nb[c] + pad_blocks) + nb[c]] for c in range(nc)]
            want = orc.backend_run(d, planes, QT)
            fr = got[f * fstride:f * fstride + rows * stride].reshape(rows, stride)
            assert np.array_equal(fr[:, :rb], want), (case, samp, w, h, aq, nf, f, stride)
            assert np.all(fr[:, rb:] == 0xA5), (case, "row padding touched")
        assert np.all(got[nf * fstride:] == 0xA5), (case, "bytes past the last frame touched")
