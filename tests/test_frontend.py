"""CPU suite for the host front end, no GPU needed.

The entropy decoder fills full-frame planes; here the CPU oracle (test infrastructure) stands in for
the GPU back end so that the front end is pinned on its own:

  * planes == the planes the reference's own VLD decoded (tests/golden/*_planes.npz, made by
    oracle/make_golden.py through ref_dump_baseline_planes);
  * for every file of tests/golden/jpeg/ : same status as the reference binary (its exit code is the
    erreur_code_t) and, image by image (one per scan for progressive files, the partial image of a
    damaged baseline file), the same bytes -- expectations in expect.json come from
    oracle/_ref/jpeg2ppm_sans_opt via oracle/make_jpeg_fixtures.py;
  * the reference's header fixtures (test/test_file) give the codes test/entete_test.c:338-378 expects.
"""
import hashlib
import json
import os
import re

import numpy as np
import pytest

import helpers
import jpeg_decoder_amd as jg
from jpeg_decoder_amd import frontend as F
from oracle import pyoracle

JDIR = os.path.join(helpers.GOLDEN, "jpeg")
EXPECT = json.load(open(os.path.join(JDIR, "expect.json")))["files"]
ERR = {n: i for i, n in enumerate(
    "SUCCESS ERR_0XFF00 ERR_AC_BAD ERR_DC_BAD ERR_SOF_BAD ERR_HUFF_CODE_1 ERR_DIFF_AH_AL ERR_PARAM ERR_OPT ERR_NO_HT ERR_NO_QT "
    "ERR_INVALID_FILE_PATH ERR_INVALID_FILE_EXT ERR_NB_COMP ERR_SOF_NON_SUPPORTE ERR_APP0_LEN ERR_NO_APP0 ERR_COMP_ID ERR_COM_LEN "
    "ERR_DHT_START_0 ERR_DHT_LEN ERR_NO_DHT ERR_DQT_LEN ERR_DQT_PRECISION ERR_DQT_ID ERR_NO_DQT ERR_EOI_BEFORE_SOS ERR_NO_EOI "
    "ERR_HUFF_BAD ERR_HUFF_ID ERR_HUFF_MORE_256 ERR_JFIF_VERSION ERR_NO_JFIF ERR_MARKER_BAD ERR_MARKER_UNKNOWN ERR_SEVERAL_SOF "
    "ERR_SOF_PRECISION ERR_SOF_LEN ERR_NO_SOF ERR_SEVERAL_SOI ERR_NO_SOI ERR_SOS_COMP_ID ERR_SOS_SS ERR_SOS_SE ERR_SOS_AH ERR_SOS_AL "
    "ERR_SOS_NB_COMP ERR_SOS_LEN".split())}


def expected_images(e):
    """sha256 of the images the reference wrote, in scan order (x.pnm for baseline, x-<k>.pnm)."""
    outs = [o for o in e["outputs"] if "sha256" in o]
    outs.sort(key=lambda o: int(o["name"][2:-4]) if "-" in o["name"] else 0)
    return [o["sha256"] for o in outs]


def decode_with(data, backend):
    """The driver loop of src/baseline.c / src/progressive.c over our front end; <backend>(jpeg, desc,
    mcu_rows) renders the current planes.  -> (status, [image file bytes])."""
    imgs = []
    try:
        j = F.Jpeg(data)
    except jg.JpegGpuError as e:
        return e.code, imgs
    d = j.desc
    while not j.eoi:
        try:
            j.decode_scan()
        except jg.JpegGpuError as e:
            if j.scan.num_sof == 0:   # the reference has already written the complete MCU rows
                n = j.scan.mcu_rows_done
                body = backend(j, d, n)[:n * d.max_v * 8].tobytes() if n else b""
                imgs.append(F.pnm_header(d) + body)
            return e.code, imgs
        imgs.append(F.pnm_header(d) + backend(j, d, d.nbmcuV).tobytes())
        if j.eoi:
            break
        try:
            j.next_header()
        except jg.JpegGpuError as e:
            return e.code, imgs
    return 0, imgs


def oracle_backend(j, d, mcu_rows):
    od = pyoracle.FrameDesc.from_buffer_copy(bytes(d))
    return pyoracle.oracle.backend_run(od, j.planes(), j.qtables(), mcu_rows=(0, mcu_rows))


def test_header_declares_every_symbol_exported():
    text = open(os.path.join(helpers.ROOT, "include", "jpeggpu_decode.h")).read()
    names = sorted(set(re.findall(r"\b(jpeggpu_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(jg.lib, n), n


@pytest.mark.parametrize("name", ["shaun_the_sheep", "invader_normal"])
def test_planes_equal_reference_vld(name):
    g = np.load(os.path.join(helpers.GOLDEN, name + "_planes.npz"))
    j = F.Jpeg(open(os.path.join(helpers.GOLDEN, name + ".jpeg"), "rb").read())
    j.decode_scan()
    assert j.eoi and j.scan.scans_decoded == 1
    d = j.desc
    for c in range(d.nbcomp):
        assert np.array_equal(j.plane(c), g[f"plane{c}"].reshape(-1, 64)), c
    for i, q in enumerate(j.qtables()):
        assert (q is not None) == bool(g["qt_present"][i])
        if q is not None:
            assert np.array_equal(q, g["qtables"][i].reshape(64))


@pytest.mark.parametrize("name", ["shaun_the_sheep.jpeg:shaun_the_sheep.ppm", "invader_normal.jpeg:invader.pgm"])
def test_reference_golden_files_through_front_end(name):
    src, gold = name.split(":")
    rc, imgs = decode_with(open(os.path.join(helpers.GOLDEN, src), "rb").read(), oracle_backend)
    assert rc == 0 and len(imgs) == 1
    assert imgs[0] == open(os.path.join(helpers.GOLDEN, gold), "rb").read()   # valid_out/ of the reference


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_same_status_and_images_as_reference(name):
    e = EXPECT[name]
    rc, imgs = decode_with(open(os.path.join(JDIR, name), "rb").read(), oracle_backend)
    assert rc == e["rc"], (rc, e["err"], jg.last_error())
    assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(e)
    if "entete_test_expect" in e:
        assert rc == ERR[e["entete_test_expect"]]


def test_header_model_of_reference_fixtures():
    """test/entete_test.c:36-130,190-296: what decode_entete must leave in img_t."""
    j = F.Jpeg(open(os.path.join(JDIR, "ref_invader_indice_diff.jpeg"), "rb").read())
    d, s = j.desc, j.scan
    assert (d.width, d.height, d.nbcomp, d.max_h, d.max_v, d.nbmcuH, d.nbmcuV) == (8, 8, 1, 1, 1, 1, 1)
    assert d.comp[0].idq == 3 and (s.ss, s.se, s.ah, s.al, s.num_sof) == (0, 63, 0, 0, 0)
    q = j.qtables()
    assert q[3] is not None and (q[3] == 1).all() and q[0] is None and q[1] is None and q[2] is None
    j = F.Jpeg(open(os.path.join(JDIR, "ref_shaun_the_sheep.jpeg"), "rb").read())
    d = j.desc
    assert (d.width, d.height, d.nbcomp, d.max_h, d.max_v, d.nbmcuH, d.nbmcuV) == (300, 225, 3, 2, 2, 19, 15)
    assert [(d.comp[c].hs, d.comp[c].vs, d.comp[c].idq) for c in range(3)] == [(2, 2, 0), (1, 1, 1), (1, 1, 1)]
    assert np.array_equal(j.qtables()[0], helpers_shaun_qt()[0]) and np.array_equal(j.qtables()[1], helpers_shaun_qt()[1])


def helpers_shaun_qt():
    from jpeg_decoder_amd import synth
    return synth.shaun_qtables()


def test_comment_segment():
    j = F.Jpeg(open(os.path.join(JDIR, "base_444_comment_24x24.jpeg"), "rb").read())
    assert j.comments == [b"bonjour"]


def test_progressive_scan_script_and_dirty_components():
    j = F.Jpeg(open(os.path.join(JDIR, "c4_prog422_160x120.jpg"), "rb").read())
    d = j.desc
    assert j.scan.num_sof == 2 and (d.comp[0].hs, d.comp[0].vs, d.comp[1].hs) == (2, 1, 1)
    seen = []
    while not j.eoi:
        s = j.scan
        seen.append((list(s.scan_comp[:s.nb_scan_comp]), s.ss, s.se, s.ah, s.al))
        before = j.planes()
        j.decode_scan()
        after = j.planes()
        for c in range(3):   # a scan only touches the components it names (first one only for AC scans)
            if c not in (seen[-1][0] if s.se == 0 else seen[-1][0][:1]):
                assert np.array_equal(before[c], after[c])
        j.next_header()
    assert seen[0] == ([0, 1, 2], 0, 0, 0, 1) and len(seen) == 10


def test_out_file_name():
    # src/utils.c:56-90
    assert F.out_file_name("dir/a.b.jpeg", None, 3, 0) == "dir/a.b.ppm"
    assert F.out_file_name("dir/a.jpg", None, 1, 0) == "dir/a.pgm"
    assert F.out_file_name("dir/a.jpg", None, 3, 7) == "dir/a-7.ppm"
    assert F.out_file_name("a.jpg", "out/x.pnm", 3, 0) == "out/x.pnm"
    assert F.out_file_name("a.jpg", "out/x.pnm", 1, 12) == "out/x-12.pnm"


def test_write_pnm_keeps_padded_rows(tmp_path):
    d = helpers.oracle_desc(5, 3, [(1, 1)])
    gd = helpers.to_gpu_desc(jg, d)
    raster = np.arange(8 * 5, dtype=np.uint8).reshape(8, 5)
    p = tmp_path / "o.pgm"
    F.write_pnm(p, gd, raster)
    assert p.read_bytes() == b"P5\n5 3\n255\n" + raster.tobytes()   # 8 rows for a 3-row image, like the reference
    F.write_pnm(tmp_path / "c.pgm", gd, raster, rows=3)               # optional crop to the header's height
    assert (tmp_path / "c.pgm").read_bytes() == b"P5\n5 3\n255\n" + raster[:3].tobytes()
    with pytest.raises(jg.JpegGpuError):
        F.write_pnm(tmp_path / "c.pgm", gd, raster, rows=9)
    d3 = helpers.to_gpu_desc(jg, helpers.oracle_desc(2, 8, [(1, 1)] * 3))
    r3 = np.arange(8 * 6, dtype=np.uint8).reshape(8, 6)
    F.write_pnm(tmp_path / "o.ppm", d3, r3)
    assert (tmp_path / "o.ppm").read_bytes() == b"P6\n2 8\n255\n" + r3.tobytes()


def test_open_file_checks(tmp_path):
    # ouverture_fichier_in, src/utils.c:43-54
    p = tmp_path / "x.png"
    p.write_bytes(b"\xff\xd8\xff\xd9")
    with pytest.raises(jg.JpegGpuError) as e:
        F.Jpeg.from_file(p)
    assert e.value.code == ERR["ERR_INVALID_FILE_EXT"]
    with pytest.raises(jg.JpegGpuError) as e:
        F.Jpeg.from_file(tmp_path / "missing.jpg")
    assert e.value.code == ERR["ERR_INVALID_FILE_PATH"]
    q = tmp_path / "x.jpg"
    q.write_bytes(b"\xff\xd8\xff\xd9")
    with pytest.raises(jg.JpegGpuError) as e:
        F.Jpeg.from_file(q)
    assert e.value.code == ERR["ERR_EOI_BEFORE_SOS"]


def test_render_needs_a_gpu():
    """No CPU fallback: without a device the render step fails loudly."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    j = F.Jpeg(open(os.path.join(JDIR, "base_420_61x47_q80.jpg"), "rb").read())
    j.decode_scan()
    with pytest.raises(jg.JpegGpuError) as e:
        j.render()
    assert e.value.code in (jg.ERR_NO_DEVICE, jg.ERR_HIP)


def test_front_end_survives_damaged_files(tmp_path):
    """tools/fuzz_frontend.cpp under AddressSanitizer + UBSan: 24 000 randomly damaged variants of the
    fixtures, every scan decoded -- no invalid access, no hang, only status codes; and each variant decoded three ways
    (default readers, the same without BMI2, the checked reader everywhere) with equal status, message and coefficients."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "fuzz"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(helpers.ROOT, "jpeg-decoder_amd", "csrc"), "-I" + os.path.join(helpers.ROOT, "include"),
                           "-o", str(exe), os.path.join(helpers.ROOT, "tools", "fuzz_frontend.cpp"),
                           os.path.join(helpers.ROOT, "jpeg-decoder_amd", "csrc", "frontend.cpp")])
    r = subprocess.run([str(exe), JDIR, "24000"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "iterations 24000" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "MISMATCH" not in r.stdout
