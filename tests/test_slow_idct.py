"""Row f4: the slow transform the reference uses under "-f" (src/idct.c) as a second GPU
variant.  CPU part: the oracle restatement and the library's coefficient table against vectors taken
from the reference objects (tests/golden/idct_slow_vectors.npz) and against the reference binary run
with -f (expect.json: outputs_slow).  GPU part: the product path in that mode."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import helpers
import jpeg_decoder_amd as jg
from oracle import pyoracle
from test_frontend import EXPECT, JDIR, decode_with, expected_images, oracle_backend

orc = pyoracle.oracle
VEC = np.load(os.path.join(helpers.GOLDEN, "idct_slow_vectors.npz"))
SLOW = sorted(n for n, e in EXPECT.items() if "outputs_slow" in e)


@pytest.fixture
def oracle_slow():
    orc.set_idct_fast(False)
    yield
    orc.set_idct_fast(True)


def test_coefficient_table_bits():
    """calc_coef (src/idct.c:35-48): reference objects == oracle == the table the library uploads."""
    want = VEC["coef"].view(np.uint32)
    assert np.array_equal(orc.idct_coef().view(np.uint32), want)
    assert np.array_equal(jg.idct_coef_table().view(np.uint32), want)


def test_oracle_slow_idct_equals_reference_vectors():
    assert np.array_equal(orc.idct_blocks(VEC["blocks"]), VEC["out"])


def test_oracle_slow_idct_against_compiled_reference():
    ref = pyoracle.load_ref("strict")
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    b = rng.integers(-2048, 2048, size=(4000, 8, 8)).astype(np.int16)
    assert np.array_equal(orc.idct_blocks(b), ref.idct_blocks(b))


@pytest.mark.parametrize("name", SLOW)
def test_front_end_with_slow_oracle_equals_reference_dash_f(oracle_slow, name):
    e = dict(EXPECT[name], outputs=EXPECT[name]["outputs_slow"])
    rc, imgs = decode_with(open(os.path.join(JDIR, name), "rb").read(), oracle_backend)
    assert rc == 0
    assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(e)


def test_mode_switch_is_host_state():
    assert jg.get_idct_fast()
    jg.set_idct_fast(False)
    assert not jg.get_idct_fast()
    jg.set_idct_fast(True)
    assert jg.get_idct_fast()


# ---- GPU -------------------------------------------------------------------------------------------

@pytest.fixture
def gpu_slow(gpu, oracle_slow):
    gpu.set_idct_fast(False)
    yield gpu
    gpu.set_idct_fast(True)


@pytest.mark.gpu
def test_gpu_slow_idct_blocks(gpu):
    assert np.array_equal(gpu.idct_blocks(VEC["blocks"]), VEC["out"])
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.integers(-32768, 32768, size=(3000, 8, 8)), rng.integers(-300, 300, size=(5000, 8, 8))]).astype(np.int16)
    slow = gpu.idct_blocks(b)
    assert np.array_equal(slow, orc.idct_blocks(b))
    # the two transforms agree within 1 (test/idct_opt_test.c:35) but are not the same function: a few samples per million differ
    fast = gpu.idct_opt_blocks(b)
    diff = np.abs(slow.astype(int) - fast.astype(int))
    assert diff.max() == 1 and 0 < np.count_nonzero(diff) < 100


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,samp", [(300, 225, [(2, 2), (1, 1), (1, 1)]), (129, 65, [(1, 1)]), (200, 100, [(1, 1)] * 3),
                                      (131, 77, [(2, 1), (1, 1), (1, 1)]), (67, 45, [(4, 2), (2, 1), (1, 2)])])
@pytest.mark.parametrize("kind", [0, 2])
def test_gpu_slow_frames_equal_oracle(gpu_slow, w, h, samp, kind):
    d = helpers.oracle_desc(w, h, samp)
    planes = helpers.synth_planes(d, 900 + w, kind)
    qt = helpers.qtables_ramp()
    got = gpu_slow.backend_run(helpers.to_gpu_desc(gpu_slow, d), planes, qt)
    assert np.array_equal(got, orc.backend_run(d, planes, qt))
    gpu_slow.set_idct_fast(True)       # and the switch really selects the other transform
    orc.set_idct_fast(True)
    fast = gpu_slow.backend_run(helpers.to_gpu_desc(gpu_slow, d), planes, qt)
    assert np.array_equal(fast, orc.backend_run(d, planes, qt))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SLOW)
def test_gpu_files_equal_reference_dash_f(gpu_slow, name):
    e = dict(EXPECT[name], outputs=EXPECT[name]["outputs_slow"])
    rc, imgs = decode_with(open(os.path.join(JDIR, name), "rb").read(), lambda j, d, n: j.render())
    assert rc == 0
    assert [hashlib.sha256(b).hexdigest() for b in imgs] == expected_images(e)


@pytest.mark.gpu
def test_command_line_dash_f(gpu, tmp_path):
    exe = os.path.join(helpers.ROOT, "jpeg-decoder_amd", "jpeg2ppm_gpu")
    name = "c4_prog422_160x120.jpg"
    out = tmp_path / "o"
    r = subprocess.run([exe, "-f", "-o", str(out / "x.pnm"), os.path.join(JDIR, name)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-300:]
    got = {f: hashlib.sha256(open(out / f, "rb").read()).hexdigest() for f in os.listdir(out) if not os.path.islink(out / f)}
    assert got == {o["name"]: o["sha256"] for o in EXPECT[name]["outputs_slow"] if "sha256" in o}
