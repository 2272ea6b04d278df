"""CPU suite: the arithmetic shortcuts the HIP kernel takes are exact.

  * tools/explore_exact.c  : fma(x, hi, x*lo) == the reference's (float)((double)x * SQRT_8),
    (float)((double)x / SQRT_2) and x / (float)SQRT_2 for EVERY float mantissa and sign.
  * tools/explore_color.c  : the integer colour formulas == ycc2rgb_pixel on all 2^24 inputs.
  * host_mirror_check      : jpeg_math.hpp itself (the file the kernels include), compiled for the
    host, == the oracle on random/adversarial blocks and on all 2^24 colours.
"""
import os
import shutil
import subprocess

import pytest

import helpers

ROOT = helpers.ROOT


def build_and_run(src, tmp_path, libs, expect):
    exe = str(tmp_path / "a.out")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", src)] + libs)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    for e in expect:
        assert e in out.stdout, (e, out.stdout[-2000:])
    return out.stdout


def test_fp32_hi_lo_sequences_are_exact(tmp_path):
    out = build_and_run("explore_exact.c", tmp_path, ["-lquadmath", "-lm"], [])
    # the split the kernel uses is "bump 0"
    sect = out.split("== bump 0")[1].split("== bump 1")[0]
    for name in ("scale2", "shrink2", "divk2"):
        assert f"{name}" in sect and ": 0 mismatches of 2^24" in [l for l in sect.splitlines() if l.startswith(name)][0]
    # the constants in jpeg_math.hpp are that split (times exact powers of two)
    hdr = open(os.path.join(ROOT, "jpeg-decoder_amd", "csrc", "jpeg_math.hpp")).read()
    assert "Sh=0x1.6a09e6p+1 Sl=0x1.9fcef4p-25" in sect and "0x1.6a09e6p-2f" in hdr and "0x1.9fcef4p-28f" in hdr
    assert "Rh=0x1.6a09e6p-1 Rl=0x1.9fcef4p-27" in sect and "0x1.9fcef4p-26f" in hdr
    assert "Kh=0x1.6a09e6p-1 Kl=0x1.9fcef4p-26" in sect and "0x1.9fcef4p-25f" in hdr


def test_integer_colour_formulas_are_exact(tmp_path):
    out = build_and_run("explore_color.c", tmp_path, [], ["0 mismatches over 2^24 inputs"])
    assert "R: S=16 A=91881 B in [0, 214)" in out
    assert "B: S=16 A=116130 B in [0, 274)" in out
    assert "G: S=23 A1=2886855 A2=5990640 B in [4, 237)" in out


def test_device_math_header_on_host():
    exe = os.path.join(ROOT, "jpeg-decoder_amd", "host_mirror_check")
    if not os.path.exists(exe):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("host_mirror_check not built and no hipcc")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jpeg-decoder_amd", "csrc")])
    out = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "host_mirror_check: OK" in out.stdout, out.stdout[-2000:]
