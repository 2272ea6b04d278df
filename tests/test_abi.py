"""CPU suite: the C-ABI library loads, exports every symbol include/jpeggpu.h declares, and its
host-only logic (geometry, validation, error convention) behaves like the reference's.  No compute
call succeeds without a GPU: the product path fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import jpeg_decoder_amd as jg

HDR = os.path.join(helpers.ROOT, "include", "jpeggpu.h")


def has_gpu():
    # counting devices does not initialise the GPU
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def test_every_declared_symbol_is_exported():
    text = open(HDR).read()
    names = sorted(set(re.findall(r"\b(jpeggpu_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(jg.lib, n), n


def test_struct_layout_matches_header():
    assert C.sizeof(jg.FrameDesc) == 28
    assert C.sizeof(jg.Component) == 4
    assert jg.FrameDesc.nbmcuH.offset == 8 and jg.FrameDesc.comp.offset == 16


@pytest.mark.parametrize("w,h,samp,mcu", [
    (300, 225, ((2, 2), (1, 1), (1, 1)), (19, 15)),      # shaun: test/entete_test.c:278-280
    (8, 8, ((1, 1),), (1, 1)),
    (1920, 1080, ((2, 2), (1, 1), (1, 1)), (120, 68)),
    (4096, 4096, ((2, 1), (1, 1), (1, 1)), (256, 512)),
    (17, 9, ((4, 2), (2, 1), (1, 2)), (1, 1)),
    (65535, 1, ((1, 1), (1, 1), (1, 1)), (8192, 1)),
])
def test_geometry_like_calcul_image_information(w, h, samp, mcu):
    d = jg.make_desc(w, h, samp)
    assert (d.nbmcuH, d.nbmcuV) == mcu
    od = helpers.oracle_desc(w, h, samp)
    assert (od.nbmcuH, od.nbmcuV, od.max_h, od.max_v) == (d.nbmcuH, d.nbmcuV, d.max_h, d.max_v)
    assert jg.raster_rows(d) == d.nbmcuV * d.max_v * 8
    assert jg.raster_row_bytes(d) == w * (3 if len(samp) == 3 else 1)
    for c in range(len(samp)):
        assert jg.plane_blocks(d, c) == d.nbmcuH * samp[c][0] * d.nbmcuV * samp[c][1]


def test_err_nb_comp_keeps_reference_code():
    # src/decoder_utils.c:50,67 -> ERR_NB_COMP == 13 in include/erreur.h
    d = jg.make_desc(16, 16, ((1, 1),))
    d.nbcomp = 2
    with pytest.raises(jg.JpegGpuError) as e:
        jg.backend_run(d, [np.zeros((4, 64), np.int16)] * 2, helpers.qtables_ramp())
    assert e.value.code == jg.ERR_NB_COMP == 13
    assert "composante" in jg.last_error()


def test_validation_errors_need_no_gpu():
    qt = helpers.qtables_ramp()
    d = jg.make_desc(16, 16, ((2, 2), (1, 1), (1, 1)))
    planes = [np.zeros((jg.plane_blocks(d, c), 64), np.int16) for c in range(3)]
    d.comp[1].idq = 3                                   # table 3 is NULL: src/progressive.c:262-267
    with pytest.raises(jg.JpegGpuError) as e:
        jg.backend_run(d, planes, qt)
    assert e.value.code == jg.ERR_NO_QT
    d = jg.make_desc(16, 16, ((2, 2), (1, 1), (1, 1)))
    d.comp[1].hs = 3                                    # 2 % 3 != 0: reference would index out of bounds
    with pytest.raises(jg.JpegGpuError) as e:
        jg.backend_run(d, planes, qt)
    assert e.value.code == jg.ERR_SAMPLING
    d = jg.make_desc(16, 16, ((1, 1),))
    d.width = 0
    with pytest.raises(jg.JpegGpuError) as e:
        jg.backend_run(d, planes[:1], qt)
    assert e.value.code == jg.ERR_INVALID_ARG


def test_empty_stage_calls_are_noops():
    assert jg.izz_blocks(np.zeros((0, 64), np.int16)).shape == (0, 8, 8)
    assert jg.idct_opt_blocks(np.zeros((0, 8, 8), np.int16)).shape == (0, 8, 8)
    assert jg.ycc2rgb_pixels([], [], []).shape == (0, 3)


@pytest.mark.skipif(has_gpu(), reason="this check is for hosts without a GPU")
def test_no_cpu_fallback_without_a_gpu():
    d = jg.make_desc(16, 16, ((1, 1),))
    with pytest.raises(jg.JpegGpuError) as e:
        jg.backend_run(d, [np.zeros((4, 64), np.int16)], helpers.qtables_ramp())
    assert e.value.code in (jg.ERR_NO_DEVICE, jg.ERR_HIP)
    with pytest.raises(jg.JpegGpuError):
        jg.idct_opt_blocks(np.zeros((1, 8, 8), np.int16))
