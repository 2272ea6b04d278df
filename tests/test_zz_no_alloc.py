"""GPU test of jpeggpu_backend_run_noalloc (the allocation-free variant of jpeggpu_backend_run whose device staging
is a static array in the library's device image): through the C ABI, bit-exact against jpeggpu_backend_run on two
calls in a row, the second with a smaller input; over-limit frames are refused.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

import helpers
from oracle import pyoracle

pytestmark = pytest.mark.gpu
QT = helpers.qtables_ramp()
orc = pyoracle.oracle


def _call_noalloc(gpu, gd, planes, qts, out, out_stride=0, ctx=None):
    lib = gpu.lib
    vp = C.c_void_p
    lib.jpeggpu_backend_run_noalloc.argtypes = [vp, C.POINTER(gpu.FrameDesc), C.POINTER(vp), C.POINTER(vp), vp, C.c_size_t]
    lib.jpeggpu_backend_run_noalloc.restype = C.c_int
    keep = [np.ascontiguousarray(p, dtype=np.int16) for p in planes]
    cp = (vp * 3)(*([a.ctypes.data for a in keep] + [None] * (3 - len(keep))))
    qk = [None if q is None else np.ascontiguousarray(q, dtype=np.uint16) for q in qts]
    qp = (vp * 4)(*[None if q is None else q.ctypes.data for q in qk])
    return lib.jpeggpu_backend_run_noalloc(ctx, C.byref(gd), cp, qp, out.ctypes.data, out_stride)


def test_noalloc_entry_equals_backend_run_twice_in_a_row(gpu):
    # first a frame that nearly fills the static buffer, then a much smaller one: the second call runs over the
    # first call's left-overs in the shared array
    for (w, h, samp, seed) in ((1920, 1080, ((2, 2), (1, 1), (1, 1)), 71), (301, 75, ((1, 1),) * 3, 72), (64, 8, ((1, 1),), 73)):
        d = helpers.oracle_desc(w, h, samp)
        planes = helpers.synth_planes(d, seed)
        gd = helpers.to_gpu_desc(gpu, d)
        want = gpu.backend_run(gd, planes, QT)                       # the existing entry point
        assert np.array_equal(want, orc.backend_run(d, planes, QT)), (w, h, samp)      # ... and the oracle, directly
        got = np.full_like(want, 0x5A)
        rc = _call_noalloc(gpu, gd, planes, QT, got)
        assert rc == 0, gpu.last_error()
        assert np.array_equal(got, want), (w, h, samp)
    # padded output rows
    stride = want.shape[1] + 11
    buf = np.full((want.shape[0], stride), 0x77, dtype=np.uint8)
    assert _call_noalloc(gpu, gd, planes, QT, buf, stride) == 0, gpu.last_error()
    assert np.array_equal(buf[:, :want.shape[1]], want) and np.all(buf[:, want.shape[1]:] == 0x77)


def test_noalloc_entry_on_a_context_and_its_limit(gpu):
    d = helpers.oracle_desc(333, 70, ((2, 1), (1, 1), (1, 1)))
    planes = helpers.synth_planes(d, 74)
    gd = helpers.to_gpu_desc(gpu, d)
    want = gpu.backend_run(gd, planes, QT)
    with gpu.Context(0) as ctx:
        got = np.zeros_like(want)
        assert _call_noalloc(gpu, gd, planes, QT, got, ctx=ctx.handle) == 0, gpu.last_error()
        assert np.array_equal(got, want)
    # 4096x4096 4:4:4 needs 151 MB of staging: refused with a clear error, nothing is touched
    big = gpu.make_desc(4096, 4096, ((1, 1),) * 3)
    one = np.zeros((1, 64), np.int16)
    out = np.zeros(16, np.uint8)
    rc = _call_noalloc(gpu, big, [one, one, one], QT, out)
    assert rc == gpu.ERR_INVALID_ARG and "static buffer" in gpu.last_error()


def test_noalloc_entry_leaves_device_and_pinned_memory_alone(gpu):
    """The entry's defining property: no hipMalloc / hipHostMalloc behind it.  Free device memory (hipMemGetInfo) is the same
    before and after two calls in a row; also with idct_fast == 0 once its table is on the device; results equal the oracle."""
    import torch
    d = helpers.oracle_desc(640, 360, ((2, 2), (1, 1), (1, 1)))
    planes = helpers.synth_planes(d, 75)
    gd = helpers.to_gpu_desc(gpu, d)
    want = orc.backend_run(d, planes, QT)
    got = np.zeros_like(want)
    assert _call_noalloc(gpu, gd, planes, QT, got) == 0, gpu.last_error()       # everything lazy (code object, streams) is set up now
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for (w, h) in ((640, 360), (200, 100)):
        d2 = helpers.oracle_desc(w, h, ((2, 2), (1, 1), (1, 1)))
        p2 = helpers.synth_planes(d2, 76 + w)
        o2 = np.zeros_like(orc.backend_run(d2, p2, QT))
        assert _call_noalloc(gpu, helpers.to_gpu_desc(gpu, d2), p2, QT, o2) == 0, gpu.last_error()
        assert np.array_equal(o2, orc.backend_run(d2, p2, QT)), (w, h)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    # "-f": the direct transform needs its table on the device; an ordinary call puts it there, then the entry may run
    gpu.set_idct_fast(False)
    try:
        orc.set_idct_fast(False)
        want_slow = orc.backend_run(d, planes, QT)
        assert np.array_equal(gpu.backend_run(gd, planes, QT), want_slow)
        torch.cuda.synchronize()
        free2, _ = torch.cuda.mem_get_info()
        got = np.zeros_like(want)
        assert _call_noalloc(gpu, gd, planes, QT, got) == 0, gpu.last_error()
        assert np.array_equal(got, want_slow)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free2
    finally:
        gpu.set_idct_fast(True)
        orc.set_idct_fast(True)
