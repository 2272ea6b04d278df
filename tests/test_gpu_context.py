"""GPU suite: per-device contexts (jpeggpu_ctx_*), the threading contract, and the pipelined host-pointer form
(stripes over an upload and a download stream; pinned caller memory moved in place, pageable memory staged, large
pageable outputs emptied by a second host thread).  Bit-exact vs the oracle."""
import os
import threading

import numpy as np
import pytest

import helpers
from oracle import pyoracle

pytestmark = pytest.mark.gpu
orc = pyoracle.oracle
QT = helpers.qtables_ramp()


def _frame(w, h, samp, seed, kind=0):
    d = helpers.oracle_desc(w, h, samp)
    planes = helpers.synth_planes(d, seed, kind)
    return d, planes, orc.backend_run(d, planes, QT)


def _page_aligned(a):
    """A copy of `a` on pages of its own (hipHostRegister pins whole pages; two buffers must not share one)."""
    raw = np.empty(a.nbytes + 2 * 4096, dtype=np.uint8)
    o = (-raw.ctypes.data) % 4096
    v = raw[o:o + a.nbytes].view(a.dtype).reshape(a.shape)
    v[...] = a
    return v


def test_two_contexts_from_two_threads(gpu):
    """Two contexts on device 0, each driven by its own thread (the shape of 'one host thread per GPU'), many frames
    each, host-pointer and device-pointer forms: every raster equals the oracle's."""
    import torch
    cases = [_frame(333, 70, ((2, 2), (1, 1), (1, 1)), 21), _frame(500, 41, ((1, 1),) * 3, 22)]
    errors = []

    def worker(idx):
        try:
            d, planes, want = cases[idx]
            gd = helpers.to_gpu_desc(gpu, d)
            with gpu.Context(0) as ctx:
                assert ctx.device == 0
                for _ in range(12):
                    got = gpu.backend_run(gd, planes, QT, ctx=ctx)
                    assert np.array_equal(got, want)
                dev = [torch.from_numpy(p).cuda() for p in planes]
                out = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                for _ in range(12):
                    gpu.backend_run_device(gd, [t.data_ptr() for t in dev], QT, out.data_ptr(), ctx=ctx)
                ctx.synchronize()
                assert np.array_equal(out.cpu().numpy().reshape(want.shape), want)
        except BaseException as e:   # noqa: BLE001  (reported by the main thread)
            errors.append((idx, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


def test_process_wide_entry_from_a_second_thread(gpu):
    """HIP's current device is per thread: an entry point called from a thread that never called jpeggpu_init binds
    itself to the context's device."""
    d, planes, want = _frame(129, 33, ((2, 1), (1, 1), (1, 1)), 23)
    res = {}

    def worker():
        try:
            res["got"] = gpu.backend_run(helpers.to_gpu_desc(gpu, d), planes, QT)
            res["idct"] = gpu.idct_opt_blocks(np.zeros((3, 8, 8), np.int16))
        except BaseException as e:   # noqa: BLE001
            res["err"] = repr(e)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert "err" not in res, res
    assert np.array_equal(res["got"], want) and np.all(res["idct"] == 128)


def test_context_option_is_per_context(gpu):
    """idct_fast is part of the context: a '-f' context and a default one side by side."""
    d = helpers.oracle_desc(64, 16, ((1, 1),))
    planes = helpers.synth_planes(d, 24)
    gd = helpers.to_gpu_desc(gpu, d)
    fast = orc.backend_run(d, planes, QT)
    orc.set_idct_fast(False)
    try:
        slow = orc.backend_run(d, planes, QT)
    finally:
        orc.set_idct_fast(True)
    with gpu.Context(0) as a, gpu.Context(0) as b:
        b.set_idct_fast(False)
        assert a.get_idct_fast() and not b.get_idct_fast() and gpu.get_idct_fast()
        ra = gpu.backend_run(gd, planes, QT, ctx=a)
        rb = gpu.backend_run(gd, planes, QT, ctx=b)
        assert np.array_equal(ra, fast)
        assert np.array_equal(rb, slow)
        assert np.max(np.abs(ra.astype(int) - rb.astype(int))) <= 1    # test/idct_opt_test.c:35
        assert np.array_equal(gpu.backend_run(gd, planes, QT), fast)  # the default context is untouched


@pytest.mark.parametrize("nctx", [1, 2, 3, 8])
def test_one_process_several_contexts_stripes_and_batches(gpu, nctx):
    """jpeggpu_multi_backend_run / _batch: the shards of SURVEY 8e driven from ONE process, one host thread per context
    (contexts on device 0 here; one per device on a multi-GPU host -- the same code path)."""
    import torch
    ndev = torch.cuda.device_count()
    ctxs = [gpu.Context(i % ndev) for i in range(nctx)]
    try:
        for samp, (w, h) in ((((2, 2), (1, 1), (1, 1)), (1000, 333)), (((1, 1),) * 3, (517, 50)), (((1, 1),), (300, 17))):
            d, planes, want = _frame(w, h, samp, 41)     # 17 rows of gray = 3 MCU rows: fewer stripes than contexts when nctx = 8
            gd = helpers.to_gpu_desc(gpu, d)
            assert np.array_equal(gpu.multi_backend_run(ctxs, gd, planes, QT), want), (nctx, samp)
        d = helpers.oracle_desc(333, 70, ((2, 1), (1, 1), (1, 1)))
        gd = helpers.to_gpu_desc(gpu, d)
        nf = 5
        nb = [helpers.plane_shape(d, c)[0] for c in range(3)]
        host = [pyoracle.synth_blocks(nb[c] * nf, 45 + c) for c in range(3)]
        got = gpu.multi_backend_run_batch(ctxs, gd, nf, host, QT)
        for f in range(nf):
            assert np.array_equal(got[f], orc.backend_run(d, [host[c][f * nb[c]:(f + 1) * nb[c]] for c in range(3)], QT)), (nctx, f)
    finally:
        [c.close() for c in ctxs]


def test_ctx_errors(gpu):
    with pytest.raises(gpu.JpegGpuError) as e:
        gpu.Context(9999)
    assert e.value.code == gpu.ERR_NO_DEVICE


@pytest.mark.parametrize("stripe", [1, 200000, 0], ids=["one_row_stripes", "200kB_stripes", "default"])
@pytest.mark.parametrize("pinned", [None, "alloc"], ids=["pageable", "host_alloc"])
def test_host_pointer_pipeline(gpu, stripe, pinned):
    """jpeggpu_backend_run with many / few / one stripe, caller memory pageable (staged) or pinned (jpeggpu_host_alloc: DMA in place),
    packed and padded output rows."""
    old = os.environ.get("JPEGGPU_STRIPE_BYTES")
    if stripe:
        os.environ["JPEGGPU_STRIPE_BYTES"] = str(stripe)
    gpu.reload_tuning()
    try:
        for samp, (w, h) in ((((2, 2), (1, 1), (1, 1)), (1000, 333)), (((1, 1),) * 3, (517, 130)), (((1, 1),), (2049, 97))):
            d, planes, want = _frame(w, h, samp, 31)
            gd = helpers.to_gpu_desc(gpu, d)
            rb = want.shape[1]
            for stride in (0, rb + 37):
                buf = np.full((want.shape[0], stride or rb), 0xCD, dtype=np.uint8)
                src, held = planes, []
                if pinned == "alloc":
                    held = [gpu.PinnedArray(p.shape, p.dtype) for p in planes] + [gpu.PinnedArray(buf.shape, buf.dtype)]
                    for hp, p in zip(held, planes + [buf]):
                        hp.array[...] = p
                    src, buf = [hp.array for hp in held[:-1]], held[-1].array
                try:
                    got = gpu.backend_run(gd, src, QT, out=buf, out_stride=stride).copy()
                    pad = buf[:, rb:].copy()
                finally:
                    del src, buf
                    [hp.free() for hp in held]
                assert np.array_equal(got[:, :rb], want), (samp, w, h, stride)
                assert np.all(pad == 0xCD)
    finally:
        if old is None:
            os.environ.pop("JPEGGPU_STRIPE_BYTES", None)
        else:
            os.environ["JPEGGPU_STRIPE_BYTES"] = old
        gpu.reload_tuning()


@pytest.mark.parametrize("stripe", [1 << 20, 0], ids=["1MiB_stripes", "default"])
@pytest.mark.parametrize("mix", ["pageable", "pinned_in", "pinned_out"])
def test_host_pointer_helper_thread(gpu, stripe, mix):
    """A frame whose raster is above 16 MiB in three stripes and more: with a pageable raster the download staging is emptied
    by a second host thread (capi.cpp run_host).  Pageable and mixed pinned / pageable caller memory, packed and padded
    rows, twice in a row on the same context (the staging and the event rings are reused)."""
    old = os.environ.get("JPEGGPU_STRIPE_BYTES")
    if stripe:
        os.environ["JPEGGPU_STRIPE_BYTES"] = str(stripe)
    gpu.reload_tuning()
    try:
        d, planes, want = _frame(4096, 1400, ((2, 2), (1, 1), (1, 1)), 77)
        assert want.nbytes >= 16 << 20
        gd = helpers.to_gpu_desc(gpu, d)
        rb = want.shape[1]
        for stride in (0, rb + 64):
            buf = np.full((want.shape[0], stride or rb), 0xCD, dtype=np.uint8)
            src, held = planes, []
            if mix == "pinned_in":
                held = [gpu.PinnedArray(p.shape, p.dtype) for p in planes]
                for hp, p in zip(held, planes):
                    hp.array[...] = p
                src = [hp.array for hp in held]
            elif mix == "pinned_out":
                held = [gpu.PinnedArray(buf.shape, buf.dtype)]
                held[0].array[...] = buf
                buf = held[0].array
            try:
                for _ in range(2):
                    buf[:, :rb] = 0
                    got = gpu.backend_run(gd, src, QT, out=buf, out_stride=stride).copy()
                    assert np.array_equal(got[:, :rb], want), (mix, stride)
                    assert np.all(buf[:, rb:] == 0xCD)
            finally:
                del src, buf
                [hp.free() for hp in held]
    finally:
        if old is None:
            os.environ.pop("JPEGGPU_STRIPE_BYTES", None)
        else:
            os.environ["JPEGGPU_STRIPE_BYTES"] = old
        gpu.reload_tuning()


@pytest.mark.parametrize("mem", ["pageable", "pinned"])
@pytest.mark.parametrize("stripe,staging", [(0, 0), (60000, 0), (60000, 400000)], ids=["default", "60kB_stripes", "60kB_stripes_small_ring"])
def test_host_pointer_batch(gpu, mem, stripe, staging):
    """jpeggpu_backend_run_batch: N frames in host memory through ONE pipeline (units = stripes of all frames).  Dense and
    padded strides (planes, rows, frames), pageable and pinned memory, on the default context and on a context of its
    own, more units than staging buffers and -- with a small device-side ring -- than device slots; every frame equals
    the oracle's."""
    old, old2 = os.environ.get("JPEGGPU_STRIPE_BYTES"), os.environ.get("JPEGGPU_DEVICE_STAGING_BYTES")
    if stripe:
        os.environ["JPEGGPU_STRIPE_BYTES"] = str(stripe)
    if staging:
        os.environ["JPEGGPU_DEVICE_STAGING_BYTES"] = str(staging)
    gpu.reload_tuning()
    held = []

    def place(a):
        if mem == "pageable":
            return a
        h = gpu.PinnedArray(a.shape, a.dtype)
        h.array[...] = a
        held.append(h)
        return h.array

    try:
        for samp, (w, h), nf in ((((2, 2), (1, 1), (1, 1)), (200, 120), 7), (((1, 1),) * 3, (97, 33), 5), (((1, 1),), (130, 50), 4)):
            d = helpers.oracle_desc(w, h, samp)
            gd = helpers.to_gpu_desc(gpu, d)
            nb = [helpers.plane_shape(d, c)[0] for c in range(d.nbcomp)]
            pad = 3                                                           # blocks of padding between frames of a plane
            host = [pyoracle.synth_blocks((nb[c] + pad) * nf, 900 + c).reshape(nf, nb[c] + pad, 64) for c in range(d.nbcomp)]
            want = [orc.backend_run(d, [host[c][f, :nb[c]] for c in range(d.nbcomp)], QT) for f in range(nf)]
            rows, rb = want[0].shape
            for dense in (True, False):
                if dense:
                    planes = [place(np.ascontiguousarray(host[c][:, :nb[c]])) for c in range(d.nbcomp)]
                    cfs, ostride, ofs = None, 0, 0
                    out = place(np.full((nf, rows, rb), 0xCD, dtype=np.uint8))
                else:
                    planes = [place(host[c]) for c in range(d.nbcomp)]
                    cfs = [(nb[c] + pad) * 128 for c in range(d.nbcomp)]
                    ostride, ofs = rb + 19, (rows + 2) * (rb + 19)
                    out = place(np.full((nf, rows + 2, rb + 19), 0xCD, dtype=np.uint8))
                for ctx in (None, gpu.Context(0)):
                    for _ in range(2):
                        out[...] = 0xCD
                        gpu.backend_run_batch(gd, nf, planes, QT, out=out, out_stride=ostride, out_frame_stride=ofs, coef_frame_stride=cfs, ctx=ctx)
                        for f in range(nf):
                            assert np.array_equal(out[f, :rows, :rb], want[f]), (samp, dense, f)
                        if not dense:
                            assert np.all(out[:, rows:, :] == 0xCD) and np.all(out[:, :, rb:] == 0xCD)
                    if ctx is not None:
                        ctx.close()
    finally:
        [h.free() for h in held]
        for name, val in (("JPEGGPU_STRIPE_BYTES", old), ("JPEGGPU_DEVICE_STAGING_BYTES", old2)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
        gpu.reload_tuning()
