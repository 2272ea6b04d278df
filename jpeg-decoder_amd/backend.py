"""ctypes binding of include/jpeggpu.h.

Host-side mirror of the reference's L1b interface for Python callers (tests, bench): same names
and argument meaning as the C ABI, numpy arrays for the reference's block types:

    blocl16_t[n] -> int16  array (n, 64)      zigzag order      (include/jpeg2ppm.h:31-34)
    bloct16_t[n] -> int16  array (n, 8, 8)    data[col][row]    (include/jpeg2ppm.h:37-40)
    bloctu8_t[n] -> uint8  array (n, 8, 8)    data[col][row]    (include/jpeg2ppm.h:25-28)
    qtable_t     -> uint16 array (64,)        zigzag order      (include/img.h:8-11)

There is no fallback: if libjpeggpu.so is missing the import fails, and every data-producing call
raises JpegGpuError when no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

SUCCESS, ERR_NB_COMP = 0, 13
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_SAMPLING, ERR_NO_QT = -1, -2, -3, -4, -5


class JpegGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jpeggpu error {code}: {msg}")
        self.code = code


class Component(C.Structure):
    _fields_ = [("hs", C.c_uint8), ("vs", C.c_uint8), ("idq", C.c_uint8), ("_pad", C.c_uint8)]


class FrameDesc(C.Structure):
    """jpeggpu_frame_desc_t (fields of img_t the back end needs, include/img.h:88-103)."""
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("nbcomp", C.c_uint8),
                ("max_h", C.c_uint8), ("max_v", C.c_uint8), ("apply_iquant", C.c_uint8),
                ("nbmcuH", C.c_uint32), ("nbmcuV", C.c_uint32), ("comp", Component * 3)]


lib_path = os.environ.get("JPEGGPU_LIB") or os.path.join(os.path.dirname(os.path.realpath(__file__)), "libjpeggpu.so")
if not os.path.exists(lib_path):
    raise ImportError(f"{lib_path} is missing: build it with `make -C jpeg-decoder_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's).  If libjpeggpu.so pulled in the system copy and
    torch later loaded its own, the process would hold two runtimes and torch would see no GPU.
    Loading torch's copy first (without importing torch) makes both resolve to the same object,
    whatever the import order."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


_preload_hip_runtime()
lib = C.CDLL(lib_path)

_vp, _sz, _u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
lib.jpeggpu_last_error.restype = C.c_char_p
lib.jpeggpu_init.argtypes = [C.c_int]
lib.jpeggpu_frame_geometry.argtypes = [C.POINTER(FrameDesc)]
for _n in ("jpeggpu_raster_rows", "jpeggpu_raster_row_bytes"):
    getattr(lib, _n).restype = _sz
    getattr(lib, _n).argtypes = [C.POINTER(FrameDesc)]
lib.jpeggpu_plane_blocks.restype = _sz
lib.jpeggpu_plane_blocks.argtypes = [C.POINTER(FrameDesc), C.c_int]
lib.jpeggpu_backend_run.argtypes = [C.POINTER(FrameDesc), C.POINTER(_vp), C.POINTER(_vp), _vp, _sz]
lib.jpeggpu_backend_run_device.argtypes = [C.POINTER(FrameDesc), C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp]
lib.jpeggpu_backend_run_batch_device.argtypes = [C.POINTER(FrameDesc), C.c_uint32, C.POINTER(_vp), C.POINTER(_sz),
                                                 C.POINTER(_vp), _vp, _sz, _sz, _vp]
lib.jpeggpu_synchronize.argtypes = [_vp]
lib.jpeggpu_iquant_blocks.argtypes = [_vp, _sz, C.c_uint8, C.c_uint8, _vp]
lib.jpeggpu_izz_blocks.argtypes = [_vp, _vp, _sz]
lib.jpeggpu_idct_opt_blocks.argtypes = [_vp, _vp, _sz]
lib.jpeggpu_ycc2rgb_pixels.argtypes = [_vp, _vp, _vp, _vp, _sz]
lib.jpeggpu_idct_blocks.argtypes = [_vp, _vp, _sz]
lib.jpeggpu_idct_coef_table.argtypes = [_vp]
lib.jpeggpu_idct_coef_table.restype = None
lib.jpeggpu_set_idct_fast.argtypes = [C.c_int]
lib.jpeggpu_set_idct_fast.restype = None
lib.jpeggpu_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
lib.jpeggpu_ctx_destroy.argtypes = [_vp]
lib.jpeggpu_ctx_destroy.restype = None
lib.jpeggpu_ctx_device.argtypes = [_vp]
lib.jpeggpu_ctx_set_idct_fast.argtypes = [_vp, C.c_int]
lib.jpeggpu_ctx_set_idct_fast.restype = None
lib.jpeggpu_ctx_get_idct_fast.argtypes = [_vp]
lib.jpeggpu_ctx_backend_run.argtypes = [_vp] + lib.jpeggpu_backend_run.argtypes
lib.jpeggpu_ctx_backend_run_device.argtypes = [_vp] + lib.jpeggpu_backend_run_device.argtypes
lib.jpeggpu_ctx_backend_run_batch_device.argtypes = [_vp] + lib.jpeggpu_backend_run_batch_device.argtypes
lib.jpeggpu_ctx_synchronize.argtypes = [_vp, _vp]
lib.jpeggpu_multi_backend_run.argtypes = [C.POINTER(_vp), C.c_int, C.POINTER(FrameDesc), C.POINTER(_vp), C.POINTER(_vp), _vp, _sz]
lib.jpeggpu_multi_backend_run_batch.argtypes = [C.POINTER(_vp), C.c_int, C.POINTER(FrameDesc), C.c_uint32, C.POINTER(_vp), C.POINTER(_sz),
                                                C.POINTER(_vp), _vp, _sz, _sz]
lib.jpeggpu_backend_run_batch.argtypes = [C.POINTER(FrameDesc), C.c_uint32, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), _vp, _sz, _sz]
lib.jpeggpu_ctx_backend_run_batch.argtypes = [_vp, C.POINTER(FrameDesc), C.c_uint32, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), _vp, _sz, _sz]
lib.jpeggpu_host_alloc.argtypes = [_sz]
lib.jpeggpu_host_alloc.restype = _vp
lib.jpeggpu_host_free.argtypes = [_vp]
lib.jpeggpu_host_free.restype = None
lib.jpeggpu_internal_reload_tuning.restype = None


def last_error():
    return lib.jpeggpu_last_error().decode("utf-8", "replace")


def _check(rc):
    if rc != 0:
        raise JpegGpuError(rc, last_error())


def init(device=0):
    _check(lib.jpeggpu_init(int(device)))


def shutdown():
    lib.jpeggpu_shutdown()


def synchronize(stream=None):
    _check(lib.jpeggpu_synchronize(_vp(stream or 0)))


def make_desc(width, height, sampling=((1, 1),), idq=None, apply_iquant=True):
    """sampling: one (hs, vs) per component in frame order Y[, Cb, Cr].  Geometry is derived like
    calcul_image_information (src/entete.c:176-197)."""
    d = FrameDesc()
    d.width, d.height, d.nbcomp = width, height, len(sampling)
    d.apply_iquant = 1 if apply_iquant else 0
    for c, (hs, vs) in enumerate(sampling[:3]):
        d.comp[c].hs, d.comp[c].vs = hs, vs
        d.comp[c].idq = (idq[c] if idq is not None else (0 if c == 0 else 1))
    _check(lib.jpeggpu_frame_geometry(C.byref(d)))
    return d


def plane_blocks(desc, c):
    return lib.jpeggpu_plane_blocks(C.byref(desc), c)


def raster_rows(desc):
    return lib.jpeggpu_raster_rows(C.byref(desc))


def raster_row_bytes(desc):
    return lib.jpeggpu_raster_row_bytes(C.byref(desc))


def _qt_array(qtables):
    """qtables: sequence of up to 4 entries, each None or a (64,) uint16 array."""
    keep, ptrs = [], (_vp * 4)()
    for i in range(4):
        q = qtables[i] if qtables is not None and i < len(qtables) else None
        if q is None:
            ptrs[i] = None
        else:
            a = np.ascontiguousarray(q, dtype=np.uint16).reshape(64)
            keep.append(a)
            ptrs[i] = a.ctypes.data
    return ptrs, keep


def reload_tuning():
    """The JPEGGPU_* tuning knobs are read from the environment once; tests and tools that change them inside one
    process ask for them to be read again."""
    lib.jpeggpu_internal_reload_tuning()


class PinnedArray:
    """A numpy view of jpeggpu_host_alloc memory (pinned: jpeggpu_backend_run moves it by DMA in place)."""

    def __init__(self, shape, dtype):
        self.array = None
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib.jpeggpu_host_alloc(n)
        if not self.ptr:
            raise JpegGpuError(ERR_HIP, last_error())
        self.array = np.frombuffer((C.c_uint8 * n).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib.jpeggpu_host_free(self.ptr)
            self.ptr = None


def backend_run(desc, coef, qtables, out=None, out_stride=0, ctx=None):
    """jpeggpu_backend_run: host planes in, host raster out (rows x row_bytes uint8)."""
    nc = desc.nbcomp if desc.nbcomp in (1, 3) else min(len(coef), 3)
    planes = [np.ascontiguousarray(coef[c], dtype=np.int16) for c in range(min(nc, len(coef)))]
    cp = (_vp * 3)()
    for c, a in enumerate(planes):
        cp[c] = a.ctypes.data
    qp, keep = _qt_array(qtables)
    rows, rb = raster_rows(desc), raster_row_bytes(desc)
    stride = out_stride or rb
    if out is None:
        out = np.zeros((max(rows, 1), max(stride, 1)), dtype=np.uint8)
    if ctx is not None:
        _check(lib.jpeggpu_ctx_backend_run(ctx.handle, C.byref(desc), cp, qp, out.ctypes.data, out_stride))
    else:
        _check(lib.jpeggpu_backend_run(C.byref(desc), cp, qp, out.ctypes.data, out_stride))
    return out[:rows, :stride]


def backend_run_device(desc, d_coef, qtables, d_out, out_stride=0, stream=None, ctx=None):
    """Device pointers (ints) in, asynchronous on `stream` (a hipStream_t as int, None = library stream)."""
    cp = (_vp * 3)(*[int(p) if p else None for p in list(d_coef) + [None] * (3 - len(d_coef))])
    qp, keep = _qt_array(qtables)
    if ctx is not None:
        _check(lib.jpeggpu_ctx_backend_run_device(ctx.handle, C.byref(desc), cp, qp, _vp(int(d_out)), out_stride, _vp(stream or 0)))
    else:
        _check(lib.jpeggpu_backend_run_device(C.byref(desc), cp, qp, _vp(int(d_out)), out_stride, _vp(stream or 0)))


def backend_run_batch_device(desc, nframes, d_coef, coef_frame_stride, qtables, d_out, out_stride=0,
                             out_frame_stride=0, stream=None, ctx=None):
    cp = (_vp * 3)(*[int(p) if p else None for p in list(d_coef) + [None] * (3 - len(d_coef))])
    fs = None
    if coef_frame_stride is not None:
        fs = (_sz * 3)(*(list(coef_frame_stride) + [0] * (3 - len(coef_frame_stride))))
    qp, keep = _qt_array(qtables)
    if ctx is not None:
        _check(lib.jpeggpu_ctx_backend_run_batch_device(ctx.handle, C.byref(desc), nframes, cp, fs, qp, _vp(int(d_out)), out_stride,
                                                        out_frame_stride, _vp(stream or 0)))
    else:
        _check(lib.jpeggpu_backend_run_batch_device(C.byref(desc), nframes, cp, fs, qp, _vp(int(d_out)), out_stride,
                                                    out_frame_stride, _vp(stream or 0)))


def multi_backend_run(ctxs, desc, coef, qtables, out=None, out_stride=0):
    """jpeggpu_multi_backend_run: one frame, MCU-row stripes over the contexts (one host thread each)."""
    planes = [np.ascontiguousarray(coef[c], dtype=np.int16) for c in range(desc.nbcomp)]
    cp = (_vp * 3)(*([a.ctypes.data for a in planes] + [None] * (3 - len(planes))))
    qp, keep = _qt_array(qtables)
    rows, rb = raster_rows(desc), raster_row_bytes(desc)
    stride = out_stride or rb
    if out is None:
        out = np.zeros((rows, stride), dtype=np.uint8)
    hs = (_vp * len(ctxs))(*[c.handle for c in ctxs])
    _check(lib.jpeggpu_multi_backend_run(hs, len(ctxs), C.byref(desc), cp, qp, out.ctypes.data, out_stride))
    return out[:rows, :stride]


def backend_run_batch(desc, nframes, coef, qtables, out=None, out_stride=0, out_frame_stride=0, coef_frame_stride=None, ctx=None):
    """jpeggpu_backend_run_batch / jpeggpu_ctx_backend_run_batch: host memory, coef[c] = frame-major int16 array (frame f of plane
    c at byte offset f * coef_frame_stride[c], default dense); returns (nframes, rows, row_bytes) uint8 (or fills <out>).  Arrays
    are used where they lie (no copy) when they are C-contiguous int16 / uint8 -- pinned arrays stay pinned."""
    planes = [coef[c] if (coef[c].dtype == np.int16 and coef[c].flags["C_CONTIGUOUS"]) else np.ascontiguousarray(coef[c], dtype=np.int16)
              for c in range(desc.nbcomp)]
    cp = (_vp * 3)(*([a.ctypes.data for a in planes] + [None] * (3 - len(planes))))
    fs = None
    if coef_frame_stride is not None:
        fs = (_sz * 3)(*(list(coef_frame_stride) + [0] * (3 - len(coef_frame_stride))))
    qp, keep = _qt_array(qtables)
    rows, rb = raster_rows(desc), raster_row_bytes(desc)
    if out is None:
        out = np.zeros((nframes, rows, out_stride or rb), dtype=np.uint8)
    if ctx is not None:
        _check(lib.jpeggpu_ctx_backend_run_batch(ctx.handle, C.byref(desc), nframes, cp, fs, qp, out.ctypes.data, out_stride, out_frame_stride))
    else:
        _check(lib.jpeggpu_backend_run_batch(C.byref(desc), nframes, cp, fs, qp, out.ctypes.data, out_stride, out_frame_stride))
    return out


def multi_backend_run_batch(ctxs, desc, nframes, coef, qtables):
    """jpeggpu_multi_backend_run_batch: coef[c] = (nframes * blocks_c, 64) int16, frame-major; returns (nframes, rows, row_bytes)."""
    planes = [np.ascontiguousarray(coef[c], dtype=np.int16) for c in range(desc.nbcomp)]
    cp = (_vp * 3)(*([a.ctypes.data for a in planes] + [None] * (3 - len(planes))))
    qp, keep = _qt_array(qtables)
    rows, rb = raster_rows(desc), raster_row_bytes(desc)
    out = np.zeros((nframes, rows, rb), dtype=np.uint8)
    hs = (_vp * len(ctxs))(*[c.handle for c in ctxs])
    _check(lib.jpeggpu_multi_backend_run_batch(hs, len(ctxs), C.byref(desc), nframes, cp, None, qp, out.ctypes.data, 0, 0))
    return out


class Context:
    """jpeggpu_ctx_t: a per-device context (own stream, staging and idct_fast option).  Pass it as ctx= to the
    backend_run* functions; one per host thread that drives a GPU."""

    def __init__(self, device=0):
        h = _vp()
        _check(lib.jpeggpu_ctx_create(int(device), C.byref(h)))
        self.handle = h

    @property
    def device(self):
        return lib.jpeggpu_ctx_device(self.handle)

    def set_idct_fast(self, fast):
        lib.jpeggpu_ctx_set_idct_fast(self.handle, 1 if fast else 0)

    def get_idct_fast(self):
        return bool(lib.jpeggpu_ctx_get_idct_fast(self.handle))

    def synchronize(self, stream=None):
        _check(lib.jpeggpu_ctx_synchronize(self.handle, _vp(stream or 0)))

    def close(self):
        if self.handle:
            lib.jpeggpu_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def iquant_blocks(blocks, s_start, s_end, qtable):
    """iquant (src/iqzz.c:9-14) over an (n, 64) int16 array; returns a new array."""
    a = np.array(blocks, dtype=np.int16, order="C").reshape(-1, 64)
    q = np.ascontiguousarray(qtable, dtype=np.uint16).reshape(64)
    _check(lib.jpeggpu_iquant_blocks(a.ctypes.data, a.shape[0], s_start, s_end, q.ctypes.data))
    return a


def izz_blocks(blocks):
    a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
    out = np.zeros((a.shape[0], 8, 8), dtype=np.int16)
    _check(lib.jpeggpu_izz_blocks(a.ctypes.data, out.ctypes.data, a.shape[0]))
    return out


def idct_opt_blocks(blocks):
    a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 8, 8)
    out = np.zeros((a.shape[0], 8, 8), dtype=np.uint8)
    _check(lib.jpeggpu_idct_opt_blocks(a.ctypes.data, out.ctypes.data, a.shape[0]))
    return out


def idct_blocks(blocks):
    """idct (src/idct.c:50-71), the slow transform of the reference's -f option."""
    a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 8, 8)
    out = np.zeros((a.shape[0], 8, 8), dtype=np.uint8)
    _check(lib.jpeggpu_idct_blocks(a.ctypes.data, out.ctypes.data, a.shape[0]))
    return out


def idct_coef_table():
    t = np.zeros(4096, dtype=np.float32)
    lib.jpeggpu_idct_coef_table(t.ctypes.data)
    return t.reshape(8, 8, 8, 8)


def set_idct_fast(fast):
    lib.jpeggpu_set_idct_fast(1 if fast else 0)


def get_idct_fast():
    return bool(lib.jpeggpu_get_idct_fast())


def ycc2rgb_pixels(y, cb, cr):
    y, cb, cr = (np.ascontiguousarray(v, dtype=np.uint8).reshape(-1) for v in (y, cb, cr))
    out = np.zeros((y.shape[0], 3), dtype=np.uint8)
    _check(lib.jpeggpu_ycc2rgb_pixels(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, out.ctypes.data, y.shape[0]))
    return out
