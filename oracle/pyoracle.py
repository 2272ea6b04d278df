"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/liboracle.so (CPU restatement) and, when
present, oracle/_ref/libjpegref_*.so (the real reference objects built by oracle/Makefile).

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.realpath(__file__))


class Component(C.Structure):
    _fields_ = [("hs", C.c_uint8), ("vs", C.c_uint8), ("idq", C.c_uint8), ("_pad", C.c_uint8)]


class FrameDesc(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("nbcomp", C.c_uint8),
                ("max_h", C.c_uint8), ("max_v", C.c_uint8), ("apply_iquant", C.c_uint8),
                ("nbmcuH", C.c_uint32), ("nbmcuV", C.c_uint32), ("comp", Component * 3)]


_vp, _sz = C.c_void_p, C.c_size_t


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class _Backend:
    """Common calling code for liboracle (prefix 'oracle_') and libjpegref (prefix 'ref_')."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        f = getattr(lib, prefix + "backend_rows")
        f.argtypes = [C.POINTER(FrameDesc), C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, C.c_uint32, C.c_uint32]
        getattr(lib, prefix + "iquant_blocks").argtypes = [_vp, _sz, C.c_int, C.c_int, _vp]
        getattr(lib, prefix + "izz_blocks").argtypes = [_vp, _vp, _sz]
        getattr(lib, prefix + "idct_opt_blocks").argtypes = [_vp, _vp, _sz]
        getattr(lib, prefix + "ycc2rgb_pixels").argtypes = [_vp, _vp, _vp, _vp, _sz]
        getattr(lib, prefix + "idct_blocks").argtypes = [_vp, _vp, _sz]
        getattr(lib, prefix + "idct_coef").argtypes = [_vp]
        getattr(lib, prefix + "set_idct_fast").argtypes = [C.c_int]

    def backend_run(self, desc, coef, qtables, mcu_rows=None):
        d = FrameDesc.from_buffer_copy(bytes(desc))
        planes = [np.ascontiguousarray(a, dtype=np.int16) for a in coef[:d.nbcomp]]
        cp = (_vp * 3)()
        for c, a in enumerate(planes):
            cp[c] = a.ctypes.data
        keep, qp = [], (_vp * 4)()
        for i in range(4):
            q = qtables[i] if qtables is not None and i < len(qtables) else None
            if q is not None:
                a = np.ascontiguousarray(q, dtype=np.uint16).reshape(64)
                keep.append(a)
                qp[i] = a.ctypes.data
        rows = d.nbmcuV * d.max_v * 8
        rb = d.width * (3 if d.nbcomp == 3 else 1)
        out = np.zeros((rows, rb), dtype=np.uint8)
        m0, m1 = mcu_rows if mcu_rows is not None else (0, d.nbmcuV)
        rc = getattr(self.lib, self.p + "backend_rows")(C.byref(d), cp, qp, out.ctypes.data, 0, m0, m1)
        if rc != 0:
            raise RuntimeError(f"{self.p}backend_rows returned {rc}")
        return out

    def iquant_blocks(self, blocks, s0, s1, q):
        a = np.array(blocks, dtype=np.int16, order="C").reshape(-1, 64)
        q = np.ascontiguousarray(q, dtype=np.uint16).reshape(64)
        getattr(self.lib, self.p + "iquant_blocks")(a.ctypes.data, a.shape[0], s0, s1, q.ctypes.data)
        return a

    def izz_blocks(self, blocks):
        a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
        out = np.zeros((a.shape[0], 8, 8), dtype=np.int16)
        getattr(self.lib, self.p + "izz_blocks")(a.ctypes.data, out.ctypes.data, a.shape[0])
        return out

    def idct_opt_blocks(self, blocks):
        a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 8, 8)
        out = np.zeros((a.shape[0], 8, 8), dtype=np.uint8)
        getattr(self.lib, self.p + "idct_opt_blocks")(a.ctypes.data, out.ctypes.data, a.shape[0])
        return out

    def idct_blocks(self, blocks):
        """The slow transform of -f (src/idct.c:50-71)."""
        a = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 8, 8)
        out = np.zeros((a.shape[0], 8, 8), dtype=np.uint8)
        getattr(self.lib, self.p + "idct_blocks")(a.ctypes.data, out.ctypes.data, a.shape[0])
        return out

    def idct_coef(self):
        t = np.zeros((8, 8, 8, 8), dtype=np.float32)
        getattr(self.lib, self.p + "idct_coef")(t.ctypes.data)
        return t

    def set_idct_fast(self, fast):
        """all_option.idct_fast for backend_run: True = idct_opt, False = idct."""
        getattr(self.lib, self.p + "set_idct_fast")(1 if fast else 0)

    def ycc2rgb_pixels(self, y, cb, cr):
        y, cb, cr = (np.ascontiguousarray(v, dtype=np.uint8).reshape(-1) for v in (y, cb, cr))
        out = np.zeros((y.shape[0], 3), dtype=np.uint8)
        getattr(self.lib, self.p + "ycc2rgb_pixels")(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, out.ctypes.data, y.shape[0])
        return out


_olib = _load(os.path.join(HERE, "liboracle.so"))
if _olib is None:
    raise ImportError("oracle/liboracle.so missing: run `make -C oracle oracle`")
_olib.oracle_synth_blocks.argtypes = [_vp, _sz, C.c_uint64, C.c_uint64, C.c_int]
_olib.oracle_frame_geometry.argtypes = [C.POINTER(FrameDesc)]

oracle = _Backend(_olib, "oracle_")


def load_fast_oracle():
    """-Ofast build of the same restatement: timing stand-in only, never a parity oracle."""
    lib = _load(os.path.join(HERE, "liboracle_fast.so"))
    return _Backend(lib, "oracle_") if lib is not None else None


def load_ref(kind="strict"):
    """The real reference objects (None when oracle/_ref was not built / did not travel)."""
    lib = _load(os.path.join(HERE, "_ref", f"libjpegref_{kind}.so"))
    return _Backend(lib, "ref_") if lib is not None else None


def synth_blocks(n, seed, first_block=0, kind=0):
    """Deterministic synthetic zigzag coefficient blocks, (n, 64) int16."""
    a = np.zeros((n, 64), dtype=np.int16)
    if n:
        _olib.oracle_synth_blocks(a.ctypes.data, n, seed, first_block, kind)
    return a


def frame_geometry(desc):
    d = FrameDesc.from_buffer_copy(bytes(desc))
    rc = _olib.oracle_frame_geometry(C.byref(d))
    return rc, d
