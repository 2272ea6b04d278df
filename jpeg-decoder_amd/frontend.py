"""ctypes binding of include/jpeggpu_decode.h: the host front end (header parser + Huffman VLD into
full-frame planes), the per-scan hand-off of progressive files and the PPM/PGM writer.

Mirrors the reference's L2 drivers for Python callers:

    decode_entete(f, True, img)            -> Jpeg(data) / Jpeg.from_file(path)     (src/entete.c:199-248)
    decode_progressif_dc/ac + finir_octet  -> Jpeg.decode_scan()                    (src/progressive.c:98-173)
    the block loop + save_mcu_ligne_*      -> Jpeg.render()              [GPU]      (src/progressive.c:257-312)
    decode_entete(f, False, img)           -> Jpeg.next_header()                    (src/progressive.c:322)
    main()                                 -> jpeg2ppm(path, outfile)    [GPU]      (src/jpeg2ppm.c:62-141)
"""
import ctypes as C

import numpy as np

from . import backend as B

lib = B.lib
_vp, _sz = C.c_void_p, C.c_size_t

ERR_UNSUPPORTED, ERR_IO = -6, -7


class ScanInfo(C.Structure):
    _fields_ = [("num_sof", C.c_uint8), ("nb_scan_comp", C.c_uint8), ("scan_comp", C.c_uint8 * 3),
                ("ss", C.c_uint8), ("se", C.c_uint8), ("ah", C.c_uint8), ("al", C.c_uint8), ("eoi", C.c_uint8),
                ("scans_decoded", C.c_uint16), ("mcu_rows_done", C.c_uint32)]


lib.jpeggpu_jpeg_open_memory.argtypes = [_vp, _sz, C.POINTER(_vp)]
lib.jpeggpu_jpeg_open_file.argtypes = [C.c_char_p, C.POINTER(_vp)]
lib.jpeggpu_jpeg_close.argtypes = [_vp]
lib.jpeggpu_jpeg_close.restype = None
lib.jpeggpu_jpeg_frame_desc.argtypes = [_vp, C.POINTER(B.FrameDesc)]
lib.jpeggpu_jpeg_scan_info.argtypes = [_vp, C.POINTER(ScanInfo)]
lib.jpeggpu_jpeg_qtable.argtypes = [_vp, C.c_int]
lib.jpeggpu_jpeg_qtable.restype = _vp
lib.jpeggpu_jpeg_plane.argtypes = [_vp, C.c_int]
lib.jpeggpu_jpeg_plane.restype = _vp
lib.jpeggpu_jpeg_nb_comments.argtypes = [_vp]
lib.jpeggpu_jpeg_comment.argtypes = [_vp, C.c_int]
lib.jpeggpu_jpeg_comment.restype = C.c_char_p
lib.jpeggpu_jpeg_decode_scan.argtypes = [_vp]
lib.jpeggpu_jpeg_next_header.argtypes = [_vp]
lib.jpeggpu_jpeg_render.argtypes = [_vp, _vp, _sz]
lib.jpeggpu_jpeg_uploaded_bytes.argtypes = [_vp]
lib.jpeggpu_jpeg_uploaded_bytes.restype = _sz
lib.jpeggpu_write_pnm.argtypes = [C.c_char_p, C.POINTER(B.FrameDesc), _vp, _sz]
lib.jpeggpu_write_pnm_rows.argtypes = [C.c_char_p, C.POINTER(B.FrameDesc), _vp, _sz, _sz]
lib.jpeggpu_out_file_name.argtypes = [C.c_char_p, C.c_char_p, C.c_uint8, C.c_uint8, C.c_char_p, _sz]
lib.jpeggpu_jpeg2ppm.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
lib.jpeggpu_internal_vld_force_checked.argtypes = [C.c_int]
lib.jpeggpu_internal_vld_force_checked.restype = None


def vld_force_checked(mode):
    """Test hook: 1/True = entropy-decode every block with the reader that checks for the end of the scan at every read,
    2 = the default readers without the BMI2 form of the refinement scans, 0/False = default."""
    lib.jpeggpu_internal_vld_force_checked(int(mode))


class Jpeg:
    """One JPEG file between its first SOS and EOI."""

    def __init__(self, data=None, path=None):
        self._h = _vp()
        if path is not None:
            B._check(lib.jpeggpu_jpeg_open_file(str(path).encode(), C.byref(self._h)))
        else:
            self._buf = np.frombuffer(bytes(data), dtype=np.uint8)   # borrowed by the handle: keep alive
            B._check(lib.jpeggpu_jpeg_open_memory(self._buf.ctypes.data, self._buf.size, C.byref(self._h)))

    @classmethod
    def from_file(cls, path):
        return cls(path=path)

    def close(self):
        if self._h:
            lib.jpeggpu_jpeg_close(self._h)
            self._h = _vp()

    __del__ = close

    @property
    def desc(self):
        d = B.FrameDesc()
        B._check(lib.jpeggpu_jpeg_frame_desc(self._h, C.byref(d)))
        return d

    @property
    def scan(self):
        s = ScanInfo()
        B._check(lib.jpeggpu_jpeg_scan_info(self._h, C.byref(s)))
        return s

    @property
    def eoi(self):
        return bool(self.scan.eoi)

    @property
    def comments(self):
        return [lib.jpeggpu_jpeg_comment(self._h, i) for i in range(lib.jpeggpu_jpeg_nb_comments(self._h))]

    def qtables(self):
        out = []
        for i in range(4):
            p = lib.jpeggpu_jpeg_qtable(self._h, i)
            out.append(None if not p else np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(64,)).copy())
        return out

    def plane(self, c):
        """Copy of frame component c's plane: (blocks, 64) int16, quantised, zigzag order."""
        d = self.desc
        n = B.plane_blocks(d, c)
        p = lib.jpeggpu_jpeg_plane(self._h, c)
        if not p or n == 0:
            return np.zeros((0, 64), dtype=np.int16)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(n, 64)).copy()

    def planes(self):
        return [self.plane(c) for c in range(self.desc.nbcomp)]

    def decode_scan(self):
        B._check(lib.jpeggpu_jpeg_decode_scan(self._h))

    def next_header(self):
        B._check(lib.jpeggpu_jpeg_next_header(self._h))

    @property
    def uploaded_bytes(self):
        """Host-to-device bytes queued by the renders of this file so far."""
        return lib.jpeggpu_jpeg_uploaded_bytes(self._h)

    def render(self, out=None):
        """GPU back end over the current planes -> (rows, row_bytes) uint8 raster.  out: a C-contiguous uint8 array of at
        least (rows, row_bytes) to fill instead (its row pitch is used; a PinnedArray receives the download in place)."""
        d = self.desc
        rows, rb = B.raster_rows(d), B.raster_row_bytes(d)
        if out is None:
            out = np.zeros((max(rows, 1), max(rb, 1)), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.ndim == 2 and out.shape[0] >= rows and out.shape[1] >= rb and out.strides[1] == 1
        B._check(lib.jpeggpu_jpeg_render(self._h, out.ctypes.data, out.strides[0]))
        return out[:rows, :rb]


def pnm_header(desc):
    return (b"P5\n" if desc.nbcomp == 1 else b"P6\n") + b"%d %d\n255\n" % (desc.width, desc.height)


def write_pnm(path, desc, raster, rows=None):
    """rows=None: every raster row like the reference; rows=desc.height: payload cropped to the header's height."""
    r = np.ascontiguousarray(raster, dtype=np.uint8)
    stride = r.strides[0] if r.ndim == 2 else 0
    if rows is None:
        B._check(lib.jpeggpu_write_pnm(str(path).encode(), C.byref(desc), r.ctypes.data, stride))
    else:
        B._check(lib.jpeggpu_write_pnm_rows(str(path).encode(), C.byref(desc), r.ctypes.data, stride, rows))


def out_file_name(filepath, outfile, nbcomp, nb):
    buf = C.create_string_buffer(4096)
    n = lib.jpeggpu_out_file_name(str(filepath).encode(), None if outfile is None else str(outfile).encode(), nbcomp, nb, buf, 4096)
    if n < 0:
        raise B.JpegGpuError(n, "out_file_name")
    return buf.value.decode()


def jpeg2ppm(filepath, outfile=None):
    """Whole-file decode like the reference's main(); returns the number of images written."""
    n = C.c_int(0)
    B._check(lib.jpeggpu_jpeg2ppm(str(filepath).encode(), None if outfile is None else str(outfile).encode(), C.byref(n)))
    return n.value


class Timing(C.Structure):
    """jpeggpu_timing_t: host wall-clock seconds of the stages the reference times under -t."""
    _fields_ = [("init_s", C.c_double), ("header_s", C.c_double), ("vld_s", C.c_double), ("gpu_wait_s", C.c_double),
                ("write_s", C.c_double), ("total_s", C.c_double)]


lib.jpeggpu_jpeg2ppm_timed.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(Timing)]
lib.jpeggpu_jpeg2ppm_timed.restype = C.c_int


def jpeg2ppm_timed(filepath, outfile=None):
    """jpeg2ppm that also returns the stage times: (images written, Timing)."""
    n, t = C.c_int(0), Timing()
    B._check(lib.jpeggpu_jpeg2ppm_timed(str(filepath).encode(), None if outfile is None else str(outfile).encode(),
                                        C.byref(n), C.byref(t)))
    return n.value, t


class Blob(C.Structure):
    _fields_ = [("data", _vp), ("len", _sz)]


class BatchStats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("vld_seconds", C.c_double), ("images_ok", C.c_uint32), ("threads", C.c_uint32),
                ("chunk_frames", C.c_uint32), ("chunks", C.c_uint32), ("output_registered", C.c_uint32)]


lib.jpeggpu_decode_batch.argtypes = [C.POINTER(Blob), _sz, _vp, _sz, C.c_int, C.POINTER(C.c_int), C.POINTER(B.FrameDesc),
                                     C.POINTER(BatchStats)]


def decode_batch(files, nthreads=0, out=None):
    """jpeggpu_decode_batch: <files> = sequence of bytes objects holding same-geometry JPEG files.
    -> (rasters (n, rows, row_bytes) uint8, status list, FrameDesc, BatchStats)."""
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    blobs = (Blob * max(n, 1))()
    for i, a in enumerate(bufs):
        blobs[i].data, blobs[i].len = a.ctypes.data, a.size
    # geometry first (host only), to size the output
    probe = None
    for f in files:
        try:
            probe = Jpeg(f)
            break
        except B.JpegGpuError:
            continue
    if probe is None:
        # let the library report the error of the first file
        d, st, stats = B.FrameDesc(), (C.c_int * max(n, 1))(), BatchStats()
        dummy = np.zeros(16, dtype=np.uint8)
        B._check(lib.jpeggpu_decode_batch(blobs, n, dummy.ctypes.data, 0, nthreads, st, C.byref(d), C.byref(stats)))
        return np.zeros((n, 0, 0), np.uint8), list(st)[:n], d, stats
    d0 = probe.desc
    rows, rb = B.raster_rows(d0), B.raster_row_bytes(d0)
    probe.close()
    if out is None:
        out = np.empty((n, rows, rb), dtype=np.uint8)
    d, st, stats = B.FrameDesc(), (C.c_int * max(n, 1))(), BatchStats()
    B._check(lib.jpeggpu_decode_batch(blobs, n, out.ctypes.data, 0, nthreads, st, C.byref(d), C.byref(stats)))
    return out, list(st)[:n], d, stats
