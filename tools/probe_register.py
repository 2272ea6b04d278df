"""Probe: which (alignment, size) combinations hipHostRegister accepts on this box, and what it costs."""
import ctypes as C, time, mmap, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import torch  # noqa
import numpy as np
import jpeg_decoder_amd as jg
jg.init(0)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipGetErrorString.restype = C.c_char_p
libc = C.CDLL(None)
libc.posix_memalign.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]
def try_reg(ptr, n, tag):
    t0 = time.perf_counter(); r = hip.hipHostRegister(ptr, n, 0); t1 = time.perf_counter()
    msg = hip.hipGetErrorString(r).decode()
    u = None
    if r == 0:
        t2 = time.perf_counter(); u = hip.hipHostUnregister(ptr); t3 = time.perf_counter()
        print(f"{tag}: ok register {1e3*(t1-t0):.2f} ms unregister {1e3*(t3-t2):.2f} ms rc={u}")
    else:
        print(f"{tag}: FAILED {r} {msg}")
for size in (4096, 100000, 1 << 20, 150 << 20):
    p = C.c_void_p()
    libc.posix_memalign(C.byref(p), 4096, size + 4096)
    C.memset(p, 1, size)
    try_reg(p.value, size, f"posix_memalign aligned size={size}")
    try_reg(p.value + 64, size, f"posix_memalign +64 size={size}")
a = np.ones(1 << 20, dtype=np.uint8)
try_reg(a.ctypes.data, a.nbytes, f"numpy 1MiB ptr%4096={a.ctypes.data % 4096}")
m = mmap.mmap(-1, 1 << 20)
b = np.frombuffer(m, dtype=np.uint8); b[:] = 1
try_reg(b.ctypes.data, b.nbytes, "mmap 1MiB")
raw = np.empty(100000 + 8192, dtype=np.uint8); o = (-raw.ctypes.data) % 4096; raw[:] = 1
try_reg(raw.ctypes.data + o, 100000, "numpy interior aligned 100000")
