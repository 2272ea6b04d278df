#!/usr/bin/env python3
"""tools/bench_vld.py -- host entropy decoding alone (header parser + VLD into full-frame planes), no GPU involved:
best of N passes over files encoded on the spot, and a hash of the planes so that two builds can be compared."""
import hashlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from jpeg_decoder_amd import frontend as F  # noqa: E402
import bench_frontend as bf  # noqa: E402

CASES = [("c2_4096_444_base", (4096, 4096), dict(quality=90, subsampling=0), False),
         ("1080p_420_base", (1920, 1080), dict(quality=85, subsampling=2), False),
         ("c4_4096_422_prog", (4096, 4096), dict(quality=90, subsampling=1, progressive=True), False),
         ("1080p_420_prog", (1920, 1080), dict(quality=85, subsampling=2, progressive=True), False),
         ("c1_512_gray_base", (512, 512), dict(quality=90), True)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for name, (w, h), kw, gray in CASES:
        b = io.BytesIO()
        bf.picture(w, h, 5, gray).save(b, "JPEG", **kw)
        data = b.getvalue()
        best, hs = 1e9, ""
        for _ in range(reps):
            j = F.Jpeg(data)
            t = 0.0
            while not j.eoi:
                t0 = time.perf_counter()
                j.decode_scan()
                t += time.perf_counter() - t0
                j.next_header()
            best = min(best, t)
            hs = hashlib.sha256(b"".join(j.plane(c).tobytes() for c in range(j.desc.nbcomp))).hexdigest()[:16]
            j.close()
        print(f"{name}: vld {best * 1e3:8.2f} ms  {len(data) / best / 1e6:6.1f} MB/s  {w * h / best / 1e6:7.1f} MPix/s  planes {hs}", flush=True)


if __name__ == "__main__":
    main()
