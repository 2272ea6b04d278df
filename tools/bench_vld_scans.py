#!/usr/bin/env python3
"""tools/bench_vld_scans.py -- host entropy decoding of a 4096x4096 4:2:2 progressive file, scan by scan (best of N passes)."""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from jpeg_decoder_amd import frontend as F  # noqa: E402
import bench_frontend as bf  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    b = io.BytesIO()
    bf.picture(4096, 4096, 5, False).save(b, "JPEG", quality=90, subsampling=1, progressive=True)
    data = b.getvalue()
    best = {}
    for _ in range(reps):
        j = F.Jpeg(data)
        k = 0
        while not j.eoi:
            t0 = time.perf_counter()
            j.decode_scan()
            dt = time.perf_counter() - t0
            s = j.scan
            key = (k, tuple(s.scan_comp[:s.nb_scan_comp]), s.ss, s.se, s.ah, s.al)
            best[key] = min(best.get(key, 1e9), dt)
            k += 1
            j.next_header()
        j.close()
    for key, v in best.items():
        print(f"scan {key[0]}: components {key[1]} Ss {key[2]} Se {key[3]} Ah {key[4]} Al {key[5]}: {v * 1e3:7.2f} ms", flush=True)
    print(f"total {sum(best.values()) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
