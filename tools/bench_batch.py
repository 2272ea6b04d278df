#!/usr/bin/env python3
"""End-to-end rate of jpeggpu_decode_batch on real 1920x1080 4:2:0 files (BASELINE.json configs[4] on one
GPU): host threads for the Huffman decoding + batched launches.  Prints one JSON line per thread count and
the reference's jpeg2ppm_fast time per file for scale.  PCIe and VLD inclusive: never the headline value."""
import io, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import jpeg_decoder_amd as jg
from jpeg_decoder_amd import frontend as F
from PIL import Image

def picture(i, w=1920, h=1080):
    rng = np.random.default_rng(i)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x * (i % 5 + 1) + y) % 256, (x + y * (i % 3 + 2)) % 256, (x // 2 + y * 3 + 7 * i) % 256], -1).astype(np.float32)
    a += 40 * np.sin(x / (11.0 + i))[..., None] + rng.normal(0, 8, a.shape).astype(np.float32)
    b = io.BytesIO(); Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=85, subsampling=2)
    return b.getvalue()

def main():
    jg.init(0)
    base = [picture(i) for i in range(32)]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    files = [base[i % 32] for i in range(n)]
    mb = sum(len(f) for f in files) / 1e6
    out = None
    pinned = None
    for threads, pin in ((1, False), (4, False), (8, False), (16, False), (16, True), (32, False)):
        o, st, d, s = F.decode_batch(files, threads, out)
        out = o
        dst = out
        if pin:     # the caller's raster array in pinned memory (jpeggpu_host_alloc): downloads land there, no staging copy
            if pinned is None:
                pinned = jg.PinnedArray(out.shape, out.dtype)
            dst = pinned.array
            F.decode_batch(files, threads, dst)
        o, st, d, s = F.decode_batch(files, threads, dst)      # second pass: pages of the output already touched, chunk buffers reused
        assert all(x == 0 for x in st)
        if pin:
            assert np.array_equal(dst, out)
        print(json.dumps({"threads": s.threads, "images": n, "file_MB": round(mb, 1), "seconds": round(s.seconds, 3),
                          "images_per_s": round(n / s.seconds, 1), "MPix_per_s": round(n * 1920 * 1080 / s.seconds / 1e6, 1),
                          "vld_ms_per_image": round(s.vld_seconds / n * 1e3, 2), "chunk_frames": s.chunk_frames, "chunks": s.chunks,
                          "output_pinned": bool(pin), "output_registered": s.output_registered}), flush=True)
    if pinned is not None:
        pinned.free()
    ref = os.path.join(ROOT, "oracle", "_ref", "jpeg2ppm_fast")
    if os.path.exists(ref):
        tmp = tempfile.mkdtemp(); p = os.path.join(tmp, "a.jpg"); open(p, "wb").write(base[0])
        t0 = time.perf_counter()
        for _ in range(5): subprocess.run([ref, "-o", os.path.join(tmp, "o", "x.ppm"), p], check=True)
        print(json.dumps({"reference_jpeg2ppm_fast_ms_per_image_1core": round((time.perf_counter() - t0) / 5 * 1e3, 1)}))

if __name__ == "__main__":
    main()
