#!/usr/bin/env python3
"""End-to-end timing of real files (SURVEY 8d configs C1/C2/C4 as files): host entropy decode, GPU
render (upload + kernel + download), whole jpeggpu_jpeg2ppm, and the reference's jpeg2ppm_fast binary
on the same file when oracle/_ref travelled.  Not the headline metric (that is bench.py's device-resident
back end); these numbers go to DESIGN.md as the PCIe- and VLD-inclusive picture."""
import io, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import jpeg_decoder_amd as jg
from jpeg_decoder_amd import frontend as F
from PIL import Image

def picture(w, h, seed, gray=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x + y) % 256, (x * 3 + y // 2) % 256, (x // 3 + y * 2) % 256], -1).astype(np.float32)
    a += 50 * np.sin(x / 23.0)[..., None] * np.cos(y / 17.0)[..., None]
    a += rng.normal(0, 10, a.shape).astype(np.float32)
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    return im.convert("L") if gray else im

def main():
    jg.init(0)
    tmp = tempfile.mkdtemp(prefix="fe")
    cases = [("c1_512_gray_base", (512, 512), dict(quality=90), True),
             ("c2_4096_444_base", (4096, 4096), dict(quality=90, subsampling=0), False),
             ("c4_4096_422_prog", (4096, 4096), dict(quality=90, subsampling=1, progressive=True), False),
             ("1080p_420_base", (1920, 1080), dict(quality=85, subsampling=2), False)]
    ref = os.path.join(ROOT, "oracle", "_ref", "jpeg2ppm_fast")
    for name, (w, h), kw, gray in cases:
        b = io.BytesIO(); picture(w, h, 5, gray).save(b, "JPEG", **kw); data = b.getvalue()
        path = os.path.join(tmp, name + ".jpg"); open(path, "wb").write(data)
        res = {"case": name, "file_bytes": len(data), "mpix": w * h / 1e6}
        for rep in range(2):   # second pass = warm
            t0 = time.perf_counter(); j = F.Jpeg(data); t_open = time.perf_counter() - t0
            t_dec = t_ren = 0.0; n = 0
            while not j.eoi:
                t0 = time.perf_counter(); j.decode_scan(); t_dec += time.perf_counter() - t0
                t0 = time.perf_counter(); j.render(); t_ren += time.perf_counter() - t0
                n += 1
                if j.eoi: break
                j.next_header()
            j.close()
        res.update(scans=n, open_ms=round(t_open * 1e3, 2), vld_ms=round(t_dec * 1e3, 2), render_ms=round(t_ren * 1e3, 2),
                   vld_MBps=round(len(data) / t_dec / 1e6, 1), vld_MPixps=round(w * h * n / t_dec / 1e6, 1))
        out = os.path.join(tmp, "o_" + name, "x.pnm")
        t0 = time.perf_counter(); F.jpeg2ppm(path, out); res["jpeg2ppm_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        if os.path.exists(ref):
            out2 = os.path.join(tmp, "r_" + name, "x.pnm")
            t0 = time.perf_counter(); subprocess.run([ref, "-o", out2, path], check=True); res["reference_fast_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            same = all(open(os.path.join(tmp, "o_" + name, f), "rb").read() == open(os.path.join(tmp, "r_" + name, f), "rb").read()
                       for f in os.listdir(os.path.join(tmp, "r_" + name)) if not os.path.islink(os.path.join(tmp, "r_" + name, f)))
            res["identical_to_reference_fast"] = same   # -Ofast build: may differ by 1 LSB in rare samples
        print(json.dumps(res), flush=True)

if __name__ == "__main__":
    main()
