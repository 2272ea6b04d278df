#!/usr/bin/env python3
"""tools/bench_jpeg2ppm_stages.py -- jpeggpu_jpeg2ppm_timed's stage times for the C2 / C4 files (second run of each: buffers reused)."""
import io
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import jpeg_decoder_amd as jg  # noqa: E402
from jpeg_decoder_amd import frontend as F  # noqa: E402
import bench_frontend as bf  # noqa: E402


def main():
    jg.init(0)
    tmp = tempfile.mkdtemp(prefix="j2p")
    for name, kw in (("c2_4096_444_base", dict(quality=90, subsampling=0)), ("c4_4096_422_prog", dict(quality=90, subsampling=1, progressive=True))):
        b = io.BytesIO()
        bf.picture(4096, 4096, 5, False).save(b, "JPEG", **kw)
        path = os.path.join(tmp, name + ".jpg")
        open(path, "wb").write(b.getvalue())
        for rep in range(3):
            n, t = F.jpeg2ppm_timed(path, os.path.join(tmp, "o", name + ".ppm"))
        print(name, n, "images:", {f: round(getattr(t, f) * 1e3, 2) for f, _ in t._fields_}, flush=True)


if __name__ == "__main__":
    main()
