#!/bin/bash
# Cold start of the command-line driver: one process per file, GPU start-up included (wall clock), next to the reference binary.
python3 - <<EOF
import io, subprocess, sys, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_frontend as bf
for name, size, kw in (("c2", (4096, 4096), dict(quality=90, subsampling=0)), ("hd", (1920, 1080), dict(quality=85, subsampling=2))):
    b = io.BytesIO(); bf.picture(size[0], size[1], 5, False).save(b, "JPEG", **kw); open(f"/tmp/{name}.jpg", "wb").write(b.getvalue())
for rep in range(2):
    for name in ("c2", "hd"):
        t0 = time.perf_counter()
        r = subprocess.run(["jpeg-decoder_amd/jpeg2ppm_gpu", "-t", "-o", f"/tmp/out_{name}.ppm", f"/tmp/{name}.jpg"], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        print(f"== jpeg2ppm_gpu -t {name}.jpg (process wall clock {dt * 1e3:.0f} ms, exit {r.returncode})")
        print("\n".join("   " + l for l in r.stdout.splitlines()[-6:]))
for name in ("hd", "c2"):
    t0 = time.perf_counter()
    r = subprocess.run(["oracle/_ref/jpeg2ppm_fast", "-o", f"/tmp/ref_{name}.ppm", f"/tmp/{name}.jpg"], capture_output=True)
    print(f"== reference jpeg2ppm_fast {name}.jpg: process wall clock {(time.perf_counter() - t0) * 1e3:.0f} ms, exit {r.returncode}")
EOF
