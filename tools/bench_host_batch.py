#!/usr/bin/env python3
"""tools/bench_host_batch.py -- N frames of 1920x1080 4:2:0 (BASELINE.json configs[4]'s frame) in HOST memory through the
back end: a loop of jpeggpu_backend_run calls against one jpeggpu_backend_run_batch call (one pipeline across frames),
pinned and pageable caller memory.  PCIe-inclusive: never the headline value.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from jpeg_decoder_amd import backend as jg, synth  # noqa: E402


def best(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w, h, samp = 1920, 1080, [(2, 2), (1, 1), (1, 1)]
    jg.init(0)
    qt = synth.shaun_qtables()
    d = jg.make_desc(w, h, samp, idq=[0, 1, 1])
    nb = [jg.plane_blocks(d, c) for c in range(3)]
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    nbytes = nf * (sum(nb) * 128 + rows * rb)
    planes = [synth.synth_blocks(nb[c] * nf, 0xB0 + c).reshape(nf, nb[c], 64) for c in range(3)]
    out = np.zeros((nf, rows, rb), dtype=np.uint8)
    held = [jg.PinnedArray(p.shape, p.dtype) for p in planes] + [jg.PinnedArray(out.shape, out.dtype)]
    for hp, p in zip(held, planes):
        hp.array[...] = p
    pp, po = [hp.array for hp in held[:-1]], held[-1].array

    def loop(pl, o):
        for f in range(nf):
            jg.backend_run(d, [pl[c][f] for c in range(3)], qt, out=o[f])

    for name, pl, o in (("pinned", pp, po), ("pageable", planes, out)):
        t_loop = best(lambda: loop(pl, o))
        ref = o.copy()
        o[...] = 0
        t_batch = best(lambda: jg.backend_run_batch(d, nf, pl, qt, out=o))
        assert np.array_equal(o, ref)
        print(f"{name:9s} {nf} frames, {nbytes / 1e6:.0f} MB in+out: loop of backend_run {t_loop * 1e3:8.2f} ms = {nf / t_loop:7.0f} frames/s "
              f"{nbytes / t_loop / 1e9:5.1f} GB/s | backend_run_batch {t_batch * 1e3:8.2f} ms = {nf / t_batch:7.0f} frames/s {nbytes / t_batch / 1e9:5.1f} GB/s",
              flush=True)
    for hp in held:
        hp.free()
    jg.shutdown()


if __name__ == "__main__":
    main()
