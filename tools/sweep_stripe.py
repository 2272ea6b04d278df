#!/usr/bin/env python3
"""tools/sweep_stripe.py -- jpeggpu_backend_run (host pointers) over stripe sizes, pinned and pageable caller memory.
Run on the GPU box:  python tools/sweep_stripe.py [c2|c3|c4|c5] ...   (default c2 c3)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

from jpeg_decoder_amd import backend as jg, synth  # noqa: E402

CONFIGS = {
    "c2": (4096, 4096, [(1, 1), (1, 1), (1, 1)]),
    "c3": (8192, 8192, [(2, 2), (1, 1), (1, 1)]),
    "c4": (4096, 4096, [(2, 1), (1, 1), (1, 1)]),
    "c5": (1920, 1080, [(2, 2), (1, 1), (1, 1)]),
}


def best_ms(f, reps=7):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3


def main():
    names = sys.argv[1:] or ["c2", "c3"]
    jg.init(0)
    qt = synth.shaun_qtables()
    for name in names:
        w, h, samp = CONFIGS[name]
        d = jg.make_desc(w, h, samp, idq=[0, 1, 1][: len(samp)])
        nb = [jg.plane_blocks(d, c) for c in range(d.nbcomp)]
        rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
        nbytes = sum(nb) * 128 + rows * rb
        planes = [synth.synth_blocks(nb[c], 0x57 + c) for c in range(d.nbcomp)]
        out = np.empty((rows, rb), dtype=np.uint8)
        held = [jg.PinnedArray(p.shape, p.dtype) for p in planes] + [jg.PinnedArray(out.shape, out.dtype)]
        for hp, p in zip(held, planes):
            hp.array[...] = p
        pp, po = [hp.array for hp in held[:-1]], held[-1].array
        print(f"{name}: {w}x{h}, {nbytes / 1e6:.1f} MB in+out", flush=True)
        for mib in (1, 2, 4, 6, 8, 12, 16, 24, 32, 64):
            os.environ["JPEGGPU_STRIPE_BYTES"] = str(mib << 20)
            jg.reload_tuning()
            pm, pmed = best_ms(lambda: jg.backend_run(d, pp, qt, out=po))
            gm, gmed = best_ms(lambda: jg.backend_run(d, planes, qt, out=out))
            assert np.array_equal(po, out)
            print(f"  stripe {mib:3d} MiB: pinned {pm:7.3f} ms (median {pmed:7.3f}) {nbytes / pm / 1e6:6.1f} GB/s | "
                  f"pageable {gm:7.3f} ms (median {gmed:7.3f}) {nbytes / gm / 1e6:6.1f} GB/s", flush=True)
        for hp in held:
            hp.free()
    jg.shutdown()


if __name__ == "__main__":
    main()
