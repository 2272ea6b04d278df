#!/usr/bin/env python3
"""Do consecutive batches on alternating streams overlap one launch's drain with the next one's ramp-up?
Wall-clock throughput of K launches of F frames on 1 stream vs round-robin over 2 and 3 streams."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
import torch
import jpeg_decoder_amd as jg
from jpeg_decoder_amd import synth

def main():
    jg.init(0)
    w = h = 4096
    samp = ((1, 1),) * 3
    d = jg.make_desc(w, h, samp)
    qt = synth.shaun_qtables()
    nb = [jg.plane_blocks(d, c) for c in range(3)]
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    for F in (2, 4, 8):
        NBUF = 3
        bufs = []
        for b in range(NBUF):
            planes = [torch.from_numpy(np.concatenate([synth.synth_blocks(nb[c], 0x5EED0000 + c, (b * F + f) * nb[c], 0) for f in range(F)])).cuda() for c in range(3)]
            out = torch.empty(F * rows * rb, dtype=torch.uint8, device="cuda")
            bufs.append((planes, out))
        for ns in (1, 2, 3):
            streams = [torch.cuda.Stream() for _ in range(ns)]
            def run(K):
                for k in range(K):
                    planes, out = bufs[k % NBUF]
                    jg.backend_run_batch_device(d, F, [t.data_ptr() for t in planes], None, qt, out.data_ptr(), 0, 0, streams[k % ns].cuda_stream)
            run(300); torch.cuda.synchronize()
            K = 600
            t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            byts = K * F * (sum(nb) * 128 + rows * rb)
            print(f"frames/launch={F} streams={ns}: {dt / K * 1e6:.1f} us per launch, {byts / dt / 1e9:.0f} GB/s = {byts / dt / 8e12 * 100:.1f} % of peak", flush=True)

main()
