#!/usr/bin/env python3
"""Does a lone 4096x4096 4:4:4 frame finish sooner as two half-frames (MCU-row stripes) on two streams?
Latency per frame (synchronise after every frame) and back-to-back rate, 1 launch vs 2 half launches vs 4 quarter launches."""
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
    NBUF = 4     # 4 x 151 MB > the 256 MB Infinity Cache
    bufs = []
    for b in range(NBUF):
        planes = [torch.from_numpy(synth.synth_blocks(nb[c], 0x5EED0000 + c, b * nb[c], 0)).cuda() for c in range(3)]
        bufs.append((planes, torch.empty(rows * rb, dtype=torch.uint8, device="cuda")))
    byts = sum(nb) * 128 + rows * rb
    for parts in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        main_s = streams[0]
        evs = [torch.cuda.Event() for _ in range(parts)]
        fork = torch.cuda.Event()
        sd = jg.FrameDesc.from_buffer_copy(bytes(d))
        sd.nbmcuV = d.nbmcuV // parts
        per_row = [d.nbmcuH * d.comp[c].hs * d.comp[c].vs * 128 for c in range(3)]
        def frame(k):
            planes, out = bufs[k % NBUF]
            if parts > 1:
                fork.record(main_s)
            for i in range(parts):
                s = streams[i]
                if i:
                    s.wait_event(fork)
                m0 = i * sd.nbmcuV
                jg.backend_run_device(sd, [planes[c].data_ptr() + m0 * per_row[c] for c in range(3)], qt,
                                      out.data_ptr() + m0 * 8 * rb, 0, s.cuda_stream)
                if i:
                    evs[i].record(s)
            for i in range(1, parts):
                main_s.wait_event(evs[i])
        for mode in ("latency", "back-to-back"):
            for k in range(200):
                frame(k)
            torch.cuda.synchronize()
            K = 400
            t0 = time.perf_counter()
            for k in range(K):
                frame(k)
                if mode == "latency":
                    main_s.synchronize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K
            print(f"parts={parts} {mode:13s}: {dt * 1e6:7.1f} us per frame, {byts / dt / 8e12 * 100:5.1f} % of peak", flush=True)

main()
