#!/usr/bin/env python3
"""bench.py -- MPix/s of the fused IQZZ + IDCT + upsample + YCbCr->RGB back end on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5] [--frames F]

A step = one pass of the hot path over one batch of F same-geometry frames (device-resident,
distinct buffers, > 256 MiB in total so that nothing is served from the Infinity Cache), issued as
ONE launch of jpeggpu_backend_run_batch_device through the C ABI.  For N > 1 every rank (one
process per GPU, launched by torch.distributed.run) decodes its own batch of independent images --
the path shards with no exchange step, so there is no data-path collective; RCCL is used only for
the barrier and the max-over-ranks of the timing ("scaling": "weak").

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (description, width, height, sampling, frames per step)
    "c2": ("4096x4096 4:4:4 baseline colour, synthetic coefficients (BASELINE.json configs[1])", 4096, 4096, ((1, 1),) * 3, 16),
    "c3": ("8192x8192 4:2:0, 2x2 chroma upsampling fused into colour convert (configs[2])", 8192, 8192, ((2, 2), (1, 1), (1, 1)), 8),
    "c4": ("4096x4096 4:2:2, GPU part of the progressive config (configs[3])", 4096, 4096, ((2, 1), (1, 1), (1, 1)), 16),
    "g1": ("8192x8192 grayscale (save_mcu_ligne_bw path; not a BASELINE config, gray coverage only)", 8192, 8192, ((1, 1),), 8),
    "c5": ("batch of 1920x1080 4:2:0 images (configs[4], per-GPU slice)", 1920, 1080, ((2, 2), (1, 1), (1, 1)), 256),
}


def algorithmic_bytes(jg, d):
    """sum_c blocks_c*128 (int16 coefficients in) + width*padded_rows*{3|1} (pixels out)."""
    return sum(jg.plane_blocks(d, c) * 128 for c in range(d.nbcomp)) + jg.raster_rows(d) * jg.raster_row_bytes(d)


def cpu_baseline(name, w, h, samp, qt, budget_s=12.0):
    """The reference's own back end (oracle/_ref, -Ofast objects = jpeg2ppm_fast's) timed on one host
    core over whole frames of the same workload until ~budget_s of CPU time is spent."""
    import numpy as np
    from oracle import pyoracle
    from jpeg_decoder_amd import synth
    be, kind = pyoracle.load_ref("fast"), "reference"
    if be is None:
        be, kind = pyoracle.load_fast_oracle(), "port"
    if be is None:
        return None
    d = pyoracle.FrameDesc()
    d.width, d.height, d.nbcomp, d.apply_iquant = w, h, len(samp), 1
    for c, (hs, vs) in enumerate(samp):
        d.comp[c].hs, d.comp[c].vs, d.comp[c].idq = hs, vs, (0 if c == 0 else 1)
    _, d = pyoracle.frame_geometry(d)
    # bounded sample: a stripe of MCU rows sized for ~budget_s at ~25 MPix/s
    rows_px = d.max_v * 8
    want_pix = 25e6 * budget_s
    nrows = int(max(1, min(d.nbmcuV, want_pix / (w * rows_px))))
    planes = [synth.synth_blocks(nrows * d.comp[c].vs * d.nbmcuH * d.comp[c].hs, 0xBA5E + c) for c in range(d.nbcomp)]
    sd = pyoracle.FrameDesc.from_buffer_copy(bytes(d))
    sd.nbmcuV = nrows
    sd.height = min(h, nrows * rows_px)
    t0 = time.perf_counter()
    reps = 0
    while True:
        be.backend_run(sd, planes, qt)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s * 0.85 or reps >= 64:
            break
    pix = reps * w * min(h, nrows * rows_px)
    return {"value": round(pix / dt / 1e6, 2), "unit": "MPix/s", "cores": 1, "kind": kind,
            "sample": f"{reps} x {nrows} MCU rows ({w}x{nrows * rows_px} px) of workload {name}, "
                      f"{'reference -Ofast objects (oracle/_ref/libjpegref_fast.so)' if kind == 'reference' else 'oracle/liboracle_fast.so'}, "
                      f"{dt:.1f} s on 1 core"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per step (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-ms", type=float, default=1500.0,
                    help="untimed launches before the W warmup steps until the chip's clock governor has settled "
                         "(launch time drifts 138 -> 178 -> 127 us over the first ~25 ms of load on MI355X)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import jpeg_decoder_amd as jg
    from jpeg_decoder_amd import synth
    jg.init(local_rank)

    desc_text, w, h, samp, nf_default = WORKLOADS[args.workload]
    nf = args.frames or nf_default
    d = jg.make_desc(w, h, samp)
    qt = synth.shaun_qtables()
    nb = [jg.plane_blocks(d, c) for c in range(d.nbcomp)]
    # independent images per rank: block index stream offset by (rank*nf + frame)
    planes = []
    for c in range(d.nbcomp):
        host = np.empty((nf * nb[c], 64), dtype=np.int16)
        for f in range(nf):
            host[f * nb[c]:(f + 1) * nb[c]] = synth.synth_blocks(nb[c], 0x5EED0000 + c, (rank * nf + f) * nb[c], 0)
        planes.append(torch.from_numpy(host).cuda())
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    out = torch.empty(nf * rows * rb, dtype=torch.uint8, device="cuda")
    ptrs = [t.data_ptr() for t in planes]
    stream = torch.cuda.Stream()      # a real (non-NULL) HIP stream: kernels AND events go on it

    def step():
        jg.backend_run_batch_device(d, nf, ptrs, None, qt, out.data_ptr(), 0, 0, stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    fence()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    tt = torch.tensor([t_local], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_max = float(tt.item())

    pix_per_step = nf * w * h                       # nominal pixels, per rank
    value = world * pix_per_step * args.steps / t_max / 1e6
    launch_s = float(np.mean(kernel_ms)) / 1e3
    bytes_per_launch = nf * algorithmic_bytes(jg, d)
    achieved = bytes_per_launch / launch_s / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-measured HBM bytes per launch, if collected
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(f"{args.workload}_f{nf}")
        except Exception:
            traffic = None
    res = {
        "metric": "MPix/s IQZZ+IDCT+upsample+YCbCr->RGB", "value": round(value, 1), "unit": "MPix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+i32 (u8 out)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc_text}", "frames_per_step_per_gpu": nf, "width": w, "height": h,
                   "sampling": [list(s) for s in samp], "blocks_per_frame": int(sum(nb)), "sharding": "independent images per rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "kernel": "backend_kernel", "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(launch_s * 1e6, 2),
                     "min_launch_us": round(min(kernel_ms) * 1e3, 2)},
    }
    if os.environ.get("JPEGGPU_BENCH_TRACE") and rank == 0:
        print("launch_us:", " ".join(f"{x * 1e3:.0f}" for x in kernel_ms), file=sys.stderr)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.workload, w, h, samp, qt)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
