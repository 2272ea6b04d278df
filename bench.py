#!/usr/bin/env python3
"""bench.py -- MPix/s of the fused IQZZ + IDCT + upsample + YCbCr->RGB back end on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5|g1] [--frames F]
                  [--scaling weak|strong] [--images M]

A step = one pass of the hot path over one batch of same-geometry frames (device-resident, distinct
buffers, > 256 MiB in total so that nothing is served from the Infinity Cache), issued through the C ABI
(jpeggpu_backend_run_batch_device).

N > 1: one process per GPU.  When the script is started WITHOUT a launcher (no WORLD_SIZE in the environment) it
starts N ranks itself -- python -m torch.distributed.run, as a child process, before anything touches a GPU -- and
exits with the launcher's status; under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  It refuses to run when
fewer than N devices are visible: the line it prints always has n_gpus == --gpus.

The path shards with no exchange step, so there is no data-path collective; the process group is used
for the barrier, the max-over-ranks of the time and gathering per-rank launch times.
  --scaling weak   (default) every rank decodes its own batch of F frames per step            -> "scaling": "weak"
  --scaling strong the job is fixed and cut across the ranks (jpeg_decoder_amd/sharding.py)   -> "scaling": "strong"
       c5: --images M (default 8192, BASELINE.json configs[4]) independent 1080p images, image i on rank i mod N;
           a step = the whole job, each rank working through its images in launches of F frames
       c2/c3/c4/g1: the F frames of a step are cut into MCU-row stripes, rank g gets rows [g*nbmcuV/N, (g+1)*nbmcuV/N)

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import json
import os
import subprocess
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
    "c5": ("batch of 1920x1080 4:2:0 images (configs[4])", 1920, 1080, ((2, 2), (1, 1), (1, 1)), 256),
}


def algorithmic_bytes(jg, d):
    """sum_c blocks_c*128 (int16 coefficients in) + width*padded_rows*{3|1} (pixels out)."""
    return sum(jg.plane_blocks(d, c) * 128 for c in range(d.nbcomp)) + jg.raster_rows(d) * jg.raster_row_bytes(d)


def host_cores():
    """(cores this process may run on, cores of the machine)."""
    try:
        mine = len(os.sched_getaffinity(0))
    except AttributeError:
        mine = os.cpu_count() or 1
    return mine, os.cpu_count() or mine


def bench_jpeg(w, h, samp):
    """A real baseline JPEG of the workload's geometry (q90, at most ~8 MPix so that the reference stays within seconds),
    encoded on the spot with Pillow -> (bytes, height) or (None, 0)."""
    try:
        import io
        import numpy as np
        from PIL import Image
    except ImportError:
        return None, 0
    if len(samp) not in (1, 3):
        return None, 0
    sub = {((1, 1),) * 3: 0, ((2, 1), (1, 1), (1, 1)): 1, ((2, 2), (1, 1), (1, 1)): 2}.get(tuple(samp))
    if len(samp) == 3 and sub is None:
        return None, 0
    hh = min(h, max(16, (8 << 20) // w // 16 * 16))
    y, x = np.mgrid[0:hh, 0:w]
    a = np.stack([(x + y) % 256, (x * 3 + y // 2) % 256, (x // 3 + y * 2) % 256], -1).astype(np.float64)
    a += np.random.default_rng(5).normal(0, 12, a.shape)
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    kw = {"quality": 90}
    if len(samp) == 1:
        im = im.convert("L")
    else:
        kw["subsampling"] = sub
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue(), hh


def whole_file_times(w, h, samp):
    """jpeggpu_jpeg2ppm_timed (host entropy decoding + GPU back end + file writing, overlapped) on the same JPEG that
    cpu_baseline.jpeg2ppm_fast_t hands to the reference binary; second run (buffers of the first one reused)."""
    import tempfile
    from jpeg_decoder_amd import frontend as F
    data, hh = bench_jpeg(w, h, samp)
    if data is None:
        return None
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "in.jpg")
        with open(src, "wb") as f:
            f.write(data)
        t = None
        for _ in range(2):
            _, t = F.jpeg2ppm_timed(src, os.path.join(tmp, "o", "out.pnm"))
        return {"entry": "jpeggpu_jpeg2ppm_timed", "picture": f"{w}x{hh} q90 JPEG ({len(data)} bytes), the one of cpu_baseline.jpeg2ppm_fast_t",
                "header_s": round(t.header_s, 6), "vld_s": round(t.vld_s, 6), "gpu_wait_s": round(t.gpu_wait_s, 6),
                "write_wait_s": round(t.write_s, 6), "total_s": round(t.total_s, 6), "MPix_s": round(w * hh / t.total_s / 1e6, 1)}


def reference_stage_times(name, w, h, samp):
    """jpeg2ppm_fast -t (the reference's own -Ofast binary and its own timers, src/baseline.c:224-228) on a real JPEG
    of the workload's geometry encoded on the spot.  None when the binary (oracle/_ref, built where /root/reference
    exists) or Pillow is missing."""
    exe = os.path.join(ROOT, "oracle", "_ref", "jpeg2ppm_fast")
    if not os.path.exists(exe):
        return None
    import tempfile
    data, hh = bench_jpeg(w, h, samp)
    if data is None:
        return None
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "in.jpg")
        with open(src, "wb") as f:
            f.write(data)
        try:
            r = subprocess.run([exe, "-t", "-o", os.path.join(tmp, "out.pnm"), src], capture_output=True, timeout=120)
        except (subprocess.TimeoutExpired, OSError):
            return None
        if r.returncode != 0:
            return None
        t = {}
        for line in r.stdout.decode("utf-8", "replace").splitlines():
            if " : " in line and line.endswith(" s"):
                k, v = line.rsplit(" : ", 1)
                t[k.strip()] = float(v[:-2])
        izz, idct, wr = t.get("IZZ"), t.get("IDCT"), t.get("Ecriture de l'image")
        if None in (izz, idct, wr):
            return None
        back = izz + idct + wr
        return {"binary": "oracle/_ref/jpeg2ppm_fast -t", "picture": f"{w}x{hh} q90 JPEG encoded on the spot ({os.path.getsize(src)} bytes)",
                "izz_s": izz, "idct_s": idct, "write_s": wr, "vld_iquant_s": t.get("Décodage DC/AC et Quantification"),
                "total_s": t.get("Décodage complet de l'image"), "backend_mpix_s": round(w * hh / back / 1e6, 2) if back > 0 else None,
                "note": "back end = IZZ + IDCT + Ecriture timers (two gettimeofday per block and stage included, as the reference measures itself)"}


def cpu_baseline(name, w, h, samp, qt, budget_s=12.0):
    """The reference's own back end (oracle/_ref, -Ofast objects = jpeg2ppm_fast's) timed on one host
    core over whole frames of the same workload until ~budget_s of CPU time is spent."""
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
    mine, machine = host_cores()
    v = pix / dt / 1e6
    return {"value": round(v, 2), "unit": "MPix/s", "cores": 1, "kind": kind,
            "sample": f"{reps} x {nrows} MCU rows ({w}x{nrows * rows_px} px) of workload {name}, "
                      f"{'reference -Ofast objects (oracle/_ref/libjpegref_fast.so) driven by oracle/ref_driver.c' if kind == 'reference' else 'oracle/liboracle_fast.so'}, "
                      f"{dt:.1f} s on 1 core",
            "host_cores": machine, "host_cores_usable": mine,
            "n_process": {"value": round(v * mine, 1), "processes": mine,
                          "note": "EXTRAPOLATION, not measured: the reference is single-threaded (no threads anywhere in src/); "
                                  "N independent processes on independent images / MCU-row stripes = N x the one-core figure"},
            "jpeg2ppm_fast_t": reference_stage_times(name, w, h, samp)}


def end_to_end(jg, np, d, qt, w, h, reps=3):
    """jpeggpu_backend_run (host pointers: H2D + kernel + D2H, PCIe-bound) on ONE frame of the workload, caller memory
    pageable (staged through pinned stripes) and pinned (jpeggpu_host_alloc: DMA in place).  Never the headline value."""
    from jpeg_decoder_amd import synth
    nb = [jg.plane_blocks(d, c) for c in range(d.nbcomp)]
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    nbytes = sum(nb) * 128 + rows * rb
    planes = [synth.synth_blocks(nb[c], 0xE2E + c) for c in range(d.nbcomp)]
    res = {"frame_bytes_in_plus_out": int(nbytes), "pcie_spec_GBps_per_direction": 63.0,
           "link_measured": "57 GB/s up, 57 GB/s down, 85 GB/s both at once (profiles/r02_probe_pcie.txt); GBps below = (in + out) / time"}
    out = np.empty((rows, rb), dtype=np.uint8)
    best = None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        jg.backend_run(d, planes, qt, out=out)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    res["pageable"] = {"ms": round(best * 1e3, 3), "MPix_s": round(w * h / best / 1e6, 1), "GBps": round(nbytes / best / 1e9, 2)}
    held = []
    try:
        held = [jg.PinnedArray(p.shape, p.dtype) for p in planes] + [jg.PinnedArray(out.shape, out.dtype)]
        for hp, p in zip(held, planes):
            hp.array[...] = p
        pp, po = [hp.array for hp in held[:-1]], held[-1].array
        best = None
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            jg.backend_run(d, pp, qt, out=po)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["pinned"] = {"ms": round(best * 1e3, 3), "MPix_s": round(w * h / best / 1e6, 1), "GBps": round(nbytes / best / 1e9, 2)}
        assert np.array_equal(po, out)
        del pp, po
    finally:
        for hp in held:
            hp.free()
    # BASELINE.json configs[4]'s frame from HOST memory: 32 frames of 1920x1080 4:2:0 through one jpeggpu_backend_run_batch call
    # (one pipeline across the frames), pinned memory
    held = []
    try:
        nf = 32
        d5 = jg.make_desc(1920, 1080, [(2, 2), (1, 1), (1, 1)], idq=[0, 1, 1])
        nb5 = [jg.plane_blocks(d5, c) for c in range(3)]
        rows5, rb5 = jg.raster_rows(d5), jg.raster_row_bytes(d5)
        held = [jg.PinnedArray((nf, nb5[c], 64), np.int16) for c in range(3)] + [jg.PinnedArray((nf, rows5, rb5), np.uint8)]
        for c in range(3):
            held[c].array[...] = synth.synth_blocks(nb5[c] * nf, 0xB5 + c).reshape(nf, nb5[c], 64)
        pl, o = [hp.array for hp in held[:3]], held[3].array
        best = None
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            jg.backend_run_batch(d5, nf, pl, qt, out=o)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        b5 = nf * (sum(nb5) * 128 + rows5 * rb5)
        res["batch_1080p_420_pinned"] = {"frames": nf, "ms": round(best * 1e3, 3), "frames_per_s": round(nf / best, 1),
                                         "MPix_s": round(nf * 1920 * 1080 / best / 1e6, 1), "GBps": round(b5 / best / 1e9, 2)}
        del pl, o
    finally:
        for hp in held:
            hp.free()
    return res


def measure(jg, workload, frames, steps, warmup, prewarm_ms, strong, images, rank, world, dry_run, host_collectives=False, verify=False):
    """W warm-up steps, then exactly `steps` timed steps of `workload` (barrier + synchronize on both sides, HIP events around
    every launch on the launch stream).  Returns the raw figures; the caller turns them into the JSON line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from jpeg_decoder_amd import sharding, synth
    desc_text, w, h, samp, nf_default = WORKLOADS[workload]
    nf = frames or nf_default
    d = jg.make_desc(w, h, samp)              # host-only geometry
    full_rows = d.nbmcuV
    qt = synth.shaun_qtables()

    # ---- what this rank does in one step -------------------------------------------------------------
    # launches: list of frame counts; every launch works on its own buffers
    if strong and workload == "c5":
        my_images = sharding.shard_images(images, rank, world)           # image i -> rank i mod N
        launches = [min(nf, len(my_images) - i) for i in range(0, len(my_images), nf)]
        total_pixels_per_step = images * w * h                            # whole job, all ranks
        shard_text = f"strong: {images} independent images, image i on rank i mod {world}, {len(my_images)} on this rank in launches of {nf}; no collective"
    elif strong:
        m0, m1 = sharding.shard_mcu_rows(full_rows, rank, world)               # MCU-row stripe of every frame
        d.nbmcuV = m1 - m0
        launches = [nf] if m1 > m0 else []
        total_pixels_per_step = nf * w * h
        shard_text = f"strong: {nf} frames per step cut into MCU-row stripes, rank g decodes rows [g*{full_rows}/{world}, (g+1)*{full_rows}/{world}); no collective"
    else:
        launches = [nf]
        total_pixels_per_step = world * nf * w * h
        shard_text = "weak: independent images per rank, no collective"
    nb = [jg.plane_blocks(d, c) for c in range(d.nbcomp)]
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    frames_resident = sum(launches)
    bytes_per_frame = algorithmic_bytes(jg, d)

    kernel_ms, t_local, sha = [], 0.0, None
    if dry_run:
        t_local = 1e-3 * steps * max(1, len(launches))
        kernel_ms = [1.0] * (steps * max(1, len(launches)))
    elif launches:
        # distinct synthetic frames, generated on the device (synth.synth_blocks_torch: the bytes of synth.synth_blocks);
        # beyond KMAX the buffers repeat frame f mod KMAX -- every frame still has its own memory
        KMAX = 512
        kdist = min(frames_resident, KMAX)
        planes = []
        for c in range(d.nbcomp):      # frame f of this rank = blocks [(rank * frames + f) * nb, + nb) of component c's stream
            t = torch.empty((frames_resident * nb[c], 64), dtype=torch.int16, device="cuda")
            t[:kdist * nb[c]] = synth.synth_blocks_torch(kdist * nb[c], 0x5EED0000 + c, rank * frames_resident * nb[c], 0, "cuda", 1 << 19)
            for f0 in range(kdist, frames_resident, kdist):
                n = min(kdist, frames_resident - f0) * nb[c]
                t[f0 * nb[c]:f0 * nb[c] + n] = t[:n]
            planes.append(t)
        out = torch.empty(frames_resident * rows * rb, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.Stream()      # a real (non-NULL) HIP stream: kernels AND events go on it
        firsts = [sum(launches[:i]) for i in range(len(launches))]

        def launch(i):
            f0 = firsts[i]
            jg.backend_run_batch_device(d, launches[i], [planes[c].data_ptr() + f0 * nb[c] * 128 for c in range(d.nbcomp)], None, qt,
                                        out.data_ptr() + f0 * rows * rb, 0, 0, stream.cuda_stream)

        def step():
            for i in range(len(launches)):
                launch(i)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
            for _ in range(max(1, 20 // len(launches))):
                step()
            torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        fence()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps * len(launches))]
        t0 = time.perf_counter()
        k = 0
        for _ in range(steps):
            for i in range(len(launches)):
                a, b = ev[k]
                k += 1
                a.record(stream)
                launch(i)
                b.record(stream)
        torch.cuda.synchronize()
        t_local = time.perf_counter() - t0
        fence()
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
        if verify:      # what the tests compare with the oracle: this rank's first raster
            import hashlib
            sha = hashlib.sha256(out[:rows * rb].cpu().numpy().tobytes()).hexdigest()
        del planes, out
        torch.cuda.empty_cache()
    if world > 1:
        dev = "cpu" if dry_run or host_collectives else "cuda"
        tt = torch.tensor([t_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_max = float(tt.item())
        mine = torch.tensor([float(np.mean(kernel_ms)) * 1e3 if kernel_ms else 0.0], dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank_us = [round(float(x.item()), 2) for x in per_rank]
        shas = [None] * world
        if verify:
            dist.all_gather_object(shas, sha)
    else:
        t_max = t_local
        per_rank_us = [round(float(np.mean(kernel_ms)) * 1e3, 2)]
        shas = [sha]
    return dict(first_raster_sha256=shas, desc_text=desc_text, w=w, h=h, samp=samp, nf=nf, desc=d, qt=qt, launches=launches, kernel_ms=kernel_ms, t_max=t_max,
                per_rank_us=per_rank_us, total_pixels_per_step=total_pixels_per_step, shard_text=shard_text, nb=nb,
                frames_resident=frames_resident, bytes_per_frame=bytes_per_frame)


def roofline_of(m):
    """(launch seconds, bytes per launch, achieved GB/s) of the full-size launches of one measure()."""
    import numpy as np
    launches, kernel_ms = m["launches"], m["kernel_ms"]
    full = [ms for j, ms in enumerate(kernel_ms) if launches and launches[j % len(launches)] == max(launches)]
    launch_s = float(np.mean(full)) / 1e3 if full else float("nan")
    bytes_per_launch = (max(launches) if launches else 0) * m["bytes_per_frame"]
    achieved = bytes_per_launch / launch_s / 1e9 if full else 0.0
    return full, launch_s, bytes_per_launch, achieved


def other_workloads(jg, steps=10):
    """The other GPU configurations of BASELINE.json (configs[2..4]) and ONE c2 frame per launch, each for a few steps on
    this box, so that every configuration is timed by whoever runs the default bench line.  Never part of `value`."""
    import torch
    res = {}
    for wl in ("c3", "c4", "c5"):
        m = measure(jg, wl, 0, steps, 2, 300.0, False, 0, 0, 1, False)
        full, launch_s, bpl, achieved = roofline_of(m)
        res[wl] = {"workload": WORKLOADS[wl][0], "frames_per_launch": m["nf"], "steps": steps, "avg_launch_us": round(launch_s * 1e6, 2),
                   "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
                   "MPix_s": round(m["nf"] * m["w"] * m["h"] / launch_s / 1e6, 1)}
    # configs[1] read literally: ONE 4096x4096 4:4:4 frame per launch (16 resident frames in turn, so that the 256 MB
    # Infinity Cache cannot serve it): launches back to back, and launch + synchronise after every frame
    from jpeg_decoder_amd import synth
    desc_text, w, h, samp, _ = WORKLOADS["c2"]
    d = jg.make_desc(w, h, samp)
    qt = synth.shaun_qtables()
    nb = [jg.plane_blocks(d, c) for c in range(3)]
    rows, rb = jg.raster_rows(d), jg.raster_row_bytes(d)
    K = 16
    planes = [torch.cat([synth.synth_blocks_torch(nb[c], 0x5EED0000 + c, f * nb[c], 0, "cuda", 1 << 19) for f in range(K)]) for c in range(3)]
    out = torch.empty(K * rows * rb, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()

    def launch(f):
        jg.backend_run_batch_device(d, 1, [planes[c].data_ptr() + f * nb[c] * 128 for c in range(3)], None, qt, out.data_ptr() + f * rows * rb,
                                    0, 0, stream.cuda_stream)
    for f in range(4 * K):
        launch(f % K)
    torch.cuda.synchronize()
    n = 20 * K
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for f in range(n):
        launch(f % K)
    e1.record(stream)
    torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) * 1e-3 / n
    t0 = time.perf_counter()
    for f in range(n):
        launch(f % K)
        stream.synchronize()
    lat = (time.perf_counter() - t0) / n
    fb = algorithmic_bytes(jg, d)
    res["c2_one_frame_per_launch"] = {
        "workload": "one 4096x4096 4:4:4 frame per launch (4096 workgroups on 2304 resident slots: ramp-up and drain are not amortised)",
        "back_to_back_us": round(b2b * 1e6, 2), "back_to_back_frac": round(fb / b2b / 1e9 / HBM_PEAK_GBPS, 4),
        "launch_and_sync_us": round(lat * 1e6, 2), "launch_and_sync_frac": round(fb / lat / 1e9 / HBM_PEAK_GBPS, 4)}
    del planes, out
    torch.cuda.empty_cache()
    return res


def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching the HIP/HSA runtime (the parent of the ranks must stay clear of the GPU):
    the *_VISIBLE_DEVICES lists if set, else the kfd topology nodes that have SIMDs."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        return None
    return n


def spawn_ranks(args):
    """No launcher around us: start the N ranks as children (nothing here has touched a GPU yet) and pass their
    status on."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.realpath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per launch (0 = workload default)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--images", type=int, default=8192, help="strong scaling of c5: images in the whole job")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short runs of c3, c4, c5 and of one c2 frame per launch")
    ap.add_argument("--share-device", action="store_true",
                    help="N > 1 ranks on ONE GPU (device 0), gloo for the three tiny collectives: exercises the N > 1 path with real kernels "
                         "on a one-GPU box (tests); not a scaling measurement, and the line says so")
    ap.add_argument("--verify", action="store_true", help="add the SHA-256 of every rank's first raster to the line (tests compare it with the oracle)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding / reporting path only: gloo, no GPU, no kernel (CPU test of the N > 1 path)")
    ap.add_argument("--prewarm-ms", type=float, default=1500.0,
                    help="untimed launches before the W warmup steps until the chip's clock governor has settled "
                         "(launch time drifts 138 -> 178 -> 127 us over the first ~25 ms of load on MI355X)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        if not args.dry_run and not args.share_device:
            n = visible_gpu_count()          # no HIP call in the parent of the ranks
            if n is not None and n < args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but only {n} GPU(s) visible")
        raise SystemExit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line printed must describe the run that happened")

    import numpy as np
    import torch
    import torch.distributed as dist
    from jpeg_decoder_amd import sharding

    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo")
    elif args.share_device:
        local_rank = 0
        torch.cuda.set_device(0)
        if world > 1:
            dist.init_process_group("gloo")
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import jpeg_decoder_amd as jg
    from jpeg_decoder_amd import synth
    if not args.dry_run:
        jg.init(local_rank)

    m = measure(jg, args.workload, args.frames, args.steps, args.warmup, args.prewarm_ms, args.scaling == "strong", args.images,
                rank, world, args.dry_run, host_collectives=args.share_device, verify=args.verify)
    desc_text, w, h, samp, nf, d = m["desc_text"], m["w"], m["h"], m["samp"], m["nf"], m["desc"]
    launches, kernel_ms, t_max, per_rank_us = m["launches"], m["kernel_ms"], m["t_max"], m["per_rank_us"]
    total_pixels_per_step, shard_text, nb, frames_resident, bytes_per_frame = m["total_pixels_per_step"], m["shard_text"], m["nb"], m["frames_resident"], m["bytes_per_frame"]
    strong = args.scaling == "strong"
    qt = m["qt"]
    value = total_pixels_per_step * args.steps / t_max / 1e6
    # roofline of the dominant kernel: full-size launches of this rank
    full, launch_s, bytes_per_launch, achieved = roofline_of(m)
    traffic, traffic_source = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-measured HBM bytes per launch, if collected
    if os.path.exists(tf) and not strong:
        try:
            traffic = json.load(open(tf)).get(f"{args.workload}_f{nf}")
            if traffic is not None:
                traffic_source = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                  "(tools/prof_pmc.sh), collected separately, not measured in this run")
        except Exception:
            traffic = None
    res = {
        "metric": "MPix/s IQZZ+IDCT+upsample+YCbCr->RGB", "value": round(value, 1), "unit": "MPix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_max / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32+i32 (u8 out)",
        "data": "synthetic" + (" (frames beyond 512 per rank repeat frame f mod 512, each in its own memory)" if frames_resident > 512 else ""),
        "config": {"workload": f"{args.workload}: {desc_text}", "frames_per_launch": nf, "launches_per_step_rank0": len(launches),
                   "width": w, "height": h, "sampling": [list(s) for s in samp], "blocks_per_frame_rank0": int(sum(nb)), "sharding": shard_text},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "backend_kernel", "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(launch_s * 1e6, 2),
                     "min_launch_us": round(min(full) * 1e3, 2) if full else None, "per_rank_avg_launch_us": per_rank_us},
    }
    if args.dry_run:
        res["dry_run"] = True
    if args.verify:
        res["first_raster_sha256"] = m["first_raster_sha256"]
    if args.share_device:
        res["share_device"] = f"all {world} ranks on device 0 (gloo for the collectives): a test of the N > 1 path, not a scaling measurement"
    if os.environ.get("JPEGGPU_BENCH_TRACE") and rank == 0:
        print("launch_us:", " ".join(f"{x * 1e3:.0f}" for x in kernel_ms), file=sys.stderr)
    if rank == 0 and not args.dry_run:
        # how fast THIS box streams: a plain 1 GiB device-to-device copy (torch's copy kernel), read + write bytes per second.
        # Boxes differ by +-3.5 %; the guide's figure for a float4 copy is 6.29 TB/s = 79 % of the peak.
        try:
            a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            b = torch.empty_like(a)
            for _ in range(3):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            res["roofline"]["box_torch_copy_GBps"] = round(copy_gbps, 1)
            res["roofline"]["achieved_vs_box_torch_copy"] = round(achieved / copy_gbps, 4)
            del a, b
        except Exception as e:   # noqa: BLE001  (a calibration figure only)
            res["roofline"]["box_torch_copy_GBps"] = None
    if rank == 0:
        solo = world == 1 and not args.dry_run
        if solo and not args.no_other_workloads and args.workload == "c2" and not strong:
            try:      # reported next to the headline, never part of it
                res["roofline"]["other_workloads"] = other_workloads(jg)
            except Exception as e:      # noqa: BLE001
                res["roofline"]["other_workloads"] = {"error": repr(e)}
        res["cpu_baseline"] = cpu_baseline(args.workload, w, h, samp, qt) if solo and not args.no_cpu_baseline else None
        if solo and not args.no_end_to_end:
            try:
                res["end_to_end"] = end_to_end(jg, np, jg.make_desc(w, h, samp), qt, w, h)
                res["end_to_end"]["whole_file"] = whole_file_times(w, h, samp)
            except Exception as e:      # reported, never fatal for the headline
                res["end_to_end"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
