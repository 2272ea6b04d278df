"""First-contact GPU script: parity of every entry point vs the oracle + a rough kernel timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (first, so that one HIP runtime is shared)
import jpeg_decoder_amd as jg  # noqa: E402
from oracle import pyoracle  # noqa: E402
import helpers  # noqa: E402

jg.init(0)
orc = pyoracle.oracle
qt = helpers.qtables_ramp()
bad = 0

# stage level
blk = np.concatenate([pyoracle.synth_blocks(4000, 1, 0, k) for k in (0, 1, 2)])
a = jg.iquant_blocks(blk, 0, 63, qt[2]); b = orc.iquant_blocks(blk, 0, 63, qt[2])
print("iquant", np.array_equal(a, b)); bad += not np.array_equal(a, b)
za = jg.izz_blocks(a); zb = orc.izz_blocks(b)
print("izz", np.array_equal(za, zb)); bad += not np.array_equal(za, zb)
pa = jg.idct_opt_blocks(za); pb = orc.idct_opt_blocks(zb)
print("idct", np.array_equal(pa, pb), int((pa != pb).sum())); bad += not np.array_equal(pa, pb)
n = 1 << 24
idx = np.arange(n, dtype=np.uint32)
y, cb, cr = (idx & 255).astype(np.uint8), ((idx >> 8) & 255).astype(np.uint8), ((idx >> 16) & 255).astype(np.uint8)
ca = jg.ycc2rgb_pixels(y, cb, cr); cbb = orc.ycc2rgb_pixels(y, cb, cr)
print("ycc2rgb all 2^24", np.array_equal(ca, cbb)); bad += not np.array_equal(ca, cbb)

# frame level
combos = [((1, 1),), ((1, 1), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)),
          ((2, 2), (2, 1), (1, 1)), ((4, 1), (1, 1), (1, 1)), ((4, 2), (2, 1), (1, 2)), ((2, 2), (2, 2), (2, 2))]
for samp in combos:
    for (w, h) in ((67, 45), (300, 225), (1920, 1080), (2500, 40)):
        for aq in (1, 0):
            d = helpers.oracle_desc(w, h, samp, apply_iquant=bool(aq))
            planes = helpers.synth_planes(d, 77 + w, kind=0 if aq else 1)
            want = orc.backend_run(d, planes, qt)
            got = jg.backend_run(helpers.to_gpu_desc(jg, d), planes, qt)
            ok = np.array_equal(want, got)
            if not ok:
                bad += 1
                diff = np.argwhere(want != got)
                print("MISMATCH", samp, w, h, aq, len(diff), diff[:5])
print("frame-level parity done, failures so far:", bad)

# rough timing, device-resident, C2 / C3 like
for name, (w, h, samp) in {"c2_4096_444": (4096, 4096, ((1, 1),) * 3), "c3_8192_420": (8192, 8192, ((2, 2), (1, 1), (1, 1))),
                           "c4_4096_422": (4096, 4096, ((2, 1), (1, 1), (1, 1)))}.items():
    d = jg.make_desc(w, h, samp)
    planes = [torch.from_numpy(pyoracle.synth_blocks(jg.plane_blocks(d, c), 5 + c)).cuda() for c in range(3)]
    out = torch.empty(jg.raster_rows(d) * jg.raster_row_bytes(d), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ptrs = [p.data_ptr() for p in planes]
    for _ in range(3):
        jg.backend_run_device(d, ptrs, qt, out.data_ptr(), 0, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 20
    for _ in range(K):
        jg.backend_run_device(d, ptrs, qt, out.data_ptr(), 0, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    nbytes = sum(p.numel() * 2 for p in planes) + out.numel()
    print(f"{name}: {dt*1e6:.1f} us  {w*h/dt/1e6:.0f} MPix/s  {nbytes/dt/1e9:.0f} GB/s ({nbytes/dt/8e12*100:.1f}% of 8 TB/s)")
    # device result parity on a stripe against the oracle
    od = helpers.oracle_desc(w, h, samp)
    hp = [p.cpu().numpy() for p in planes]
    want = orc.backend_run(od, hp, qt, mcu_rows=(3, 5))
    rows_per = od.max_v * 8
    got = out.cpu().numpy().reshape(jg.raster_rows(d), -1)[3 * rows_per:5 * rows_per]
    ok = np.array_equal(want[3 * rows_per:5 * rows_per], got)
    print("   stripe parity:", ok); bad += not ok
print("TOTAL FAILURES", bad)
sys.exit(1 if bad else 0)
