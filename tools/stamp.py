"""Diagnostic: average wave time per phase from a JM_DIAG_STAMP build (jpeg-decoder_amd/variants/lib_STAMP.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JPEGGPU_LIB"] = os.path.join(ROOT, "jpeg-decoder_amd", "variants", "lib_STAMP.so")
import torch
diag = torch.zeros(8 * 200000, dtype=torch.int64, device="cuda")
os.environ["JPEGGPU_DIAG_PTR"] = str(diag.data_ptr())
import jpeg_decoder_amd as jg
from jpeg_decoder_amd import synth
sys.path.insert(0, ROOT)
import bench
jg.init(0)
for wl in sys.argv[1:] or ["c2", "c3"]:
    _, w, h, samp, nf = bench.WORKLOADS[wl]
    d = jg.make_desc(w, h, samp)
    planes = [torch.from_numpy(synth.synth_blocks(jg.plane_blocks(d, c) * nf, 5 + c)).cuda() for c in range(3)]
    out = torch.empty(nf * jg.raster_rows(d) * jg.raster_row_bytes(d), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    qt = synth.shaun_qtables()
    for i in range(301):
        if i == 300:
            torch.cuda.synchronize(); diag.zero_(); torch.cuda.synchronize()
        jg.backend_run_batch_device(d, nf, [p.data_ptr() for p in planes], None, qt, out.data_ptr(), 0, 0, st.cuda_stream)
    torch.cuda.synchronize()
    import numpy as np
    allv = diag.cpu().numpy().reshape(-1, 8).astype(float)
    allv = allv[allv[:, 4] > 0]
    n = len(allv)
    v = allv.mean(axis=0)
    span = (allv[:, 5] + allv[:, 4]).max() - allv[:, 5].min()
    print(f"{wl}: waves={n}  launch span {span * 10 / 1000:.1f} us (realtime ctr);  mean wave lifetime {v[4] * 10 / 1000:.2f} us;  ticks/ns {v[:4].sum() / v[4] / 10:.2f}")
    tot = v[:4].sum()
    print(f"     share of wave lifetime: load-wait {v[0]/tot:.1%}  transform {v[1]/tot:.1%}  barrier {v[2]/tot:.1%}  pixel phase (incl. store drain) {v[3]/tot:.1%}")
