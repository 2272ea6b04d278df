"""TEST INFRASTRUCTURE ONLY: regenerates tests/golden/ from the REAL reference objects
(oracle/_ref/libjpegref_strict.so, built by oracle/Makefile from /root/reference).  Run in the build
container:  python oracle/make_golden.py

Everything written is data: input coefficients, the reference's outputs for them, and copies of the
reference's own fixture files (two JPEG inputs and their valid_out/ goldens).
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
import helpers  # noqa: E402

REF = os.environ.get("JPEG_REF_DIR", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ref = pyoracle.load_ref("strict")
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
qt = helpers.qtables_ramp()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kat_frames(meta, blocks):
    meta["kat_frames"] = {}
    for layout in ("gray", "444", "420"):
        for (aq, idq, tag) in ((1, [0, 0, 0], "ramp"), (1, [2, 2, 2], "wrap"), (0, [0, 1, 1], "noq")):
            samp, w, h, planes = helpers.kat_frame(blocks, layout)
            d = helpers.oracle_desc(w, h, samp, idq=idq[:len(samp)], apply_iquant=bool(aq))
            assert all(planes[c].shape == helpers.plane_shape(d, c) for c in range(len(samp)))
            out = ref.backend_run(d, planes, qt)
            meta["kat_frames"][f"{layout}_{tag}"] = {"layout": layout, "apply_iquant": aq, "idq": idq, "rows": int(out.shape[0]), "sha256": sha(out)}


if "--kat-only" in sys.argv:      # adds meta.json["kat_frames"] from the committed stage vectors, leaves everything else alone
    mp = os.path.join(OUT, "meta.json")
    meta = json.load(open(mp))
    kat_frames(meta, np.load(os.path.join(OUT, "stage_vectors.npz"))["blocks"])
    with open(mp, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("kat_frames:", {k: v["sha256"][:12] for k, v in meta["kat_frames"].items()})
    sys.exit(0)


# ---- stage-level vectors --------------------------------------------------------------------
imp = np.zeros((64 * 6, 64), dtype=np.int16)
for k in range(64):
    for j, v in enumerate((1, -1, 1024, -1024, 2047, -2047)):
        imp[k * 6 + j, k] = v
special = np.zeros((4, 64), dtype=np.int16)
special[1, :] = 32767; special[2, :] = -32768; special[3, 0] = -1024
blocks = np.concatenate([pyoracle.synth_blocks(96, 0xA0, 0, 0), pyoracle.synth_blocks(64, 0xA1, 0, 1),
                         pyoracle.synth_blocks(64, 0xA2, 0, 2), imp, special])
stage = {"blocks": blocks}
for name, q in (("ramp", qt[0]), ("wrap", qt[2])):
    dq = ref.iquant_blocks(blocks, 0, 63, q)
    zz = ref.izz_blocks(dq)
    px = ref.idct_opt_blocks(zz)
    stage[f"q_{name}"] = q; stage[f"iquant_{name}"] = dq; stage[f"izz_{name}"] = zz; stage[f"idct_{name}"] = px
stage["iquant_band_5_20"] = ref.iquant_blocks(blocks, 5, 20, qt[0])
np.savez_compressed(os.path.join(OUT, "stage_vectors.npz"), **stage)

# ---- ycc2rgb: hash of all 2^24 outputs + a raw sample ----------------------------------------
idx = np.arange(1 << 24, dtype=np.uint32)
y, cb, cr = (idx & 255).astype(np.uint8), ((idx >> 8) & 255).astype(np.uint8), ((idx >> 16) & 255).astype(np.uint8)
allrgb = ref.ycc2rgb_pixels(y, cb, cr)
pick = np.concatenate([np.arange(0, 1 << 24, 4099), ((228 << 16) | (28 << 8)) + np.arange(256)])  # incl. the y-dependent pair
np.savez_compressed(os.path.join(OUT, "ycc2rgb_vectors.npz"), index=pick.astype(np.uint32), rgb=allrgb[pick])
meta = {"ycc2rgb_all_sha256": sha(allrgb), "frames": {}}
kat_frames(meta, blocks)

# ---- frame-level vectors: small rasters raw, larger ones as hashes ---------------------------
combos = {"gray": ((1, 1),), "444": ((1, 1),) * 3, "422": ((2, 1), (1, 1), (1, 1)), "420": ((2, 2), (1, 1), (1, 1)),
          "440": ((1, 2), (1, 1), (1, 1)), "411": ((4, 1), (1, 1), (1, 1)), "y22_cb21_cr11": ((2, 2), (2, 1), (1, 1)),
          "y42_cb21_cr12": ((4, 2), (2, 1), (1, 2)), "y22_cb22_cr22": ((2, 2),) * 3, "y14_c11": ((1, 4), (1, 1), (1, 1)),
          "y42_cb42_cr21": ((4, 2), (4, 2), (2, 1)), "y22_cb12_cr21": ((2, 2), (1, 2), (2, 1))}
small = {}
for name, samp in combos.items():
    for (w, h, aq, kind, seed) in ((67, 45, 1, 0, 1000), (160, 32, 0, 1, 2000), (1920, 1080, 1, 0, 3000), (517, 260, 1, 2, 4000)):
        d = helpers.oracle_desc(w, h, samp, idq=[0, 1, 2] if kind == 2 else None, apply_iquant=bool(aq))
        planes = helpers.synth_planes(d, seed, kind)
        out = ref.backend_run(d, planes, qt)
        key = f"{name}_{w}x{h}_q{aq}_k{kind}"
        meta["frames"][key] = {"sampling": samp, "width": w, "height": h, "apply_iquant": aq, "kind": kind, "seed": seed,
                               "idq": [0, 1, 2] if kind == 2 else [0, 1, 1], "rows": int(out.shape[0]), "sha256": sha(out)}
        if w <= 160:
            small[key] = out
np.savez_compressed(os.path.join(OUT, "frame_vectors_small.npz"), **small)

# ---- BASELINE.json sizes: SHA-256 of the WHOLE raster the reference objects produce ----------------------
# (same frames as tests/test_gpu_parity.py::test_baseline_sizes builds: seed 0x5EED0000 + len(name))
big = {"c2_4096x4096_444": (4096, 4096, ((1, 1),) * 3), "c3_8192x8192_420": (8192, 8192, ((2, 2), (1, 1), (1, 1))),
       "c4_4096x4096_422": (4096, 4096, ((2, 1), (1, 1), (1, 1))), "c1_512x512_gray": (512, 512, ((1, 1),)),
       "c5_1920x1080_420": (1920, 1080, ((2, 2), (1, 1), (1, 1)))}
meta["baseline_sizes"] = {}
for name, (w, h, samp) in sorted(big.items()):
    d = helpers.oracle_desc(w, h, samp)
    planes = helpers.synth_planes(d, 0x5EED0000 + len(name))
    out = ref.backend_run(d, planes, qt)
    band = d.max_v * 8
    meta["baseline_sizes"][name] = {"width": w, "height": h, "sampling": samp, "seed": 0x5EED0000 + len(name), "rows": int(out.shape[0]),
                                    "sha256": sha(out), "sha256_first_band": sha(out[:band]), "sha256_last_band": sha(out[-band:])}
    print("baseline size", name, out.shape, meta["baseline_sizes"][name]["sha256"][:16])
    del out, planes

# ---- the reference's own end-to-end goldens --------------------------------------------------
ref.lib.ref_dump_baseline_planes.argtypes = [C.c_char_p, C.POINTER(pyoracle.FrameDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
for jpg, gold in (("invader_normal.jpeg", "invader.pgm"), ("shaun_the_sheep.jpeg", "shaun_the_sheep.ppm")):
    src = os.path.join(REF, "test", "test_file", jpg)
    shutil.copyfile(src, os.path.join(OUT, jpg)); os.chmod(os.path.join(OUT, jpg), 0o644)
    shutil.copyfile(os.path.join(REF, "valid_out", gold), os.path.join(OUT, gold)); os.chmod(os.path.join(OUT, gold), 0o644)
    d = pyoracle.FrameDesc()
    pl = (C.c_void_p * 3)()
    q = np.zeros((4, 64), dtype=np.uint16)
    present = (C.c_int * 4)()
    rc = ref.lib.ref_dump_baseline_planes(src.encode(), C.byref(d), pl, q.ctypes.data, C.cast(present, C.c_void_p))
    assert rc == 0, rc
    arrs = {}
    for c in range(d.nbcomp):
        nb = d.nbmcuH * d.comp[c].hs * d.nbmcuV * d.comp[c].vs
        arrs[f"plane{c}"] = np.ctypeslib.as_array(C.cast(pl[c], C.POINTER(C.c_int16)), shape=(nb, 64)).copy()
    arrs["qtables"] = q
    arrs["qt_present"] = np.array(list(present), dtype=np.int32)
    arrs["desc"] = np.frombuffer(bytes(d), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, jpg.split(".")[0] + "_planes.npz"), **arrs)
    # sanity: reference back end on the dumped planes reproduces the reference's golden file
    planes = [arrs[f"plane{c}"] for c in range(d.nbcomp)]
    qts = [q[i] if present[i] else None for i in range(4)]
    out = ref.backend_run(d, planes, qts)
    gold_bytes = open(os.path.join(REF, "valid_out", gold), "rb").read()
    hdr = f"P{6 if d.nbcomp == 3 else 5}\n{d.width} {d.height}\n255\n".encode()
    assert gold_bytes == hdr + out.tobytes(), f"{gold}: planes + back end do not reproduce the golden file"
    print(gold, "reproduced from dumped planes:", len(gold_bytes), "bytes")

with open(os.path.join(OUT, "meta.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
print("golden fixtures written to", OUT)
