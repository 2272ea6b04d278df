#!/usr/bin/env python3
"""Per-phase instruction table of a kernel, from the ISA of a build with -DJM_PHASE_MARKS (jpeg_math.hpp: JM_MARK puts a
comment between two scheduling barriers, so an instruction is attributed to the phase it was written in; the barriers cost
the scheduler its freedom across phases, so the totals are within a few instructions of the shipped build's, not equal).

   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DJM_PHASE_MARKS --save-temps -c kernels.hip
   python3 tools/isa_phases.py kernels-hip-amdgcn-amd-amdhsa-gfx950.s backend_kernel_rs444 [--loop-trips 4/3]

Counts are STATIC (layout order); the unit_* phases are the body of the per-unit loop and run once per unit a wave owns."""
import re
import sys
from collections import OrderedDict

CLASSES = ("valu_pk", "valu", "salu", "lds", "vmem", "smem", "wait_sync")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_setreg", "s_sleep", "s_endpgm", "s_setprio")):
        return "wait_sync"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\w*%s\w*:" % re.escape(kernel), l) and (kernel + "_") not in l.split("ENS")[0])
    phases = OrderedDict()
    cur = "prologue (tile geometry, record address)"
    ops = {}
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r"; JM_PHASE (\w+)", t)
        if m:
            cur = m.group(1)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        if c is None:
            continue
        phases.setdefault(cur, dict.fromkeys(CLASSES, 0))[c] += 1
        ops.setdefault(cur, {}).setdefault(op, 0)
        ops[cur][op] += 1
    print("kernel %s (%s)" % (kernel, path.split("/")[-1]))
    print("%-42s" % "phase" + "".join("%10s" % c for c in CLASSES) + "     top opcodes")
    tot = dict.fromkeys(CLASSES, 0)
    for ph, d in phases.items():
        top = sorted(ops[ph].items(), key=lambda kv: -kv[1])[:4]
        print("%-42s" % ph + "".join("%10d" % d[c] for c in CLASSES) + "     " + ", ".join("%s x%d" % kv for kv in top))
        for c in CLASSES:
            tot[c] += d[c]
    print("%-42s" % "total (static)" + "".join("%10d" % tot[c] for c in CLASSES))


if __name__ == "__main__":
    main()
