#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: kernel stats + per-dispatch counter averages."""
import csv, glob, os, sys, collections
d = sys.argv[1]
ks = os.path.join(d, "trace", "t_kernel_stats.csv")
if os.path.exists(ks):
    print(open(ks).read())
for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
    f = os.path.join(sub, "p_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "backend_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{os.path.basename(sub):10s} {k:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
