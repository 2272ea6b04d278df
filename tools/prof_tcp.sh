#!/bin/bash
# L1 (TCP) <-> L2 (TCC) request counts for one workload: are the per-lane record loads re-fetching lines?
WL=${1:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcp_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/p1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --frames 4 --no-cpu-baseline --workload $WL > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 2; }
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --frames 4 --no-cpu-baseline --workload $WL > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 3; }
python3 - <<PY
import csv, glob, collections
for sub in ("p1","p2"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/p_counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "backend_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f"{k:34s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
PY
