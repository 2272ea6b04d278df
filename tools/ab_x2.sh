#!/bin/bash
for round in 1 2 3; do
for K in rs general; do
  echo -n "round=$round x2 $K : "
  if [ $K = general ]; then export JPEGGPU_NO_RESTAGE=1; else unset JPEGGPU_NO_RESTAGE; fi
  timeout -k 10 120 python3 tools/bench_geom.py --steps 100 --no-cpu-baseline --no-end-to-end --no-other-workloads --workload x2 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['achieved'], 'GB/s', r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'], 'min', r['roofline']['min_launch_us'])" || exit 1
done; done
