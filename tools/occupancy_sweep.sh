#!/bin/bash
for round in 1 2; do for WL in c2 c3; do for L in 0 20000 26000 32000 40000 53000 64000; do
  echo -n "round=$round wl=$WL lds=$L : "
  JPEGGPU_DIAG_LDS_BYTES=$L timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload $WL 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'])"
done; done; done
