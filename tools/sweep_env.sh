#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." ; A/B of one environment knob over the workloads, 2 rounds
VAR=$1; VALS=$2; WLS=${WLS:-"c2 c3 c4 c5"}
for round in 1 2; do for WL in $WLS; do for V in $VALS; do
  echo -n "round=$round wl=$WL $VAR=$V : "
  env $VAR=$V timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload $WL 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], 'MPix/s', r['roofline']['achieved'], 'GB/s', r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'])"
done; done; done
