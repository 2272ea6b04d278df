#!/bin/bash
# tile width sweep: JPEGGPU_TILE_MCUS = MCUs per workgroup (threads = blocks rounded up to a wave)
run() { echo -n "wl=$1 T=$2 : "; JPEGGPU_TILE_MCUS=$2 timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload $1 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'])"; }
for round in 1 2; do
for T in 0 21 42 64 85; do run c2 $T; done
for T in 0 10 21 32 42; do run c3 $T; done
for T in 0 16 32 48 64; do run c4 $T; done
done
