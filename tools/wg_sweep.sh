#!/bin/bash
# workgroup size sweep with the 512-thread build: JPEGGPU_TARGET_BLOCKS = blocks (threads) per tile
run() { echo -n "wl=$1 blocks=$2 : "; JPEGGPU_LIB=$PWD/jpeg-decoder_amd/variants/f_512.so JPEGGPU_TARGET_BLOCKS=$2 timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload $1 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'])"; }
for round in 1 2; do
for WL in c2 c3 c4; do for B in 256 384 512; do run $WL $B; done; done
done
