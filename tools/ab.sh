#!/bin/bash
# A/B of prebuilt library variants (jpeg-decoder_amd/variants/*.so), interleaved rounds in one box
WLS=${WLS:-"c2 c3"}
ROUNDS=${ROUNDS:-"1 2 3"}
for round in $ROUNDS; do
for WL in $WLS; do
for L in jpeg-decoder_amd/variants/*.so; do
  echo -n "round=$round wl=$WL lib=$(basename $L) : "
  JPEGGPU_LIB=$PWD/$L timeout -k 10 120 python3 bench.py --steps ${STEPS:-100} --no-cpu-baseline --no-end-to-end --workload $WL 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['achieved'], 'GB/s', r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'], 'min', r['roofline']['min_launch_us'])"
done; done; done
