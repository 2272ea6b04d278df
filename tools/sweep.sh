#!/bin/bash
# A/B sweep of the tuning knobs on the GPU box: prints value + roofline per configuration
WLS=${WLS:-"c2 c3 c4 c5"}
TBS=${TBS:-"96 192 384"}
for WL in $WLS; do
for TB in $TBS; do
  echo -n "wl=$WL target_blocks=$TB : "
  JPEGGPU_TARGET_BLOCKS=$TB timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload $WL 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], 'MPix/s', r['roofline']['achieved'], 'GB/s', r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'])"
done; done
