#!/bin/bash
# usage: tools/prof_mem.sh <tag> <workload> ; memory-side PMC passes of the bench command (one pass per block limit)
set -u
TAG=$1; WL=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-other-workloads --workload $WL --steps 5 --warmup 1 --prewarm-ms 300"
i=0
for SET in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_m$i -o p -- $B > $OUT/pmc_m$i.log 2>&1 || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/pmc_*/
cat $OUT/summary.txt
