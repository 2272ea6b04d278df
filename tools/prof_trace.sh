#!/bin/bash
# rocprofv3 kernel-trace statistics of the default bench command for the given workloads
# usage: tools/prof_trace.sh r01 c4 c5 g1
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/trace_$WL
  mkdir -p $OUT
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --workload $WL > $OUT/bench.json 2> $OUT/err.log || exit 1
  find $OUT -name "t_kernel_stats.csv" -exec head -3 {} \;
done
