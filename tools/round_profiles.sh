#!/bin/bash
# Collects the round's judged artefacts on the GPU box: bench lines for every workload + rocprof
# kernel stats and PMC passes for c2 and c3.  usage: tools/round_profiles.sh r01
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
for WL in c2 c3 c4 c5; do
  timeout -k 10 200 python3 bench.py --workload $WL $( [ $WL = c2 ] || echo --no-cpu-baseline ) > gpurun_out/$TAG/bench_$WL.json 2> gpurun_out/$TAG/bench_$WL.err || exit 1
done
tools/prof_pmc.sh $TAG/prof_c2 c2 > /dev/null || exit 2
tools/prof_pmc.sh $TAG/prof_c3 c3 > /dev/null || exit 3
cat gpurun_out/$TAG/bench_*.json
