#!/bin/bash
# Collects the round's judged artefacts on ONE GPU box in ONE call: bench lines for every workload, rocprofv3 kernel
# statistics and PMC passes for c2..c5.  usage: tools/round_profiles.sh r03
TAG=${1:-r03}
mkdir -p gpurun_out/$TAG
for WL in c2 c3 c4 c5 g1; do
  timeout -k 10 200 python3 bench.py --workload $WL $( [ $WL = c2 ] || echo --no-cpu-baseline --no-end-to-end ) > gpurun_out/$TAG/bench_$WL.json 2> gpurun_out/$TAG/bench_$WL.err || exit 1
  echo "bench $WL done"
done
timeout -k 10 200 python3 bench.py --workload c5 --scaling strong --images 8192 --steps 10 --no-cpu-baseline --no-end-to-end > gpurun_out/$TAG/bench_c5_strong8192.json 2> gpurun_out/$TAG/bench_c5_strong.err || exit 1
for WL in c2 c3 c4 c5; do
  tools/prof_pmc.sh $TAG/prof_$WL $WL > /dev/null || exit 2
  echo "prof $WL done"
done
cat gpurun_out/$TAG/bench_*.json
