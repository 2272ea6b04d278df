#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <workload> ; run on the GPU box from the repo root
# kernel-trace statistics of the default bench command + separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share one)
set -u
TAG=$1; WL=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-other-workloads: the default c2 line also times c3, c4, c5 and one-frame launches (the same kernels under the same names)
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-other-workloads --workload $WL"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --steps 20 > $OUT/trace.log 2>&1 || exit 1
P="--steps 5 --warmup 1 --prewarm-ms 300"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o p -- $B $P > $OUT/pmc_sq.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -o p -- $B $P > $OUT/pmc_lds.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $B $P > $OUT/pmc_fetch.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $B $P > $OUT/pmc_write.log 2>&1 || exit 5
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
rm -rf $OUT/pmc_*/ $OUT/trace/*/*_agent_info.csv 2>/dev/null
find $OUT -name "t_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
cat $OUT/summary.txt | head -40
