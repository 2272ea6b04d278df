#!/bin/bash
# usage: tools/pmc_ab.sh <workload> ; SQ counter pass for every jpeg-decoder_amd/variants/*.so (GPU box, repo root)
set -u
WL=${1:-c2}
cd /tmp && export TMPDIR=/tmp
for L in $GRAFT_REPO_ROOT/jpeg-decoder_amd/variants/*.so; do
  T=$(basename $L .so); OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab/$T; mkdir -p $OUT
  JPEGGPU_LIB=$L timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/pmc_sq.log 2>&1 || exit 2
  JPEGGPU_LIB=$L timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $OUT/pmc_b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/pmc_b.log 2>&1 || echo "pass b failed for $T"
  echo "== $T"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep -v "^\"\|^$"
done
