#!/bin/bash
# Runs the PMC passes of the bench, one counter group per run (rocprofv3 does not split
# counters over passes).  Usage on the GPU box:  bash tools/pmc_passes.sh <tag>
set -e
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- \
      python3 $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu --no-extras --pipeline 1 > $OUT/$name.log 2>&1
  echo "pass $name done"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
