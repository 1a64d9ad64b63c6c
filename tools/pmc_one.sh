#!/bin/bash
# One PMC pass of the bench with the given counters.  Usage on the GPU box:
#   bash tools/pmc_one.sh <tag> <counter> [<counter> ...]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pass -- \
    python3 $ROOT/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu --no-extras --pipeline 1 > $OUT/pass.log 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
