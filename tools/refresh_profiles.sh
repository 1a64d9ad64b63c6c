#!/bin/bash
# Regenerates the judged artefacts under gpurun_out/ on the GPU box (copy into profiles/ afterwards):
#   bash tools/refresh_profiles.sh <tag>
set -e
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
# octave-0 blur launch of 64 1080p images: 8 x 17 tiles of 480 x 128 outputs per image, 512 threads
DOM_BLOCKS=8704
DOM_THREADS=512
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 $ROOT/bench.py --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $ROOT
python3 tools/prof_summary.py $OUT/stats 33 > $OUT/kernel_stats.md
python3 tools/trace_by_grid.py $OUT/stats "" 40 > $OUT/by_grid.txt
python3 tools/dominant_from_trace.py $OUT/stats $DOM_BLOCKS 50 > $OUT/dominant_kernel_rocprof.txt
bash tools/pmc_passes.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc k_blur_fused > $OUT/pmc_blur.txt
python3 tools/make_traffic.py $OUT/pmc $((DOM_BLOCKS * DOM_THREADS)) 64 $OUT/traffic.json > /dev/null 2> $OUT/traffic.err || true
[ -s $OUT/traffic.json ] && cp $OUT/traffic.json $ROOT/profiles/traffic.json   # the bench line quotes it
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo done
