#!/bin/bash
# Regenerates the judged artefacts under gpurun_out/ on the GPU box (copy into profiles/ afterwards):
#   bash tools/refresh_profiles.sh <tag>
set -e
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
# octave-0 blur launch of the bench's batch of 1080p images that evaluates every pixel (R = 10): 8 x 4 tiles
# of 480 x 544 outputs per image, 512 threads (the launch of the last image, evaluated only where it is read,
# has another grid and stays out of the dominant group)
BATCH=320          # bench.py's default number of images per launch
DOM_BLOCKS=$((8 * 4 * BATCH))
DOM_THREADS=512
# the radius-5/7 pair of octave 0 as one launch: 8 x 2 tiles of 480 x 1080 outputs per image, 1024 threads
PAIR_BLOCKS=$((8 * 2 * BATCH))
PAIR_THREADS=1024
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 $ROOT/bench.py --no-cpu --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cd $ROOT
python3 tools/prof_summary.py $OUT/stats > $OUT/kernel_stats.md
python3 tools/trace_by_grid.py $OUT/stats "" 40 > $OUT/by_grid.txt
python3 tools/dominant_from_trace.py $OUT/stats $DOM_BLOCKS 40 $PAIR_BLOCKS > $OUT/dominant_kernel_rocprof.txt
bash tools/pmc_passes.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc k_blur_ > $OUT/pmc_blur.txt
python3 tools/make_traffic.py $OUT/pmc $((DOM_BLOCKS * DOM_THREADS)),$((PAIR_BLOCKS * PAIR_THREADS)) $BATCH $OUT/traffic.json > /dev/null 2> $OUT/traffic.err || true
[ -s $OUT/traffic.json ] && cp $OUT/traffic.json $ROOT/profiles/traffic.json   # the bench line quotes it
bash tools/pmc_one.sh $TAG/grbm GRBM_GUI_ACTIVE > $OUT/grbm.log 2>&1
python3 tools/clock_from_pmc.py $OUT/grbm > $OUT/clock_per_kernel.md
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 python3 bench.py --no-cpu --no-extras --pipeline 1 > $OUT/bench_pipeline1.json 2>> $OUT/bench.err
timeout -k 10 200 python3 tools/kernel_bench.py --batch 128 --roofline > $OUT/per_kernel_roofline_b128.md 2>&1
timeout -k 10 200 python3 tools/kernel_bench.py --batch 128 --kind white > $OUT/per_kernel_white_b128.md 2>&1
timeout -k 10 200 python3 tools/kernel_bench.py --batch 128 --kind natural > $OUT/per_kernel_natural_b128.md 2>&1
timeout -k 10 200 python3 tools/kernel_bench.py --batch 1 > $OUT/per_kernel_single_1080p.md 2>&1
timeout -k 10 200 python3 tools/kernel_bench.py --batch 1 --w 3840 --h 2160 > $OUT/per_kernel_single_4k.md 2>&1
# what the hardware gives: copy ceiling of HBM, issue rate of packed against plain f32 instructions
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/copy_bw tools/membench/copy_bw.hip && timeout -k 10 120 /tmp/copy_bw > $OUT/micro_copy_bw.txt
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/valu_sgpr tools/membench/valu_sgpr.hip && timeout -k 10 120 /tmp/valu_sgpr > $OUT/micro_valu_rate.txt
echo done
