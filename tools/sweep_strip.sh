#!/bin/bash
mkdir -p gpurun_out/r02c
for s in 0 64 96 128 192 256 384 544; do
  echo "== STRIP=$s" >> gpurun_out/r02c/strip_sweep.txt
  SIFT_AMD_STRIP=$s timeout -k 10 120 python tools/kernel_bench.py --batch 128 --iters 20 --scalespace-only 2>&1 | grep "blur launches\|blur_fused\|oversample" >> gpurun_out/r02c/strip_sweep.txt || exit 1
done
