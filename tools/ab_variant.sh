#!/bin/bash
# parity tests + per-kernel timings (white noise and smooth) with a variant build of the kernels:
# tools/ab_variant.sh NAME   (sift_amd/build/variants/NAME.so; "main" = the library itself)
set -e
V=$1
if [ "$V" != main ]; then export SIFT_AMD_LIB=$PWD/sift_amd/build/variants/$V.so; fi
timeout -k 10 300 python -u -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
for kind in white smooth; do
  timeout -k 10 120 python tools/kernel_bench.py --batch 128 --iters 5 --kind $kind 2>&1 | grep -v "^blur launches\|amdgpu.ids\|octave-0"
done
