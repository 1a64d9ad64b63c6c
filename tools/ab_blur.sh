#!/bin/bash
# A/B of the scale-space launches: tools/ab_blur.sh VARIANT... (names under sift_amd/build/variants; "main" = the library)
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset SIFT_AMD_LIB; else export SIFT_AMD_LIB=$PWD/sift_amd/build/variants/$v.so; fi
  timeout -k 10 120 python tools/kernel_bench.py --batch 128 --iters 20 --scalespace-only 2>&1 | grep "blur launches octave [012]\|blur_fused\|oversample" || exit 1
done
