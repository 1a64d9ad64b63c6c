#!/bin/bash
# A/B of the blur launches inside the full pipeline (contrast marks on): tools/ab_detect.sh VARIANT...
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset SIFT_AMD_LIB; else export SIFT_AMD_LIB=$PWD/sift_amd/build/variants/$v.so; fi
  timeout -k 10 120 python tools/kernel_bench.py --batch 128 --iters 20 2>&1 | grep "blur launches octave [012]\|^sum" || exit 1
done
