set -e
V=$1
export SIFT_AMD_LIB=$PWD/sift_amd/build/variants/$V.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 python tools/kernel_bench.py --batch 32 --iters 5 --kind white 2>&1 | grep -v "^blur launches\|amdgpu.ids"
SIFT_AMD_LIB=$PWD/sift_amd/build/variants/${V}p.so timeout -k 10 120 python tools/descr_prof.py white
