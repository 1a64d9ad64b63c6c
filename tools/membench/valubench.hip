// VALU issue-rate calibration: how many wave64 f32 add/mul instructions per cycle a SIMD sustains
// at various occupancies (non-fused mul and add, 8 independent chains per thread).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_valu(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 4
        for (int u = 0; u < 4; ++u) {
            x0 = __fadd_rn(__fmul_rn(x0, a), b); x1 = __fadd_rn(__fmul_rn(x1, a), b);
            x2 = __fadd_rn(__fmul_rn(x2, a), b); x3 = __fadd_rn(__fmul_rn(x3, a), b);
            x4 = __fadd_rn(__fmul_rn(x4, a), b); x5 = __fadd_rn(__fmul_rn(x5, a), b);
            x6 = __fadd_rn(__fmul_rn(x6, a), b); x7 = __fadd_rn(__fmul_rn(x7, a), b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    float* out;
    CK(hipMalloc(&out, 4096 * 1024 * 4));
    const int iters = 4096;
    for (int wpc : {4, 8, 16, 32}) {          // waves per CU
        const int blocks = 256 * wpc / 4;      // 256-thread blocks
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k_valu<<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        k_valu<<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double winstr = (double)blocks * 4 * iters * 4 * 16;   // wave-instructions
        const double per_simd_per_us = winstr / 1024 / (ms * 1e3);
        printf("waves/CU %2d: %.3f ms, %.1f wave-instr/us/SIMD  (at 2.4 GHz: %.2f cycles per wave-instr)\n",
               wpc, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
    }
    return 0;
}
