// Does v_writelane_b32 right after the v_cmp that produced its SGPR operand see the new value?
//   hipcc --offload-arch=gfx950 -O3 -o writelane_test tools/membench/writelane_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void k(const int* keys, unsigned long long* out)
{
    const int lane = threadIdx.x;
    const int key = keys[blockIdx.x * 64 + lane];
    int lo = 0, hi = 0;
#pragma unroll
    for (int L = 0; L < 8 && MODE != 9; ++L) {
        const unsigned long long m = __ballot(key == L);
        if (MODE == 0) { if (lane == L) { lo = (int)(unsigned)m; hi = (int)(unsigned)(m >> 32); } }
        if (MODE == 1) {
            asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
            asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L));
        }
        if (MODE == 2) {
            asm("s_nop 4\n v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
            asm("s_nop 4\n v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L));
        }
        if (MODE >= 4 && MODE <= 7) {
            // MODE - 4 wait states in front
            if (MODE == 4) { asm("s_nop 0\n v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
                             asm("s_nop 0\n v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L)); }
            if (MODE == 5) { asm("s_nop 1\n v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
                             asm("s_nop 1\n v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L)); }
            if (MODE == 6) { asm("s_nop 2\n v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
                             asm("s_nop 2\n v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L)); }
            if (MODE == 7) { asm("s_nop 3\n v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
                             asm("s_nop 3\n v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L)); }
        }
        if (MODE == 8) {   // through an SALU copy
            asm("s_mov_b32 s90, %1\n v_writelane_b32 %0, s90, %2" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L) : "s90");
            asm("s_mov_b32 s90, %1\n v_writelane_b32 %0, s90, %2" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L) : "s90");
        }
        if (MODE == 3) {
            asm("v_writelane_b32 %0, %1, %2\n s_nop 4" : "+v"(lo) : "s"((int)(unsigned)m), "n"(L));
            asm("v_writelane_b32 %0, %1, %2\n s_nop 4" : "+v"(hi) : "s"((int)(unsigned)(m >> 32)), "n"(L));
        }
    }
    if (MODE == 9) {
        unsigned long long m[8];
#pragma unroll
        for (int L = 0; L < 8; ++L) m[L] = __ballot(key == L);
        asm volatile("s_nop 4" :: "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(m[4]), "s"(m[5]), "s"(m[6]), "s"(m[7]));
#define WL(L) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)m[L]), "n"(L)); \
              asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)(m[L] >> 32)), "n"(L));
        WL(0) WL(1) WL(2) WL(3) WL(4) WL(5) WL(6) WL(7)
    }
    asm("s_nop 1" : "+v"(lo), "+v"(hi));
    out[blockIdx.x * 64 + lane] = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

int main()
{
    const int nb = 4096, n = nb * 64;
    int* h = (int*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = rand() % 10 - 1;
    int* d; unsigned long long *o, *r[10];
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 8);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    for (int m = 0; m < 10; ++m) {
        r[m] = (unsigned long long*)malloc(n * 8);
        if (m == 0) k<0><<<nb, 64>>>(d, o);
        if (m == 1) k<1><<<nb, 64>>>(d, o);
        if (m == 2) k<2><<<nb, 64>>>(d, o);
        if (m == 3) k<3><<<nb, 64>>>(d, o);
        if (m == 4) k<4><<<nb, 64>>>(d, o);
        if (m == 5) k<5><<<nb, 64>>>(d, o);
        if (m == 6) k<6><<<nb, 64>>>(d, o);
        if (m == 7) k<7><<<nb, 64>>>(d, o);
        if (m == 8) k<8><<<nb, 64>>>(d, o);
        if (m == 9) k<9><<<nb, 64>>>(d, o);
        hipMemcpy(r[m], o, n * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) bad += r[m][i] != r[0][i];
        printf("mode %d: %ld of %d lanes differ from the v_cndmask form\n", m, bad, n);
    }
    return 0;
}
