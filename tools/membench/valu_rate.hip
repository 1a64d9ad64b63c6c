// Relative issue rate of a few VALU instructions on gfx950: every wave runs a long unrolled
// sequence of one instruction on independent registers; the time per instruction relative to
// v_mov_b32 tells whether the packed / 64-bit forms are full rate.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/membench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define BODY(ASM)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                       \
        REP8(REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));))    \
    }

#define BODY32(ASM)                                                                            \
    for (int it = 0; it < iters; ++it) {                                                       \
        REP8(REP8(asm volatile(ASM : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));)) \
    }

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    v2f a = {1.f, 2.f}, b = {3.f, 4.f}, c = {5.f, 6.f}, d = {7.f, 8.f};
    v2f e = {out[threadIdx.x & 7], 1.f}, f = {out[threadIdx.x & 3], 2.f};
    if (MODE == 0) { BODY32("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5") }
    if (MODE == 1) { BODY("v_pk_mov_b32 %0, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %1, %5, %4 op_sel:[1,0]\n v_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %3, %5, %4 op_sel:[1,0]") }
    if (MODE == 2) { BODY("v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %5, %4\n v_pk_add_f32 %2, %4, %5\n v_pk_add_f32 %3, %5, %4") }
    if (MODE == 3) { BODY("v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %4\n v_mov_b64 %3, %5") }
    if (MODE == 4) { BODY32("v_add_f32 %0, %4, %5\n v_add_f32 %1, %5, %4\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %5, %4") }
    if (MODE == 5) { BODY("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %5, %4\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %5, %4") }
    if (MODE == 6) { BODY("v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %0, %0, %5\n v_pk_add_f32 %0, %0, %4") }
    if (MODE == 7) { BODY32("v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %4") }
    if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.f) out[0] = 1.f;
}

template <int MODE>
static float run(float* d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    const int iters = 2000;
    const char* names[] = {"v_mov_b32", "v_pk_mov_b32", "v_pk_add_f32", "v_mov_b64", "v_add_f32", "v_pk_mul_f32",
                           "v_pk_add_f32 dependent", "v_add_f32 dependent"};
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256 * waves;       // 4 waves per block: `waves` waves per SIMD
        float t[8];
        t[0] = run<0>(d, blocks, iters); t[1] = run<1>(d, blocks, iters); t[2] = run<2>(d, blocks, iters);
        t[3] = run<3>(d, blocks, iters); t[4] = run<4>(d, blocks, iters); t[5] = run<5>(d, blocks, iters);
        t[6] = run<6>(d, blocks, iters); t[7] = run<7>(d, blocks, iters);
        const double n = (double)iters * 64 * 4;   // instructions per wave
        for (int m = 0; m < 8; ++m)
            printf("waves/SIMD %d  %-24s %8.3f ms  %6.2f ns/instr/wave  rel %.2f\n", waves, names[m], t[m],
                   t[m] * 1e6 / n, t[m] / t[0]);
    }
    return 0;
}
