// Issue rate of plain and packed float32 VALU instructions with a VGPR or an SGPR source, and
// the shader clock the GPU holds while they run (s_memtime ticks per s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O2 -o valu_sgpr tools/membench/valu_sgpr.hip && ./valu_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define BODY_V(ASM)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                         \
        REP8(REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));))      \
    }
#define BODY_S(ASM)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                         \
        REP8(REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(se), "s"(sf));))    \
    }
#define BODY_V1(ASM)                                                                             \
    for (int it = 0; it < iters; ++it) {                                                         \
        REP8(REP8(asm volatile(ASM : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x), "v"(f.x));)) \
    }
#define BODY_S1(ASM)                                                                             \
    for (int it = 0; it < iters; ++it) {                                                         \
        REP8(REP8(asm volatile(ASM : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "s"(se.x), "s"(sf.x));)) \
    }

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* stamps, int iters, v2f se, v2f sf)
{
    v2f a = {1.f, 2.f}, b = {3.f, 4.f}, c = {5.f, 6.f}, d = {7.f, 8.f};
    v2f e = {out[threadIdx.x & 7], 1.f}, f = {out[threadIdx.x & 3], 2.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) { BODY_V1("v_add_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %5, %3") }
    if (MODE == 1) { BODY_S1("v_add_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %5, %3") }
    if (MODE == 2) { BODY_S1("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %5, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %5, %3") }
    if (MODE == 3) { BODY_V("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %5, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %5, %3") }
    if (MODE == 4) { BODY_S("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %5, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %5, %3") }
    if (MODE == 5) { BODY_V("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %5, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %5, %3") }
    if (MODE == 6) { BODY_V1("v_mul_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_mul_f32 %2, %4, %2\n v_add_f32 %3, %5, %3") }
    if (MODE == 7) { BODY_V1("v_fma_f32 %0, %4, %0, %5\n v_fma_f32 %1, %5, %1, %4\n v_fma_f32 %2, %4, %2, %5\n v_fma_f32 %3, %5, %3, %4") }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
    if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.f) out[0] = 1.f;
}

template <int MODE>
static void run(const char* name, float* d, unsigned long long* st, int waves, int iters)
{
    const int blocks = 256 * waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    v2f se = {0.999f, 0.999f}, sf = {1.001f, 1.001f};
    for (int r = 0; r < 3; ++r) k<MODE><<<blocks, 256>>>(d, st, iters, se, sf);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, st, iters, se, sf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 4 * 256 * 8];
    hipMemcpy(h, st, sizeof(unsigned long long) * 2 * 4 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int w = 0; w < 4 * blocks; ++w) { cyc += h[2 * w]; rt += h[2 * w + 1]; }
    const double n = (double)iters * 64 * 4;          // instructions per wave
    const double ghz = cyc / rt * 0.1;                 // realtime ticks at 100 MHz
    printf("waves/SIMD %d  %-26s %7.3f ms  %5.2f ns/instr/SIMD  %5.2f cycles/instr/SIMD  clock %.2f GHz\n", waves, name, ms,
           ms * 1e6 / n / waves, cyc / (4.0 * blocks) / n / waves, ghz);
}

int main()
{
    float* d;
    unsigned long long* st;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * 4 * 256 * 8);
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        run<0>("v_add_f32 v,v", d, st, waves, iters);
        run<1>("v_add_f32 s,v", d, st, waves, iters);
        run<2>("v_mul_f32 s,v", d, st, waves, iters);
        run<6>("v_mul/v_add v,v", d, st, waves, iters);
        run<7>("v_fma_f32 v,v,v", d, st, waves, iters);
        run<3>("v_pk_add_f32 v,v", d, st, waves, iters);
        run<5>("v_pk_mul_f32 v,v", d, st, waves, iters);
        run<4>("v_pk_mul_f32 s,v", d, st, waves, iters);
    }
    return 0;
}
