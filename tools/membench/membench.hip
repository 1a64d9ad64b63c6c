// Streaming-bandwidth calibration on the target GPU: what read/write mixes of the kind the
// SIFT kernels issue can reach, to put the blur kernel's achieved bandwidth in context.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ o, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) o[i] = a[i];
}
__global__ void k_r1w2(const float4* __restrict__ a, float4* __restrict__ o1, float4* __restrict__ o2, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float4 v = a[i];
        o1[i] = v;
        o2[i] = make_float4(v.x + 1.f, v.y, v.z, v.w);
    }
}
__global__ void k_r1w2_dword(const float* __restrict__ a, float* __restrict__ o1, float* __restrict__ o2, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = a[i];
        o1[i] = v;
        o2[i] = v + 1.f;
    }
}
__global__ void k_read4(const float4* __restrict__ a, float* __restrict__ o, long long n)
{
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) o[0] = s;
}
__global__ void k_read5(const float* __restrict__ a, float* __restrict__ o, long long n, long long stride)
{
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        s += a[i] + a[i + stride] + a[i + 2 * stride] + a[i + 3 * stride] + a[i + 4 * stride];
    }
    if (s == 123.456f) o[0] = s;
}
__global__ void k_write4(float4* __restrict__ o, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

__global__ void k_write4_nt(float4* __restrict__ o, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store((v4f){1.f, 2.f, 3.f, 4.f}, (v4f*)&o[i]);
}
__global__ void k_r1w2_nt(const float4* __restrict__ a, float4* __restrict__ o1, float4* __restrict__ o2, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        v4f v = __builtin_nontemporal_load((const v4f*)&a[i]);
        __builtin_nontemporal_store(v, (v4f*)&o1[i]);
        v.x += 1.f;
        __builtin_nontemporal_store(v, (v4f*)&o2[i]);
    }
}
__global__ void k_r1w2_ntst(const float4* __restrict__ a, float4* __restrict__ o1, float4* __restrict__ o2, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        v4f v = *(const v4f*)&a[i];
        __builtin_nontemporal_store(v, (v4f*)&o1[i]);
        v.x += 1.f;
        __builtin_nontemporal_store(v, (v4f*)&o2[i]);
    }
}
// one block per contiguous chunk instead of a grid-stride sweep
__global__ void k_r1w2_chunk(const float4* __restrict__ a, float4* __restrict__ o1, float4* __restrict__ o2, long long n, int per_block)
{
    long long base = (long long)blockIdx.x * per_block;
    for (int k = threadIdx.x; k < per_block; k += blockDim.x) {
        long long i = base + k;
        if (i < n) { float4 v = a[i]; o1[i] = v; o2[i] = make_float4(v.x + 1.f, v.y, v.z, v.w); }
    }
}

// tile-shaped traffic like the blur kernel: block = TWx(strip) tile of a WxH image, 8 rows per step,
// reads 1 float per (thread, row) with dword loads, writes two float4 streams
template <int TW>
__global__ void k_tile_r1w2(const float* __restrict__ a, float* __restrict__ o1, float* __restrict__ o2,
                            int w, int h, int strip, int nbx, int nby, int do_read)
{
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby;
    const long long img = (long long)(blockIdx.x / (nbx * nby)) * w * h;
    const int x0 = bx * TW, i0 = by * strip, i1 = min(i0 + strip, h);
    const int tid = threadIdx.x;
    for (int ii = i0; ii < i1; ii += 8) {
        float acc = 0.f;
        if (do_read)
            for (int g = 0; g < 8; ++g) { int x = min(x0 + tid, w - 1); acc += a[img + (long long)min(ii + g, h - 1) * w + x]; }
        for (int item = tid; item < 8 * (TW / 4); item += blockDim.x) {
            const int g = item / (TW / 4), q = item % (TW / 4);
            const int row = ii + g, x = x0 + 4 * q;
            if (row < i1 && x + 4 <= w) {
                const long long off = img + (long long)row * w + x;
                *(float4*)(o1 + off) = make_float4(acc, 1.f, 2.f, 3.f);
                *(float4*)(o2 + off) = make_float4(acc, 2.f, 2.f, 3.f);
            }
        }
    }
}

template <typename F>
static void timeit(const char* name, double bytes, F f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipDeviceSynchronize());
    const int it = 10;
    CK(hipEventRecord(a));
    for (int i = 0; i < it; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s %8.1f us  %7.2f TB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
}

int main()
{
    const long long n = 16LL * 3840 * 2160;   // floats per image stack of the 1080p batch
    float *a, *b, *c;
    CK(hipMalloc(&a, n * 4 * 5)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
    CK(hipMemset(a, 0, n * 4 * 5));
    const int grid = 256 * 16;
    timeit("copy float4 (r4+w4 B)", n * 8.0, [&] { k_copy4<<<grid, 256>>>((float4*)a, (float4*)b, n / 4); });
    timeit("read float4", n * 4.0, [&] { k_read4<<<grid, 256>>>((float4*)a, b, n / 4); });
    timeit("write float4", n * 4.0, [&] { k_write4<<<grid, 256>>>((float4*)b, n / 4); });
    timeit("r1 w2 float4 (12 B/px)", n * 12.0, [&] { k_r1w2<<<grid, 256>>>((float4*)a, (float4*)b, (float4*)c, n / 4); });
    timeit("r1 w2 dword (12 B/px)", n * 12.0, [&] { k_r1w2_dword<<<grid, 256>>>(a, b, c, n); });
    timeit("read 5 stacks dword (20 B/px)", n * 20.0, [&] { k_read5<<<grid, 256>>>(a, b, n, n); });
    timeit("write float4 nontemporal", n * 4.0, [&] { k_write4_nt<<<grid, 256>>>((float4*)b, n / 4); });
    timeit("r1 w2 float4 nt ld+st", n * 12.0, [&] { k_r1w2_nt<<<grid, 256>>>((float4*)a, (float4*)b, (float4*)c, n / 4); });
    timeit("r1 w2 float4 nt st only", n * 12.0, [&] { k_r1w2_ntst<<<grid, 256>>>((float4*)a, (float4*)b, (float4*)c, n / 4); });
    for (int pb : {1024, 4096, 16384})
        timeit("r1 w2 float4 chunked", n * 12.0, [&] { k_r1w2_chunk<<<(unsigned)((n / 4 + pb - 1) / pb), 256>>>((float4*)a, (float4*)b, (float4*)c, n / 4, pb); });
    for (int g : {1024, 2048, 8192, 32768})
        timeit("r1 w2 float4 grid sweep", n * 12.0, [&] { k_r1w2<<<g, 256>>>((float4*)a, (float4*)b, (float4*)c, n / 4); });
    {
        const int w = 3840, h = 2160, B = 16;
        for (int strip : {64, 128, 2160}) {
            { const int nbx = (w + 223) / 224, nby = (h + strip - 1) / strip;
              char nm[64]; snprintf(nm, 64, "tile 224 strip %d r1w2", strip);
              timeit(nm, n * 12.0, [&] { k_tile_r1w2<224><<<nbx * nby * B, 256>>>(a, b, c, w, h, strip, nbx, nby, 1); });
              snprintf(nm, 64, "tile 224 strip %d w2 only", strip);
              timeit(nm, n * 8.0, [&] { k_tile_r1w2<224><<<nbx * nby * B, 256>>>(a, b, c, w, h, strip, nbx, nby, 0); }); }
            { const int nbx = (w + 255) / 256, nby = (h + strip - 1) / strip;
              char nm[64]; snprintf(nm, 64, "tile 256 strip %d r1w2", strip);
              timeit(nm, n * 12.0, [&] { k_tile_r1w2<256><<<nbx * nby * B, 256>>>(a, b, c, w, h, strip, nbx, nby, 1); }); }
            { const int nbx = (w + 959) / 960, nby = (h + strip - 1) / strip;
              char nm[64]; snprintf(nm, 64, "tile 960 strip %d w2 only", strip);
              timeit(nm, n * 8.0, [&] { k_tile_r1w2<960><<<nbx * nby * B, 256>>>(a, b, c, w, h, strip, nbx, nby, 0); }); }
        }
    }
    return 0;
}
