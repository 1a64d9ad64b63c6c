// Practical HBM ceiling for a read-one-write-one stream on this GPU: a grid-stride copy with
// 16-byte accesses over buffers far larger than the caches, timed with events.
//   hipcc --offload-arch=gfx950 -O2 -o copy_bw tools/membench/copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

__global__ void __launch_bounds__(256) k_read(const v4f* __restrict__ in, float* __restrict__ out, size_t n)
{
    v4f s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}

int main()
{
    const size_t bytes = 4ull << 30, n = bytes / 16;
    v4f *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grids[] = {256 * 8, 256 * 16, 256 * 32, 256 * 64};
    for (int g : grids) {
        k_copy<<<g, 256>>>(a, b, n);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_copy<<<g, 256>>>(a, b, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("copy  grid %6d: %7.1f GB/s (read + write)\n", g, 5 * 2.0 * bytes / (ms * 1e-3) / 1e9);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_read<<<g, 256>>>(a, (float*)b, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("read  grid %6d: %7.1f GB/s\n", g, 5 * 1.0 * bytes / (ms * 1e-3) / 1e9);
    }
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D     : %7.1f GB/s (read + write)\n", 5 * 2.0 * bytes / (ms * 1e-3) / 1e9);
    return 0;
}
