// Practical HBM ceiling for a read-one-write-one stream on this GPU: copies with 16-byte
// accesses over buffers far larger than the caches, timed with events.  Variants: how many
// independent 16-byte loads a thread issues before its stores (bytes in flight per CU), plain or
// nontemporal accesses, block-contiguous or grid-stride assignment, grid size.
//   hipcc --offload-arch=gfx950 -O3 -o copy_bw tools/membench/copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

// Grid-stride: consecutive blocks touch consecutive 4 KiB pieces, U pieces in flight per thread.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
            else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

// Block-contiguous: every block owns one contiguous range (what an image tile kernel does).
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy_chunk(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[i + u * 256]) : in[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], &out[i + u * 256]);
            else out[i + u * 256] = v[u];
        }
    }
    for (; i < b1; i += 256) out[i] = in[i];
}

template <int U>
__global__ void __launch_bounds__(256) k_read(const v4f* __restrict__ in, float* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    v4f s = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) s += in[i + u * stride];
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}

template <int U>
__global__ void __launch_bounds__(256) k_write(v4f* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) out[i + u * stride] = v;
    }
}

static hipEvent_t e0, e1;
template <class F>
static double time_ms(F f, int reps)
{
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 4;
    const size_t bytes = gib << 30, n = bytes / 16;
    v4f *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("buffers: 2 x %zu GiB\n", gib);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64};
#define COPY(KERN, U, NT, label)                                                                       \
    for (int g : grids) {                                                                              \
        const double ms = time_ms([&] { KERN<U, NT><<<g, 256>>>(a, b, n); }, 5);                       \
        printf("%-22s U=%d grid %6d: %7.1f GB/s (read + write)\n", label, U, g, 2.0 * bytes / (ms * 1e-3) / 1e9); \
    }
    COPY(k_copy, 1, false, "copy stride")
    COPY(k_copy, 2, false, "copy stride")
    COPY(k_copy, 4, false, "copy stride")
    COPY(k_copy, 8, false, "copy stride")
    COPY(k_copy, 4, true, "copy stride nt")
    COPY(k_copy, 8, true, "copy stride nt")
    COPY(k_copy_chunk, 4, false, "copy chunk")
    COPY(k_copy_chunk, 8, false, "copy chunk")
    COPY(k_copy_chunk, 8, true, "copy chunk nt")
    for (int g : grids) {
        const double ms = time_ms([&] { k_read<4><<<g, 256>>>(a, (float*)b, n); }, 5);
        printf("%-22s U=4 grid %6d: %7.1f GB/s\n", "read", g, 1.0 * bytes / (ms * 1e-3) / 1e9);
    }
    for (int g : grids) {
        const double ms = time_ms([&] { k_write<4><<<g, 256>>>(b, n); }, 5);
        printf("%-22s U=4 grid %6d: %7.1f GB/s\n", "write", g, 1.0 * bytes / (ms * 1e-3) / 1e9);
    }
    {
        const double ms = time_ms([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("hipMemcpy D2D                       : %7.1f GB/s (read + write)\n", 2.0 * bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
