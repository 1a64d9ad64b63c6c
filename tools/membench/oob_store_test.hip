// Do buffer stores whose lane offset is at or above the descriptor's num_records get dropped?
// The whole target range lies inside one 3 GiB allocation, so a store that is not dropped shows up
// as a changed word instead of a fault.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* base, unsigned soff)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const unsigned lane = threadIdx.x;
    // in range: words 0..63
    __builtin_amdgcn_raw_buffer_store_b32(0x11111111u, r, lane * 4u, soff, 0);
    // lane offset with the top bit set: expected to be dropped
    __builtin_amdgcn_raw_buffer_store_b32(0x22222222u, r, 0x80000000u + lane * 4u, soff, 0);
    v4u v = {0x33333333u, 0x33333333u, 0x33333333u, 0x33333333u};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, 0x80000000u + 4096u + lane * 16u, soff, 0);
    // half of the lanes in range, half out
    __builtin_amdgcn_raw_buffer_store_b32(0x44444444u, r, (lane & 1 ? 0x80000000u : 0u) + 1024u + lane * 4u, soff, 0);
}
int main()
{
    const size_t bytes = 3ull << 30;
    unsigned* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (unsigned soff : {0u, 8192u}) {
        hipMemset(d, 0, bytes);
        k<<<1, 64>>>(d, soff);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<unsigned> lo(8192), hi(8192);
        hipMemcpy(lo.data(), (char*)d + soff, lo.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hi.data(), (char*)d + (2ull << 30) + soff, hi.size() * 4, hipMemcpyDeviceToHost);
        int in1 = 0, in4 = 0, hi_nonzero = 0;
        for (int i = 0; i < 64; i++) in1 += lo[i] == 0x11111111u;
        for (int i = 0; i < 64; i++) in4 += lo[256 + i] == 0x44444444u;
        for (unsigned x : hi) hi_nonzero += x != 0;
        printf("soffset %5u: in-range stores landed %d/64, mixed store landed %d (expect 32), words changed 2 GiB up: %d (expect 0)\n",
               soff, in1, in4, hi_nonzero);
    }
    return 0;
}
