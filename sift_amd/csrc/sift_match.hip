// Descriptor matching on gfx950: lib_matching.c:57-135 (all-pairs Euclidean distances, the
// two nearest neighbours of every query, threshold on the distance or on the distance ratio).
//
// Results are bit-identical to the scalar code:
//  * Descriptors produced by the pipeline hold integers 0..255, so the squared distance is an
//    integer below 2^24 and a float accumulation of it is exact in any order.  That path runs
//    on the matrix cores: |a-b|^2 = |a|^2 + |b|^2 - 2 a.b with a.b from v_mfma_i32_32x32x32_i8
//    on the bytes shifted by -128 (differences are unchanged by the shift).
//  * Arbitrary float descriptors take a second kernel that sums (a-b)^2 left to right in
//    float32 like lib_util.c:137-145.
//  * The distance is the correctly rounded float sqrt (lib_util.c:148-153 rounds a double sqrt
//    to float, which is the same value), and the two minima are the first two elements in the
//    order (distance, scan position) that the streaming search of lib_util.c:108-134 realises,
//    including its start-up rule that target 1 precedes target 0 when their distances tie.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sift_ctx.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            sa_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

static constexpr u64 KEY_NONE = ~0ull;

// (distance, target) packed so that unsigned comparison orders by distance then index;
// distances are non-negative floats, whose bit patterns order like the values.
__device__ __forceinline__ u64 match_key(float d, int t) { return ((u64)__float_as_uint(d) << 32) | (unsigned)t; }

// lib_util.c:151: a double sqrt rounded to float (= the correctly rounded float sqrt).
__device__ __forceinline__ float sqrt_as_reference(float d) { return (float)sqrt((double)d); }

// Integer path: squared distances below 2^22 have pairwise different float square roots, so
// they can stand in for the distance in comparisons; from 2^22 on (roots >= 2048, float
// spacing wider than the gap between consecutive roots) the root itself is used.  Its bit
// pattern (>= 0x45000000) is above every small code, so the mixed code stays monotone.
#define SA_SMALL_D2 (1 << 22)
__device__ __forceinline__ unsigned d2_code(int d2)
{
    return d2 < SA_SMALL_D2 ? (unsigned)d2 : __float_as_uint(sqrt_as_reference((float)d2));
}
__device__ __forceinline__ float d2_code_to_distance(unsigned code)
{
    return code < SA_SMALL_D2 ? sqrt_as_reference((float)code) : __uint_as_float(code);
}

__device__ __forceinline__ void top2_insert(u64& a, u64& b, u64 k)
{
    const u64 lo = k < a ? k : a, hi = k < a ? a : k;
    a = lo;
    b = hi < b ? hi : b;
}
__device__ __forceinline__ void top2_merge(u64& a, u64& b, u64 oa, u64 ob)
{
    const u64 lo = oa < a ? oa : a, hi = oa < a ? a : oa;
    const u64 m = ob < b ? ob : b;
    a = lo;
    b = hi < m ? hi : m;
}
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m)
{
    const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
    return ((u64)hi << 32) | lo;
}

// ---- packing -----------------------------------------------------------------------------
// One wave per row: bytes -> (value - 128) as int8, zero padded to dpad columns, plus the
// squared norm of the shifted row.  Rows n..n_pad-1 are zero.
__global__ void __launch_bounds__(64)
k_match_pack_u8(const unsigned char* __restrict__ src, long long stride, int n, int dim, int dpad,
                signed char* __restrict__ dst, int* __restrict__ norm)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    int s = 0;
    if (4 * lane < dpad) {
        unsigned packed = 0;
        for (int e = 0; e < 4; e++) {
            const int k = 4 * lane + e;
            int v = 0;
            if (row < n && k < dim) v = (int)src[(long long)row * stride + k] - 128;
            s += v * v;
            packed |= (unsigned)(v & 0xff) << (8 * e);
        }
        *reinterpret_cast<unsigned*>(dst + (long long)row * dpad + 4 * lane) = packed;
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) norm[row] = s;
}

// float rows (stride floats apart) -> transposed [dim][n_pad] so that lanes holding
// consecutive targets read consecutive addresses.
__global__ void __launch_bounds__(256)
k_match_transpose_f32(const float* __restrict__ src, long long stride, int n, int dim, int n_pad,
                      float* __restrict__ dst)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (t >= n_pad) return;
    dst[(long long)k * n_pad + t] = t < n ? src[(long long)t * stride + k] : 0.f;
}

// ---- distances + running two minima ------------------------------------------------------
// part[(q * nchunk + chunk) * 2 + {0,1}]: the two smallest keys of query q over the targets
// that block column `chunk` visited.

// Integer path.  Block = 4 waves sharing 32 queries; wave w of block column c visits target
// tiles (4c + w), (4c + w) + 4*nchunk, ...  The queries are the B operand, so a lane's 16
// accumulator registers all belong to query (lane & 31) and hold targets
// (reg&3) + 8*(reg>>2) + 4*(lane>>5) of the tile.  A and B fragments are read with the same
// byte -> k assignment, which is all a dot product needs.
template <int KS>
__global__ void __launch_bounds__(256)
k_match_i8(const signed char* __restrict__ Q, const int* __restrict__ nq,
           const signed char* __restrict__ T, const int* __restrict__ nt, int n2, int n2pad,
           u64* __restrict__ part, int nchunk)
{
    constexpr int DP = KS * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int q = blockIdx.x * 32 + r;
    v4i bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
        bq[ks] = *reinterpret_cast<const v4i*>(Q + (long long)q * DP + 32 * ks + 16 * h);
    const int myn = nq[q];
    u64 ka = KEY_NONE, kb = KEY_NONE;
    const int ntile = n2pad >> 5;
    for (int tile = blockIdx.y * 4 + wave; tile < ntile; tile += nchunk * 4) {
        const int t0 = tile * 32;
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const v4i a = *reinterpret_cast<const v4i*>(T + (long long)(t0 + r) * DP + 32 * ks + 16 * h);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const v4i n4 = *reinterpret_cast<const v4i*>(nt + t0 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int t = t0 + 8 * g + 4 * h + e;
                const int d2 = myn + n4[e] - 2 * acc[4 * g + e];
                const u64 key = t < n2 ? ((u64)d2_code(d2) << 32) | (unsigned)t : KEY_NONE;
                top2_insert(ka, kb, key);
            }
        }
    }
    top2_merge(ka, kb, shfl_xor_u64(ka, 32), shfl_xor_u64(kb, 32));
    __shared__ u64 red[4][32][2];
    if (h == 0) { red[wave][r][0] = ka; red[wave][r][1] = kb; }
    __syncthreads();
    if (wave == 0 && h == 0) {
        for (int w = 1; w < 4; w++) top2_merge(ka, kb, red[w][r][0], red[w][r][1]);
        u64* p = part + ((long long)q * nchunk + blockIdx.y) * 2;
        p[0] = ka; p[1] = kb;
    }
}

// lib_util.c:137-153 for one (query, target) pair, targets transposed.
__device__ __forceinline__ float dist_f32(const float* __restrict__ qrow, const float* __restrict__ TT,
                                          int n2pad, int dim, int t)
{
    float d = 0.f;
    for (int k = 0; k < dim; k++) {
        const float tt = __fsub_rn(qrow[k], TT[(long long)k * n2pad + t]);
        d = __fadd_rn(d, __fmul_rn(tt, tt));
    }
    return sqrt_as_reference(d);
}

// Float path.  One query per block; wave w of block column c visits 64-target tiles
// (4c + w), (4c + w) + 4*nchunk, ..., one target per lane.
__global__ void __launch_bounds__(256)
k_match_f32(const float* __restrict__ Q, long long q_stride, const float* __restrict__ TT, int n2, int n2pad,
            int dim, u64* __restrict__ part, int nchunk)
{
    __shared__ float sq[SA_MAX_DESCR];
    __shared__ u64 red[4][2];
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < dim; k += 256) sq[k] = Q[(long long)q * q_stride + k];
    __syncthreads();
    u64 ka = KEY_NONE, kb = KEY_NONE;
    const int ntile = n2pad >> 6;
    for (int tile = blockIdx.y * 4 + wave; tile < ntile; tile += nchunk * 4) {
        const int t = tile * 64 + lane;
        const float d = dist_f32(sq, TT, n2pad, dim, t);
        top2_insert(ka, kb, t < n2 ? match_key(d, t) : KEY_NONE);
    }
    for (int m = 32; m >= 1; m >>= 1) top2_merge(ka, kb, shfl_xor_u64(ka, m), shfl_xor_u64(kb, m));
    if (lane == 0) { red[wave][0] = ka; red[wave][1] = kb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) top2_merge(ka, kb, red[w][0], red[w][1]);
        u64* p = part + ((long long)q * nchunk + blockIdx.y) * 2;
        p[0] = ka; p[1] = kb;
    }
}

// ---- final merge, start-up rule, threshold -------------------------------------------------
struct MatchSrc {
    const signed char* Q8; const int* nq; const signed char* T8; const int* nt; int dpad;   // integer path
    const float* Qf; long long q_stride; const float* TT; int n2pad; int dim;               // float path
};

__global__ void __launch_bounds__(256)
k_match_final(const u64* __restrict__ part, int nchunk, int n1, MatchSrc s, float thresh, int flag,
              int* __restrict__ o_ia, int* __restrict__ o_ib, float* __restrict__ o_da, float* __restrict__ o_db,
              unsigned char* __restrict__ o_pass)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n1) return;
    u64 ka = KEY_NONE, kb = KEY_NONE;
    for (int c = 0; c < nchunk; c++) {
        const u64* p = part + ((long long)q * nchunk + c) * 2;
        top2_merge(ka, kb, p[0], p[1]);
    }
    int ia = (int)(unsigned)ka, ib = (int)(unsigned)kb;
    float da, db;
    if (s.Q8) { da = d2_code_to_distance((unsigned)(ka >> 32)); db = d2_code_to_distance((unsigned)(kb >> 32)); }
    else { da = __uint_as_float((unsigned)(ka >> 32)); db = __uint_as_float((unsigned)(kb >> 32)); }
    if (ia == 0 || ib == 0) {
        // lib_util.c:110-121 starts with (A,B) = (1,0) unless d[0] < d[1]: on a tie target 1
        // comes before target 0.
        float d0, d1;
        if (s.Q8) {
            int dot0 = 0, dot1 = 0;
            for (int k = 0; k < s.dpad; k++) {
                const int a = s.Q8[(long long)q * s.dpad + k];
                dot0 += a * s.T8[k];
                dot1 += a * s.T8[s.dpad + k];
            }
            d0 = sqrt_as_reference((float)(s.nq[q] + s.nt[0] - 2 * dot0));
            d1 = sqrt_as_reference((float)(s.nq[q] + s.nt[1] - 2 * dot1));
        } else {
            d0 = dist_f32(s.Qf + (long long)q * s.q_stride, s.TT, s.n2pad, s.dim, 0);
            d1 = dist_f32(s.Qf + (long long)q * s.q_stride, s.TT, s.n2pad, s.dim, 1);
        }
        if (d0 == d1) {
            if (ia == 0) { ia = 1; ib = 0; }
            else ib = 1;
        }
    }
    const float val = flag == 1 ? __fdiv_rn(da, db) : da;   // lib_matching.c:115
    o_ia[q] = ia; o_ib[q] = ib; o_da[q] = da; o_db[q] = db;
    o_pass[q] = val < thresh ? 1 : 0;
}

// ---- host side ---------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { p = nullptr; sa_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); return -1; }
        return 0;
    }
    template <class T> T* as() const { return (T*)p; }
};

int pad_to(int n, int m) { return (n + m - 1) / m * m; }

int chunks_for(int row_blocks, int tiles)
{
    // enough blocks to cover the 256 CUs a few times over, at most one block column per 4 tiles
    int c = (2048 + row_blocks - 1) / row_blocks;
    const int most = (tiles + 3) / 4;
    if (c > most) c = most;
    return c < 1 ? 1 : c;
}

template <int KS>
void launch_i8(hipStream_t st, dim3 grid, const signed char* Q, const int* nq, const signed char* T, const int* nt,
               int n2, int n2pad, u64* part, int nchunk)
{
    hipLaunchKernelGGL(k_match_i8<KS>, grid, dim3(256), 0, st, Q, nq, T, nt, n2, n2pad, part, nchunk);
}

// Sizes of the integer path's scratch: packed int8 rows and squared norms of both sets, and the
// partial minima of every (query, block column).
struct U8Plan {
    int dpad, n1pad, n2pad, nchunk;
    size_t q8_bytes, t8_bytes, nq_bytes, nt_bytes, part_bytes;
};

U8Plan plan_u8(int n1, int n2, int dim)
{
    U8Plan pl;
    pl.dpad = pad_to(dim, 32); pl.n1pad = pad_to(n1, 32); pl.n2pad = pad_to(n2, 32);
    pl.nchunk = chunks_for(pl.n1pad / 32, pl.n2pad / 32);
    pl.q8_bytes = (size_t)pl.n1pad * pl.dpad; pl.t8_bytes = (size_t)pl.n2pad * pl.dpad;
    pl.nq_bytes = pl.n1pad * sizeof(int); pl.nt_bytes = pl.n2pad * sizeof(int);
    pl.part_bytes = (size_t)pl.n1pad * pl.nchunk * 2 * sizeof(u64);
    return pl;
}

// Queues the whole integer path on `st`: pack both sets, distances + partial minima, final merge.
// Every byte of the scratch regions that a kernel reads is written earlier in this sequence
// (padding rows and columns included), so the regions need no clearing between calls.
void run_u8(hipStream_t st, const U8Plan& pl, const unsigned char* d1, long long stride1, int n1,
            const unsigned char* d2, long long stride2, int n2, int dim, float thresh, int flag,
            signed char* q8, int* nq, signed char* t8, int* nt, u64* part,
            int* o_ia, int* o_ib, float* o_da, float* o_db, unsigned char* o_pass)
{
    hipLaunchKernelGGL(k_match_pack_u8, dim3(pl.n1pad), dim3(64), 0, st, d1, stride1, n1, dim, pl.dpad, q8, nq);
    hipLaunchKernelGGL(k_match_pack_u8, dim3(pl.n2pad), dim3(64), 0, st, d2, stride2, n2, dim, pl.dpad, t8, nt);
    const dim3 grid(pl.n1pad / 32, pl.nchunk);
#define SA_MATCH_CASE(K) \
    case K: launch_i8<K>(st, grid, q8, nq, t8, nt, n2, pl.n2pad, part, pl.nchunk); break;
    switch (pl.dpad / 32) {
        SA_MATCH_CASE(1) SA_MATCH_CASE(2) SA_MATCH_CASE(3) SA_MATCH_CASE(4)
        SA_MATCH_CASE(5) SA_MATCH_CASE(6) SA_MATCH_CASE(7) SA_MATCH_CASE(8)
    }
#undef SA_MATCH_CASE
    MatchSrc src;
    memset(&src, 0, sizeof(src));
    src.Q8 = q8; src.nq = nq; src.T8 = t8; src.nt = nt; src.dpad = pl.dpad;
    hipLaunchKernelGGL(k_match_final, dim3((n1 + 255) / 256), dim3(256), 0, st, part, pl.nchunk, n1, src, thresh,
                       flag, o_ia, o_ib, o_da, o_db, o_pass);
}

// d1/d2: device rows.  is_u8: bytes with row strides in bytes, else floats with strides in floats.
int match_device(hipStream_t st, bool is_u8, const void* d1, long long stride1, int n1, const void* d2, long long stride2,
                 int n2, int dim, float thresh, int flag, int* iA, int* iB, float* dA, float* dB, unsigned char* pass)
{
    if (n1 < 0) { sa_set_error("negative number of query descriptors (%d)", n1); return -1; }
    if (n2 < 2) { sa_set_error("matching needs at least two target descriptors (got %d)", n2); return -1; }
    if (dim < 1 || dim > SA_MAX_DESCR) { sa_set_error("descriptor length %d outside 1..%d", dim, SA_MAX_DESCR); return -1; }
    if (n1 == 0) return 0;
    DevBuf q8, t8, nq, nt, tt, part, o_ia, o_ib, o_da, o_db, o_pass;
    if (o_ia.alloc(n1 * sizeof(int)) || o_ib.alloc(n1 * sizeof(int)) || o_da.alloc(n1 * sizeof(float)) ||
        o_db.alloc(n1 * sizeof(float)) || o_pass.alloc(n1))
        return -1;
    if (is_u8) {
        const U8Plan pl = plan_u8(n1, n2, dim);
        if (q8.alloc(pl.q8_bytes) || t8.alloc(pl.t8_bytes) || nq.alloc(pl.nq_bytes) || nt.alloc(pl.nt_bytes) ||
            part.alloc(pl.part_bytes))
            return -1;
        run_u8(st, pl, (const unsigned char*)d1, stride1, n1, (const unsigned char*)d2, stride2, n2, dim, thresh, flag,
               q8.as<signed char>(), nq.as<int>(), t8.as<signed char>(), nt.as<int>(), part.as<u64>(),
               o_ia.as<int>(), o_ib.as<int>(), o_da.as<float>(), o_db.as<float>(), o_pass.as<unsigned char>());
    } else {
        MatchSrc src;
        memset(&src, 0, sizeof(src));
        const int n2pad = pad_to(n2, 64);
        if (tt.alloc((size_t)n2pad * dim * sizeof(float))) return -1;
        hipLaunchKernelGGL(k_match_transpose_f32, dim3((n2pad + 255) / 256, dim), dim3(256), 0, st, (const float*)d2,
                           stride2, n2, dim, n2pad, tt.as<float>());
        const int nchunk = chunks_for(n1, n2pad / 64);
        if (part.alloc((size_t)n1 * nchunk * 2 * sizeof(u64))) return -1;
        hipLaunchKernelGGL(k_match_f32, dim3(n1, nchunk), dim3(256), 0, st, (const float*)d1, stride1, tt.as<float>(), n2,
                           n2pad, dim, part.as<u64>(), nchunk);
        src.Qf = (const float*)d1; src.q_stride = stride1; src.TT = tt.as<float>(); src.n2pad = n2pad; src.dim = dim;
        hipLaunchKernelGGL(k_match_final, dim3((n1 + 255) / 256), dim3(256), 0, st, part.as<u64>(), nchunk, n1, src, thresh,
                           flag, o_ia.as<int>(), o_ib.as<int>(), o_da.as<float>(), o_db.as<float>(), o_pass.as<unsigned char>());
    }
    HIP_TRY(hipGetLastError());
    std::vector<unsigned char> hp(n1);
    HIP_TRY(hipMemcpyAsync(iA, o_ia.p, n1 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(iB, o_ib.p, n1 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dA, o_da.p, n1 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(dB, o_db.p, n1 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hp.data(), o_pass.p, n1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int m = 0;
    for (int i = 0; i < n1; i++) { m += hp[i]; if (pass) pass[i] = hp[i]; }
    return m;
}

int match_host(int device, bool is_u8, const void* h1, int n1, const void* h2, int n2, int dim, float thresh, int flag,
               int* iA, int* iB, float* dA, float* dB, unsigned char* pass)
{
    if (n1 < 0) { sa_set_error("negative number of query descriptors (%d)", n1); return -1; }
    if (n2 < 2) { sa_set_error("matching needs at least two target descriptors (got %d)", n2); return -1; }
    if (dim < 1 || dim > SA_MAX_DESCR) { sa_set_error("descriptor length %d outside 1..%d", dim, SA_MAX_DESCR); return -1; }
    if (n1 == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    const size_t el = is_u8 ? 1 : sizeof(float);
    DevBuf a, b;
    if (a.alloc((size_t)n1 * dim * el) || b.alloc((size_t)n2 * dim * el)) return -1;
    // a stream of the call's own, not the legacy default stream (whose implicit synchronisation
    // fails while another host thread captures a hipGraph)
    struct LocalStream {
        hipStream_t s = nullptr;
        ~LocalStream() { if (s) (void)hipStreamDestroy(s); }
    } ls;
    HIP_TRY(hipStreamCreateWithFlags(&ls.s, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(a.p, h1, (size_t)n1 * dim * el, hipMemcpyHostToDevice, ls.s));
    HIP_TRY(hipMemcpyAsync(b.p, h2, (size_t)n2 * dim * el, hipMemcpyHostToDevice, ls.s));
    return match_device(ls.s, is_u8, a.p, dim, n1, b.p, dim, n2, dim, thresh, flag, iA, iB, dA, dB, pass);
}

}  // namespace

extern "C" int sift_amd_match_u8(int device, const unsigned char* desc1, int n1, const unsigned char* desc2, int n2,
                                 int dim, float thresh, int flag, int* iA, int* iB, float* dA, float* dB,
                                 unsigned char* pass)
{
    return match_host(device, true, desc1, n1, desc2, n2, dim, thresh, flag, iA, iB, dA, dB, pass);
}

extern "C" int sift_amd_match_f32(int device, const float* desc1, int n1, const float* desc2, int n2, int dim,
                                  float thresh, int flag, int* iA, int* iB, float* dA, float* dB, unsigned char* pass)
{
    return match_host(device, false, desc1, n1, desc2, n2, dim, thresh, flag, iA, iB, dA, dB, pass);
}

extern "C" int sift_amd_match_images(sift_amd_ctx* c, int img_a, int img_b, float thresh, int flag, int* iA, int* iB,
                                     float* dA, float* dB, unsigned char* pass, int cap)
{
    const void *std_a, *std_b;
    const float *fa, *fb;
    int na, nb, nd;
    if (sa_ctx_image_keys(c, img_a, &std_a, &fa, &na, &nd) || sa_ctx_image_keys(c, img_b, &std_b, &fb, &nb, &nd)) return -1;
    if (na > cap) { sa_set_error("output capacity %d below the %d keypoints of image %d", cap, na, img_a); return -1; }
    hipStream_t st = sa_ctx_stream(c);
    if (nd == 128)   // the 144-byte records carry the descriptor as bytes at offset 16
        return match_device(st, true, (const char*)std_a + 16, sizeof(SaStd), na, (const char*)std_b + 16, sizeof(SaStd),
                            nb, nd, thresh, flag, iA, iB, dA, dB, pass);
    return match_device(st, false, fa, SA_MAX_DESCR, na, fb, SA_MAX_DESCR, nb, nd, thresh, flag, iA, iB, dA, dB, pass);
}

// ---- allocation-free variant ---------------------------------------------------------------
// sift_amd_match_u8 for callers that cannot allocate (code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): descriptors and results stay in device memory that the
// caller owns, and the scratch of the integer path lives in one fixed array of the code object
// instead of hipMalloc'd buffers.  Each GPU holds its own copy of that array until the process
// ends; its device address is looked up once per device.
#define SA_NA_MAX_N 8192      // descriptors per set
#define SA_NA_MAX_DIM 256     // descriptor length
namespace {
constexpr size_t na_align(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t NA_ROWS_BYTES = na_align((size_t)SA_NA_MAX_N * SA_NA_MAX_DIM);          // int8 rows of one set
constexpr size_t NA_NORM_BYTES = na_align((size_t)SA_NA_MAX_N * sizeof(int));
// n1pad * nchunk <= 32 * (2048 + n1pad / 32 - 1) < 65536 + n1pad entries of two keys
constexpr size_t NA_PART_BYTES = na_align((size_t)(65536 + SA_NA_MAX_N) * 2 * sizeof(u64));
constexpr size_t NA_OFF_Q8 = 0, NA_OFF_T8 = NA_OFF_Q8 + NA_ROWS_BYTES, NA_OFF_NQ = NA_OFF_T8 + NA_ROWS_BYTES,
                 NA_OFF_NT = NA_OFF_NQ + NA_NORM_BYTES, NA_OFF_PART = NA_OFF_NT + NA_NORM_BYTES,
                 NA_TOTAL = NA_OFF_PART + NA_PART_BYTES;                                 // 5.3 MB
constexpr int NA_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[NA_TOTAL];
// device address of `scratch` on each GPU (the array's name is only a host-side handle)
static unsigned char* g_na_ptr[NA_MAX_DEVICES];

extern "C" int sift_amd_match_u8_noalloc(const unsigned char* d_desc1, int n1, const unsigned char* d_desc2, int n2,
                                         int dim, float thresh, int flag, int* d_iA, int* d_iB, float* d_dA,
                                         float* d_dB, unsigned char* d_pass, void* stream)
{
    if (n1 < 0) { sa_set_error("negative number of query descriptors (%d)", n1); return -1; }
    if (n2 < 2) { sa_set_error("matching needs at least two target descriptors (got %d)", n2); return -1; }
    if (dim < 1 || dim > SA_NA_MAX_DIM) { sa_set_error("descriptor length %d outside 1..%d", dim, SA_NA_MAX_DIM); return -1; }
    if (n1 > SA_NA_MAX_N || n2 > SA_NA_MAX_N) {
        sa_set_error("allocation-free matching takes at most %d descriptors per set (got %d and %d)", SA_NA_MAX_N, n1, n2);
        return -1;
    }
    if (n1 == 0) return 0;
    const U8Plan pl = plan_u8(n1, n2, dim);
    if (pl.q8_bytes > NA_ROWS_BYTES || pl.t8_bytes > NA_ROWS_BYTES || pl.nq_bytes > NA_NORM_BYTES ||
        pl.nt_bytes > NA_NORM_BYTES || pl.part_bytes > NA_PART_BYTES) {
        sa_set_error("allocation-free matching: %d x %d descriptors of length %d exceed the fixed scratch", n1, n2, dim);
        return -1;
    }
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= NA_MAX_DEVICES) { sa_set_error("device index %d outside 0..%d", dev, NA_MAX_DEVICES - 1); return -1; }
    unsigned char* base = __atomic_load_n(&g_na_ptr[dev], __ATOMIC_ACQUIRE);
    if (!base) {
        HIP_TRY(hipGetSymbolAddress((void**)&base, HIP_SYMBOL(scratch)));
        if (!base) { sa_set_error("hipGetSymbolAddress returned no address for the scratch array"); return -1; }
        __atomic_store_n(&g_na_ptr[dev], base, __ATOMIC_RELEASE);
    }
    run_u8((hipStream_t)stream, pl, d_desc1, dim, n1, d_desc2, dim, n2, dim, thresh, flag,
           (signed char*)(base + NA_OFF_Q8), (int*)(base + NA_OFF_NQ), (signed char*)(base + NA_OFF_T8),
           (int*)(base + NA_OFF_NT), (u64*)(base + NA_OFF_PART), d_iA, d_iB, d_dA, d_dB, d_pass);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- reference-named entry points (lib_matching.h) -----------------------------------------

static void match_fatal(const char* what)
{
    fprintf(stderr, "sift_amd: %s: %s\n", what, sift_amd_last_error());
    abort();
}

static int env_device()
{
    const char* e = getenv("SIFT_AMD_DEVICE");
    return e ? atoi(e) : 0;
}

// lib_matching.c:92-135
extern "C" void matching(struct sift_keypoints* k1, struct sift_keypoints* k2, struct sift_keypoints* out_k1,
                         struct sift_keypoints* out_k2A, struct sift_keypoints* out_k2B, float thresh, int flag)
{
    const int n1 = k1->size, n2 = k2->size;
    if (n1 == 0) return;
    const int dim = k1->list[0]->n_hist * k1->list[0]->n_hist * k1->list[0]->n_ori;
    // integer-valued descriptors (what sift_anatomy produces) go to the byte path
    bool bytes = true;
    std::vector<float> f1((size_t)n1 * dim), f2((size_t)n2 * dim);
    for (int pass = 0; pass < 2; pass++) {
        const struct sift_keypoints* k = pass ? k2 : k1;
        std::vector<float>& f = pass ? f2 : f1;
        for (int i = 0; i < k->size; i++) {
            const float* d = k->list[i]->descr;
            for (int e = 0; e < dim; e++) {
                const float v = d[e];
                f[(size_t)i * dim + e] = v;
                if (!(v >= 0.f && v <= 255.f && v == (float)(int)v)) bytes = false;
            }
        }
    }
    std::vector<int> iA(n1), iB(n1);
    std::vector<float> dA(n1), dB(n1);
    std::vector<unsigned char> pass(n1);
    int m;
    if (bytes) {
        std::vector<unsigned char> b1(f1.size()), b2(f2.size());
        for (size_t i = 0; i < f1.size(); i++) b1[i] = (unsigned char)f1[i];
        for (size_t i = 0; i < f2.size(); i++) b2[i] = (unsigned char)f2[i];
        m = sift_amd_match_u8(env_device(), b1.data(), n1, b2.data(), n2, dim, thresh, flag, iA.data(), iB.data(),
                              dA.data(), dB.data(), pass.data());
    } else {
        m = sift_amd_match_f32(env_device(), f1.data(), n1, f2.data(), n2, dim, thresh, flag, iA.data(), iB.data(),
                               dA.data(), dB.data(), pass.data());
    }
    if (m < 0) match_fatal("matching");
    for (int i = 0; i < n1; i++) {
        if (!pass[i]) continue;
        sift_add_keypoint_to_list(sift_malloc_keypoint_from_model_and_copy(k1->list[i]), out_k1);
        sift_add_keypoint_to_list(sift_malloc_keypoint_from_model_and_copy(k2->list[iA[i]]), out_k2A);
        sift_add_keypoint_to_list(sift_malloc_keypoint_from_model_and_copy(k2->list[iB[i]]), out_k2B);
    }
}

// lib_matching.c:137-148
extern "C" void print_pairs(const struct sift_keypoints* k1, const struct sift_keypoints* k2)
{
    for (int i = 0; i < k1->size; i++) {
        fprintf_one_keypoint(stdout, k1->list[i], 0, 0, 0);
        fprintf_one_keypoint(stdout, k2->list[i], 0, 0, 0);
        fprintf(stdout, "\n");
    }
}

// lib_matching.c:150-172
extern "C" void save_pairs_extra(const char* name, const struct sift_keypoints* k1, const struct sift_keypoints* k2A,
                                 const struct sift_keypoints* k2B)
{
    FILE* f = fopen(name, "w");
    if (!f) { fprintf(stderr, "sift_amd: cannot open %s for writing\n", name); abort(); }
    if (k1->size > 0) {
        const struct keypoint* k0 = k1->list[0];
        const int dim = k0->n_hist * k0->n_hist * k0->n_ori, n_bins = k0->n_bins;
        for (int i = 0; i < k1->size; i++) {
            fprintf_one_keypoint(f, k1->list[i], dim, n_bins, 2);
            fprintf_one_keypoint(f, k2A->list[i], dim, n_bins, 2);
            fprintf_one_keypoint(f, k2B->list[i], dim, n_bins, 2);
            fprintf(f, "\n");
        }
    }
    fclose(f);
}
