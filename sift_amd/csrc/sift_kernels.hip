// HIP kernels of the SIFT hot path for gfx950 (MI355X), wave64.
//
// Everything that feeds the DoG extrema decision is computed with the same float32
// operation order as the reference C code and must be compiled with -ffp-contract=off:
// one fused multiply-add in a blur changes last bits of the DoG and with them the set of
// discrete extrema.  Histogram accumulation (orientation, descriptor) is done in raster
// order per bin, exactly as a scalar loop would, so results do not depend on scheduling.
//
// Axis convention (the reference's): x = row index i (down), y = column index j (right),
// images are row-major im[i*w + j].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sift_dev.h"
#include "sift_launch.h"

// Native 4-float vector: 16-byte LDS/global accesses compile to single b128 / dwordx4
// instructions (HIP's float4 struct gets split into narrower accesses).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f ld4(const float* p) { return *reinterpret_cast<const v4f*>(__builtin_assume_aligned(p, 16)); }
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d)
{
    v4f v = {a, b, c, d};
    *reinterpret_cast<v4f*>(__builtin_assume_aligned(p, 16)) = v;
}

// Buffer addressing for the fused blur: a wave-uniform resource (base of one image) and a
// wave-uniform byte offset (row) live in scalar registers, the lane supplies a 32-bit byte
// offset, so an access costs no vector address arithmetic.  Images are below 2 GiB
// (checked when a context is created).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, row_bytes, 0));
}
__device__ __forceinline__ v2f buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, row_bytes, 0));
}
__device__ __forceinline__ void buf_st1(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, lane_bytes, row_bytes, 0);
}
__device__ __forceinline__ void buf_st2(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes, v2f v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, lane_bytes, row_bytes, 0);
}
// A 16-byte buffer store reads its data registers for a few cycles after it has issued.  The
// compiler only separates such a store from a following write to those registers when the store has
// no scalar offset register; with one (as here) it may place the overwrite directly behind the
// store, and on gfx950 some lanes then store the new value (seen as wrong samples in every fourth
// column).  The asm reads the four registers after the store, so nothing may overwrite them before
// its wait states have passed; tests/test_abi.py scans the disassembly for the pattern.
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned row_bytes, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, lane_bytes, row_bytes, 0);
    asm volatile("s_nop 1" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
}

#define SA_PI 3.14159265358979323846
#define SA_SQRT2 1.41421356237309504880

// ------------------------------------------------------------------ helpers --

// Half-sample symmetric extension with period 2l (lib_discrete.c:243-249), made total
// for any i.
__device__ __forceinline__ int mirror_idx(int i, int l)
{
    if ((unsigned)i < (unsigned)l) return i;
    int ll = 2 * l;
    i %= ll;
    if (i < 0) i += ll;
    if (i > l - 1) i = ll - 1 - i;
    return i;
}

// Blocks are dealt round-robin to the 8 XCDs (each with its own L2) in launch order.  For
// kernels whose neighbouring tiles share halo data, remap the linear block id so that every
// XCD works on one contiguous range of tiles: blocks l and l+8 (same XCD) become neighbours.
// Bijective for any block count; affects speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned l, unsigned total)
{
    const unsigned q = total >> 3, r = total & 7u, x = l & 7u;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (l >> 3);
}

// lane i receives the value of lane i-1 / i+1 of the wave (edge lanes receive 0).
__device__ __forceinline__ float lane_prev(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_next(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xf, 0xf, true));
}

// 64-bit value of lane i-1 (lane 0 receives 0)
__device__ __forceinline__ unsigned long long lane_prev_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, 0x138, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
// Inclusive prefix sum over the 64 lanes of a fully active wave: shifts inside the rows of 16
// lanes, then the row totals are passed on (row_bcast15 to rows 1 and 3, row_bcast31 to rows 2, 3).
__device__ __forceinline__ int wave_scan_inclusive(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    unsigned lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------- bilinear oversample --
// lib_discrete.c:207-237.  One thread per output sample.
__global__ void __launch_bounds__(256)
k_oversample(const float* __restrict__ in, int wi, int hi, long long s_in,
             float* __restrict__ out, int wo, int ho, long long s_out, float delta)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= wo) return;
    const float* src = in + (long long)blockIdx.z * s_in;
    float x = i * delta, y = j * delta;
    int im = (int)x, jm = (int)y;
    int ip = im + 1, jp = jm + 1;
    if (ip >= hi) ip = 2 * hi - 1 - ip;
    if (im >= hi) im = 2 * hi - 1 - im;
    if (jp >= wi) jp = 2 * wi - 1 - jp;
    if (jm >= wi) jm = 2 * wi - 1 - jm;
    const float fx = x - floorf(x);
    const float fy = y - floorf(y);
    const float a = src[(long long)ip * wi + jp], b = src[(long long)ip * wi + jm];
    const float c = src[(long long)im * wi + jp], d = src[(long long)im * wi + jm];
    const float top = __fadd_rn(__fmul_rn(fy, a), __fmul_rn(1 - fy, b));
    const float bot = __fadd_rn(__fmul_rn(fy, c), __fmul_rn(1 - fy, d));
    out[(long long)blockIdx.z * s_out + (long long)i * wo + j] =
        __fadd_rn(__fmul_rn(fx, top), __fmul_rn(1 - fx, bot));
}

// 8-bit pixels -> float gray / 256: io_png.c:480-520 (BT.709 weights applied in double on
// the float channel values, rounded to float once) followed by sift_cli.c:280-281.  Four
// pixels per thread.
template <int CH>
__global__ void __launch_bounds__(256)
k_ingest_u8(const unsigned char* __restrict__ in, float* __restrict__ out, long long n_px)
{
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= n_px) return;
    unsigned char b[4 * CH];
    if (p0 + 4 <= n_px) {
        const unsigned* w = reinterpret_cast<const unsigned*>(in + p0 * CH);   // 4*CH bytes, 4-byte aligned
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const unsigned v = w[k];
            b[4 * k] = v & 0xff; b[4 * k + 1] = (v >> 8) & 0xff; b[4 * k + 2] = (v >> 16) & 0xff; b[4 * k + 3] = v >> 24;
        }
    } else {
        for (int k = 0; k < 4 * CH; k++) b[k] = p0 * CH + k < n_px * CH ? in[p0 * CH + k] : 0;
    }
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (CH == 1) {
            g[e] = (float)b[e];
        } else {
            const double r = (double)(float)b[e * CH], gg = (double)(float)b[e * CH + 1], bl = (double)(float)b[e * CH + 2];
            g[e] = (float)__dadd_rn(__dadd_rn(__dmul_rn(0.212639005871510, r), __dmul_rn(0.715168678767756, gg)),
                                    __dmul_rn(0.072192315360734, bl));
        }
        g[e] = (float)((double)g[e] / 256.0);
    }
    if (p0 + 4 <= n_px) st4(out + p0, g[0], g[1], g[2], g[3]);
    else for (int e = 0; p0 + e < n_px; e++) out[p0 + e] = g[e];
}

// Zeroes n 32-bit words (n a multiple of 4, the buffer 16-byte aligned).  The pipeline clears its
// marks and counters with this kernel instead of hipMemsetAsync: a memset node of a captured
// hipGraph was seen to replay with the arguments of an unrelated later fill/copy of the process
// (ROCm 7.2: the counters of a context held another copy's operands after a graph replay that
// followed small host<->device copies on the null stream), a kernel node keeps its own.
__global__ void __launch_bounds__(256) k_clear_words(unsigned* __restrict__ p, long long n)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < n) *reinterpret_cast<v4u*>(__builtin_assume_aligned(p + i, 16)) = v4u{0u, 0u, 0u, 0u};
}

// lib_discrete.c:180-191
__global__ void __launch_bounds__(256)
k_subsample2(const float* __restrict__ in, int wi, long long s_in,
             float* __restrict__ out, int wo, int ho, long long s_out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= wo) return;
    out[(long long)blockIdx.z * s_out + (long long)i * wo + j] =
        in[(long long)blockIdx.z * s_in + (long long)(2 * i) * wi + 2 * j];
}

// ------------------------------------------------------------ Gaussian blur --
// lib_discrete.c:278-308: pass 1 along i (column direction), pass 2 along j.
// Per output:  sum = centre*g0;  sum += g[k]*(left_k + right_k)  for k = 1..R.

// Both passes in one kernel.  A block owns 224 output columns (256 with the 16-column halo on
// each side) and walks down a strip of rows, 8 rows per iteration:
//   - column pass: thread t keeps the 2R+8 newest samples of extended column t in registers (every
//     source sample is loaded once per strip, lanes of a wave read consecutive addresses) and
//     produces 8 column-filtered samples, which go to LDS together with the 8 unfiltered centre
//     samples (needed for the DoG);
//   - row pass: thread (row g, group q) reads 36 consecutive column-filtered samples of row g
//     from LDS and produces 4 adjacent outputs, stored with 16-byte accesses (448 such items
//     per iteration, two per thread).
// Per pixel the kernel reads the source image once (plus the strip/tile halo, served by L2)
// and writes gss[s] and dog[s-1]; the column-pass image never exists in memory.  Operation
// order per output is that of the reference's two passes, so results are bit-identical.
// `flags`, if given, gets a mark per 8-row x 64-column cell of the image in which some DoG
// sample exceeds flag_thr in magnitude (cells [image][row/8][col/64]); the extrema scan skips
// unmarked cells.
// `sub`, if given, receives the even samples (the next octave's first image,
// lib_discrete.c:180-191).  `dog` may be null: the pipeline forms DoG values from the Gaussian
// stack where it needs them and stores DoG images only for callers that download them.
// Source sample (i, col) of the fused blur when the source is the bilinearly oversampled
// input (OVS): same expression and operation order as k_oversample / lib_discrete.c:207-237.
struct OvsCol { int jm, jp; float fy; };
__device__ __forceinline__ float ovs_sample(__amdgpu_buffer_rsrc_t img, int wi, int hi, float delta,
                                            int i, const OvsCol& c)
{
    const float x = i * delta;
    int im = (int)x, ip = im + 1;
    if (ip >= hi) ip = 2 * hi - 1 - ip;
    if (im >= hi) im = 2 * hi - 1 - im;
    const float fx = x - floorf(x);
    // the row is the same for the whole wave (a thread is a column), but it comes out of float
    // arithmetic, which runs on the vector unit: name the row offsets wave-uniform, or every load
    // gets a loop over the distinct values of its scalar offset
    const unsigned rp = __builtin_amdgcn_readfirstlane((unsigned)(ip * wi) * 4u);
    const unsigned rm = __builtin_amdgcn_readfirstlane((unsigned)(im * wi) * 4u);
    const float a = buf_ld(img, c.jp * 4u, rp), b = buf_ld(img, c.jm * 4u, rp);
    const float cc = buf_ld(img, c.jp * 4u, rm), d = buf_ld(img, c.jm * 4u, rm);
    const float top = __fadd_rn(__fmul_rn(c.fy, a), __fmul_rn(1 - c.fy, b));
    const float bot = __fadd_rn(__fmul_rn(c.fy, cc), __fmul_rn(1 - c.fy, d));
    return __fadd_rn(__fmul_rn(fx, top), __fmul_rn(1 - fx, bot));
}

// Sum of the window pairs that start at positions a and b (same parity; compile-time constants
// after unrolling) of a window held as register pairs p[i] = {win[2i], win[2i+1]}.  Aligned
// positions take one packed add; odd positions take two plain adds on the halves, which costs
// the same issue time on gfx950 (a packed f32 instruction occupies the VALU about twice as long
// as a plain one) and needs no register moves to line the operands up.
__device__ __forceinline__ v2f win_pair_sum(const v2f* p, int a, int b)
{
    if (!(a & 1)) return p[a / 2] + p[b / 2];
    v2f t;
    t.x = __fadd_rn(p[(a - 1) / 2].y, p[(b - 1) / 2].y);
    t.y = __fadd_rn(p[(a + 1) / 2].x, p[(b + 1) / 2].x);
    return t;
}
// the pair starting at position a
__device__ __forceinline__ v2f win_pair(const v2f* p, int a)
{
    if (!(a & 1)) return p[a / 2];
    return v2f{p[(a - 1) / 2].y, p[(a + 1) / 2].x};
}
// Taps duplicated into pairs: packed multiplies take them straight from SGPR pairs.
struct SaTapPairs { v2f g[SA_MAX_RADIUS_FAST + 1]; };
struct SaTrue { static constexpr bool value = true; };
struct SaFalse { static constexpr bool value = false; };

#ifdef SA_BLUR_CLOCK
// Diagnostic build only: shader-clock ticks and 100 MHz real-time ticks spent inside the blur
// blocks, per radius, to read the clock the GPU holds while the kernel runs.
__device__ unsigned long long sa_blur_clock[2 * (SA_MAX_RADIUS_FAST + 1)];
extern "C" void sift_amd_debug_blur_clock(unsigned long long* out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_blur_clock), sizeof(sa_blur_clock));
    if (reset) { unsigned long long z[2 * (SA_MAX_RADIUS_FAST + 1)] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(sa_blur_clock), z, sizeof(z)); }
}
#endif

// OVS = true: `in` is the input image (wi x hi) and the blurred source is its oversampled
// version at step `delta`, evaluated on the fly (w x h is the oversampled size).

// PREV = false: no DoG epilogue (dog and flags are null), the buffer of unfiltered centre samples
// does not exist and a block needs half the LDS.
// SPARSE: the image is computed only where it can be read.  The last image of an octave's stack
// feeds nothing but the last DoG image, which the extrema scan and the refinement read only next to
// a sample that passed the first contrast test (lib_sift_anatomy.c:240-313, 358-405: a scan
// neighbour is 1 sample away, a refined position at most 4 moves plus one tap, so at most 6), and
// the marks of the 8 x 64 cells holding such samples (`need_cells`) are complete before this launch
// starts.  A block evaluates an 8-row group of its tile only if a cell within 6 samples of it is
// marked, and within such a group only the 64-column chunks with such a cell (a wave of the row pass
// owns a chunk x 4 rows, a wave of the column pass skips its 64 columns when no evaluated chunk reads
// them); the other samples of `out` keep whatever they held.  Strips have at most 64 groups.
template <int R, bool OVS, int EW, int NBUF, bool VW, bool PREV, bool SPARSE = false>
__global__ void __launch_bounds__(EW)
k_blur_fused(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ dog,
             float* __restrict__ sub, int w, int h, long long s_in, long long s_out,
             long long s_dog, long long s_sub, SaTapPairs tp, int strip, int nbx, int nby, int nblocks,
             int wi, int hi, float delta, int* __restrict__ flags, float flag_thr,
             const int* __restrict__ need_cells)
{
    static_assert(!SPARSE || (!OVS && !PREV && VW && NBUF == 2), "sparse form: plain launches only");
    // LDS rows are padded to 288 floats: with 16-byte reads at a 16-byte lane stride this makes
    // the row pass free of bank conflicts also where a wave straddles two rows.
    constexpr int G = 8, RP = 16, TW = EW - 2 * RP, NG = TW / 4, PITCH = EW + 32;
    constexpr int ITEMS = G * NG;
#ifdef SA_BLUR_CLOCK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ __attribute__((aligned(16))) float l_tmp[NBUF][G][PITCH];
    __shared__ __attribute__((aligned(16))) float l_prev[PREV ? NBUF : 1][PREV ? G : 1][PREV ? PITCH : 4];
    const int tid = threadIdx.x;
    const unsigned vid = xcd_remap(blockIdx.x, nblocks);
    const int bx = vid % nbx, by = (vid / nbx) % nby;
    const long long b = vid / (nbx * nby);
    const int x0 = bx * TW;
    const int col = mirror_idx(x0 - RP + tid, w);
    const __amdgpu_buffer_rsrc_t src = buf_rsrc(in + b * s_in);
    const __amdgpu_buffer_rsrc_t r_out = buf_rsrc(out ? out + b * s_out : nullptr);
    const __amdgpu_buffer_rsrc_t r_dog = buf_rsrc(dog ? dog + b * s_dog : nullptr);
    const __amdgpu_buffer_rsrc_t r_sub = buf_rsrc(sub ? sub + b * s_sub : nullptr);
    const __amdgpu_buffer_rsrc_t r_flags =
        buf_rsrc(flags ? flags + b * ((h + 7) >> 3) * (long long)((w + 63) >> 6) : nullptr);
    const unsigned ucol = (unsigned)col * 4u;
    OvsCol oc;
    if (OVS) {
        const float y = col * delta;
        oc.jm = (int)y; oc.jp = oc.jm + 1;
        if (oc.jp >= wi) oc.jp = 2 * wi - 1 - oc.jp;
        if (oc.jm >= wi) oc.jm = 2 * wi - 1 - oc.jm;
        oc.fy = y - floorf(y);
    }
    auto load = [&](int i) -> float {
        const int r = mirror_idx(i, h);
        if (OVS) return ovs_sample(src, wi, hi, delta, r, oc);
        return buf_ld(src, ucol, (unsigned)(r * w) * 4u);
    };
    int i0 = by * strip;
    int i1 = min(i0 + strip, h);
    // SPARSE: bit j of `need` = the j-th 8-row group of the strip is evaluated; bit j of `fetch` = the
    // source rows that enter the column window in iteration j are used by an evaluated group
    unsigned long long need = ~0ull;
    // SPARSE, per wave: `cneed` = groups whose column pass this wave runs (its 64 columns feed an evaluated
    // 64-column chunk of outputs), `cfetch` = iterations whose source rows it loads, `rneed[r]` = groups in
    // which the chunk of outputs of its round r is evaluated.  A wave of the row pass owns 64 columns x 4 rows,
    // so everything a wave skips is skipped by all its lanes.
    constexpr int NWV = EW / 64, NCH = (NG + 15) / 16, RROUNDS = (2 * NCH + NWV - 1) / NWV;
    unsigned long long cneed = ~0ull, cfetch = ~0ull, rneed[RROUNDS];
#pragma unroll
    for (int r = 0; r < RROUNDS; ++r) rneed[r] = ~0ull;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (SPARSE) {
        constexpr int NCC = (TW + 12 + 63) / 64 + 1;       // cell columns within 6 samples of a tile, at most
        static_assert((NCH & (NCH - 1)) == 0 && RROUNDS == (ITEMS + EW - 1) / EW, "chunk layout");
        __shared__ unsigned long long l_need_c[NCC];
        if (tid < NCC) l_need_c[tid] = 0;
        __syncthreads();
        const int cw = (w + 63) >> 6, ch = (h + 7) >> 3;
        const int* cells = need_cells + b * (long long)ch * cw;
        const int g0 = i0 >> 3, ng = (i1 - i0 + 7) >> 3;
        const int c0 = max((x0 - 6) >> 6, 0), c1 = min((min(x0 + TW, w) + 5) >> 6, cw - 1);
        const int ncol = c1 - c0 + 1;
        const unsigned long long strip_mask = ng >= 64 ? ~0ull : (1ull << ng) - 1;
        for (int t = tid; t < (ng + 2) * ncol; t += EW) {
            const int gr = t / ncol, cc = t - gr * ncol;      // cell row g0 - 1 + gr
            const int cr = g0 - 1 + gr;
            if (cr >= 0 && cr < ch && cells[cr * cw + c0 + cc]) {
                // a marked cell row needs its own group of rows and the two next to it: groups gr-2 .. gr of the strip
                const unsigned long long m = (gr >= 2 ? 7ull << (gr - 2) : 7ull >> (2 - gr)) & strip_mask;
                if (m) atomicOr(&l_need_c[cc], m);
            }
        }
        __syncthreads();
        // groups in which output columns lo..hi are evaluated: those of the cells within 6 samples of them
        auto cols_need = [&](int lo, int hi) -> unsigned long long {
            unsigned long long m = 0;
            if (lo <= hi)
                for (int cc = max((lo - 6) >> 6, c0); cc <= min((hi + 6) >> 6, c1); ++cc) m |= l_need_c[cc - c0];
            return ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32) |
                   __builtin_amdgcn_readfirstlane((unsigned)m);
        };
        auto chunk_need = [&](int c) -> unsigned long long {   // chunk c: output columns x0 + 64 c ... of the tile
            if (c < 0 || c >= NCH) return 0ull;
            return cols_need(x0 + 64 * c, min(min(x0 + 64 * c + 63, x0 + TW - 1), w - 1));
        };
        need = cols_need(x0, min(x0 + TW, w) - 1);
        if (need == 0) return;
        // column wv of the tile's LDS rows (64 samples, starting RP left of the outputs) is read by the
        // row pass of chunks wv - 1 and wv (R <= RP = 16)
        cneed = chunk_need(wv - 1) | chunk_need(wv);
#pragma unroll
        for (int r = 0; r < RROUNDS; ++r) {
            const int wr = r * NWV + wv;
            rneed[r] = wr < 2 * NCH ? chunk_need(wr & (NCH - 1)) : 0ull;
        }
        // the strip shrinks to the span between its first and last needed group
        const int lead = __builtin_ctzll(need);
        i0 += 8 * lead;
        need >>= lead;
        cneed >>= lead;
#pragma unroll
        for (int r = 0; r < RROUNDS; ++r) rneed[r] >>= lead;
        i1 = min(i1, i0 + 8 * (64 - __builtin_clzll(need)));
        cfetch = cneed;
#pragma unroll
        for (int k = 1; k <= (2 * R + 7) / 8; ++k) cfetch |= cneed >> k;
    }
    auto bit = [](unsigned long long m, int j) { return j < 64 && ((m >> j) & 1ull); };
    // VW: the launcher guarantees w % 4 == 0, every 4-sample item is then a whole 16-byte access
    const bool vec_w = VW || (w & 3) == 0;
    const int wo = w >> 1, ho = h >> 1;

    // Column window of G + 2R samples, kept as register pairs so that two adjacent output rows
    // are filtered by one packed instruction (v_pk_add_f32 / v_pk_mul_f32, IEEE like the scalar
    // forms): wp[i] = {win[2i], win[2i+1]}; see win_pair_sum for the taps at odd offsets.
    // Source rows are fetched one iteration (8 rows) ahead of their use.
    constexpr int NP = (G + 2 * R) / 2;
    v2f wp[NP];
    float nx[G];
    // G consecutive source rows.  With the oversampling fused in and a step of exactly 1/2, rows
    // 2m and 2m+1 of the oversampled image interpolate the same two input rows, so the
    // horizontally interpolated input rows H[m] = fy*I[m][jp] + (1-fy)*I[m][jm] are computed once
    // (5 or 6 of them per 8 rows) and every sample is fx*H[m+1] + (1-fx)*H[m]: the expressions
    // and their order are those of ovs_sample, only shared.  Rows that need mirroring take the
    // per-sample path.
    const bool half_step = OVS && delta == 0.5f;
    auto load_rows = [&](int ifirst, float* dst) {
        if (OVS && half_step && ifirst >= 0 && ifirst + G - 1 < h && ((ifirst + G - 1) >> 1) + 1 < hi) {
            const int m = ifirst >> 1;
            const bool odd = ifirst & 1;
            float H[G / 2 + 2];
#pragma unroll
            for (int t = 0; t < G / 2 + 2; ++t) {
                H[t] = 0.f;
                if (t < G / 2 + 1 || odd) {
                    const unsigned rp = (unsigned)((m + t) * wi) * 4u;
                    const float a = buf_ld(src, oc.jp * 4u, rp), b = buf_ld(src, oc.jm * 4u, rp);
                    H[t] = __fadd_rn(__fmul_rn(oc.fy, a), __fmul_rn(1 - oc.fy, b));
                }
            }
            if (odd) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int t = (g + 1) >> 1;
                    const float fx = ((g + 1) & 1) ? 0.5f : 0.f;
                    dst[g] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
                }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int t = g >> 1;
                    const float fx = (g & 1) ? 0.5f : 0.f;
                    dst[g] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) dst[g] = load(ifirst + g);
        }
    };
    {
        float pro[2 * R];
#pragma unroll
        for (int c = 0; c + G <= 2 * R; c += G) load_rows(i0 - R + c, &pro[c]);
#pragma unroll
        for (int t = (2 * R / G) * G; t < 2 * R; ++t) pro[t] = load(i0 - R + t);
#pragma unroll
        for (int t = 0; t < R; ++t) { wp[t].x = pro[2 * t]; wp[t].y = pro[2 * t + 1]; }
    }
    // Prefetch of the next G rows, split in two so that the loads stay in flight during a whole
    // iteration: `issue` starts them (fast path: the raw input samples of the shared rows),
    // `finish` interpolates at the top of the next iteration.
    float raw[OVS ? 2 * (G / 2 + 2) : 1];
    bool raw_pending = false, raw_odd = false;
    auto issue_rows = [&](int ifirst) {
        if (OVS && half_step && ifirst >= 0 && ifirst + G - 1 < h && ((ifirst + G - 1) >> 1) + 1 < hi) {
            const int m = ifirst >> 1;
            raw_odd = ifirst & 1;
#pragma unroll
            for (int t = 0; t < G / 2 + 2; ++t) {
                // the last row is only needed for an odd start; reading row m+t <= hi-1 is always in range
                const int row = min(m + t, hi - 1);
                const unsigned rp = (unsigned)(row * wi) * 4u;
                raw[2 * t] = buf_ld(src, oc.jp * 4u, rp);
                raw[2 * t + 1] = buf_ld(src, oc.jm * 4u, rp);
            }
            raw_pending = true;
        } else {
            raw_pending = false;
#pragma unroll
            for (int g = 0; g < G; ++g) nx[g] = load(ifirst + g);
        }
    };
    auto finish_rows = [&]() {
        if (!OVS || !raw_pending) return;
        float H[G / 2 + 2];
#pragma unroll
        for (int t = 0; t < G / 2 + 2; ++t)
            H[t] = __fadd_rn(__fmul_rn(oc.fy, raw[OVS ? 2 * t : 0]), __fmul_rn(1 - oc.fy, raw[OVS ? 2 * t + 1 : 0]));
        if (raw_odd) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int t = (g + 1) >> 1;
                const float fx = ((g + 1) & 1) ? 0.5f : 0.f;
                nx[g] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int t = g >> 1;
                const float fx = (g & 1) ? 0.5f : 0.f;
                nx[g] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
            }
        }
        raw_pending = false;
    };
    issue_rows(i0 + R);

    // One iteration: 8 rows of the strip starting at row ii.  nxr holds their newest source rows,
    // prefetch() starts the loads of a later iteration once nxr has been taken into the window.
    auto iteration = [&](int ii, int buf, float (&nxr)[G], auto&& prefetch, auto pinned, bool run, int j) {
        if constexpr (decltype(pinned)::value) {
            // asm moves: the copies out of the load registers stay here, where only these rows' loads
            // have to be complete, instead of becoming register copies at the loop's back edge
#pragma unroll
            for (int q = 0; q < G / 2; ++q) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(wp[R + q].x) : "v"(nxr[2 * q]));
                asm volatile("v_mov_b32 %0, %1" : "=v"(wp[R + q].y) : "v"(nxr[2 * q + 1]));
            }
        } else {
#pragma unroll
            for (int q = 0; q < G / 2; ++q) { wp[R + q].x = nxr[2 * q]; wp[R + q].y = nxr[2 * q + 1]; }
        }
        prefetch();
        if (!SPARSE || run) {
        // two output pairs at a time: the two accumulation chains are independent, so a wave has a
        // second instruction to issue while the first one's result is on its way
        if (!SPARSE || bit(cneed, j)) {
#pragma unroll
        for (int g = 0; g < G; g += 4) {
            const v2f ctr0 = win_pair(wp, g + R), ctr1 = win_pair(wp, g + 2 + R);
            v2f s0 = ctr0 * tp.g[0], s1 = ctr1 * tp.g[0];
#pragma unroll
            for (int k = 1; k <= R; ++k) {
                const v2f a0 = win_pair_sum(wp, g + R - k, g + R + k);
                const v2f a1 = win_pair_sum(wp, g + 2 + R - k, g + 2 + R + k);
                const v2f m0 = tp.g[k] * a0, m1 = tp.g[k] * a1;
                s0 = s0 + m0;
                s1 = s1 + m1;
            }
            l_tmp[buf][g][tid] = s0.x;
            l_tmp[buf][g + 1][tid] = s0.y;
            l_tmp[buf][g + 2][tid] = s1.x;
            l_tmp[buf][g + 3][tid] = s1.y;
            if (PREV) {                  // unfiltered centre samples, for the DoG epilogue
                l_prev[buf][g][tid] = ctr0.x;
                l_prev[buf][g + 1][tid] = ctr0.y;
                l_prev[buf][g + 2][tid] = ctr1.x;
                l_prev[buf][g + 3][tid] = ctr1.y;
            }
        }
        }
        __syncthreads();
#pragma unroll
        for (int rnd = 0; rnd < (ITEMS + EW - 1) / EW; ++rnd) {
            int hg, hq;
            bool item_ok;
            if constexpr (SPARSE) {
                // a wave = one 64-column chunk x 4 rows (16 lanes per row)
                const int wr = rnd * NWV + wv;
                hg = 4 * (wr / NCH) + ((tid & 63) >> 4);
                hq = 16 * (wr & (NCH - 1)) + (tid & 15);
                item_ok = wr < 2 * NCH && hq < NG && bit(rneed[rnd], j);
            } else {
                const int item = rnd * EW + tid;
                hg = item / NG; hq = item - hg * NG;
                item_ok = item < ITEMS;
            }
            const int row = ii + hg, xo = x0 + 4 * hq;
            if (item_ok && row < i1 && xo < w) {
                // window as aligned pairs hp[i] = {row[2i], row[2i+1]}
                v2f hp[(4 + 2 * RP) / 2];
                const float* lrow = &l_tmp[buf][hg][4 * hq];
#pragma unroll
                for (int q = 0; q < (4 + 2 * RP) / 4; ++q) {
                    const v4f v = ld4(lrow + 4 * q);
                    hp[2 * q] = v2f{v.x, v.y};
                    hp[2 * q + 1] = v2f{v.z, v.w};
                }
                v2f o01 = hp[RP / 2] * tp.g[0], o23 = hp[RP / 2 + 1] * tp.g[0];
#pragma unroll
                for (int k = 1; k <= R; ++k) {
                    o01 = o01 + tp.g[k] * win_pair_sum(hp, RP - k, RP + k);
                    o23 = o23 + tp.g[k] * win_pair_sum(hp, RP + 2 - k, RP + 2 + k);
                }
                const float o4[4] = {o01.x, o01.y, o23.x, o23.y};
                const bool vec = VW || (vec_w && (xo + 4 <= w));
                // store addresses: wave-uniform byte offset of the first row of this iteration plus
                // a lane offset that does not change from iteration to iteration
                const unsigned tile_off = (unsigned)(ii * w) * 4u;
                const unsigned loff = (unsigned)(hg * w + xo) * 4u;
                if (out) {
                    if (vec) {
                        buf_st4(r_out, loff, tile_off, v4f{o4[0], o4[1], o4[2], o4[3]});
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (xo + q < w) buf_st1(r_out, loff + 4 * q, tile_off, o4[q]);
                    }
                }
                if (PREV) {
                    const v4f pv = ld4(&l_prev[buf][hg][RP + 4 * hq]);
                    const float d4[4] = {__fsub_rn(o4[0], pv.x), __fsub_rn(o4[1], pv.y),
                                         __fsub_rn(o4[2], pv.z), __fsub_rn(o4[3], pv.w)};
                    if (dog) {
                        if (vec) {
                            buf_st4(r_dog, loff, tile_off, v4f{d4[0], d4[1], d4[2], d4[3]});
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (xo + q < w) buf_st1(r_dog, loff + 4 * q, tile_off, d4[q]);
                        }
                    }
                    // mark the 8-row x 64-column cell if a sample passes the first contrast test
                    // (columns past w hold mirrored data and only ever add marks)
                    if (flags) {
                        const float m = fmaxf(fmaxf(fabsf(d4[0]), fabsf(d4[1])), fmaxf(fabsf(d4[2]), fabsf(d4[3])));
                        if (m > flag_thr)      // strips start at multiples of 8 rows: row >> 3 is uniform
                            __builtin_amdgcn_raw_buffer_store_b32(1u, r_flags, (unsigned)(xo >> 6) * 4u,
                                                                  (unsigned)((ii >> 3) * ((w + 63) >> 6)) * 4u, 0);
                    }
                }
                if (sub && !(hg & 1) && (row >> 1) < ho) {
                    const unsigned soff = (unsigned)((hg >> 1) * wo + (xo >> 1)) * 4u;
                    const unsigned srow = (unsigned)((ii >> 1) * wo) * 4u;
                    if (VW || (((wo & 1) == 0) && (xo >> 1) + 2 <= wo)) {
                        buf_st2(r_sub, soff, srow, v2f{o4[0], o4[2]});
                    } else {
                        if ((xo >> 1) < wo) buf_st1(r_sub, soff, srow, o4[0]);
                        if ((xo >> 1) + 1 < wo) buf_st1(r_sub, soff + 4, srow, o4[2]);
                    }
                }
            }
        }
        }   // run
#pragma unroll
        for (int t = 0; t < R; ++t) wp[t] = wp[t + G / 2];
    };
    if (!OVS && (EW == 512 || SPARSE) && strip >= 8 * G) {
        // Source rows are fetched two iterations ahead, into two register sets used in turn.  When an
        // iteration waits for its rows, the eight loads of the following iteration are always younger
        // than them: the wait (vmcnt counts loads and stores in issue order) then never extends to
        // the stores the wave has just issued.  With a distance of one iteration every wave waited
        // for the write acknowledgements of its last stores once per iteration.  Loads past the
        // strip are issued with an out-of-range lane offset (dropped by the range check), so that
        // the number of loads per iteration does not depend on the path taken.
        float nxb[G];
        auto issue_into = [&](float (&dst)[G], int ifirst, bool live) {
            const unsigned lane = live ? ucol : 0x80000000u;
#pragma unroll
            for (int g = 0; g < G; ++g) dst[g] = buf_ld(src, lane, (unsigned)(mirror_idx(ifirst + g, h) * w) * 4u);
        };
        issue_into(nxb, i0 + G + R, i0 + G < i1 && bit(cfetch, 1));
        // iterations in pairs; when the strip has an odd number of them the last one finds all its
        // rows past the strip and stores nothing
        if constexpr (!SPARSE) {
            for (int ii = i0; ii < i1; ii += 2 * G) {
                iteration(ii, 0, nx, [&] { issue_into(nx, ii + 2 * G + R, ii + 2 * G < i1); }, SaTrue{}, true, 0);
                iteration(ii + G, NBUF - 1, nxb, [&] { issue_into(nxb, ii + 3 * G + R, ii + 3 * G < i1); }, SaTrue{}, true, 0);
            }
        } else {
            // the LDS buffers alternate between evaluated iterations only: each evaluated iteration has
            // one barrier, and a buffer is written again two evaluated iterations after it was read
            int lb = 0, j = 0;
            for (int ii = i0; ii < i1; ii += 2 * G, j += 2) {
                const bool r0 = bit(need, j), r1 = bit(need, j + 1);
                iteration(ii, lb, nx, [&] { issue_into(nx, ii + 2 * G + R, ii + 2 * G < i1 && bit(cfetch, j + 2)); }, SaTrue{}, r0, j);
                if (r0) lb ^= 1;
                iteration(ii + G, lb, nxb, [&] { issue_into(nxb, ii + 3 * G + R, ii + 3 * G < i1 && bit(cfetch, j + 3)); }, SaTrue{}, r1, j + 1);
                if (r1) lb ^= 1;
            }
        }
    } else {
        int buf = 0, j = 0;
        for (int ii = i0; ii < i1; ii += G, ++j) {
            finish_rows();
            const bool run = !SPARSE || bit(need, j);
            iteration(ii, buf, nx, [&] { if (ii + G < i1 && (!SPARSE || bit(cfetch, j + 1))) issue_rows(ii + G + R); }, SaFalse{}, run, j);
            if (run) buf ^= (NBUF - 1);
        }
    }
#ifdef SA_BLUR_CLOCK
    if (tid == 0 && !OVS) {
        atomicAdd(&sa_blur_clock[2 * R], __builtin_amdgcn_s_memtime() - clk_t0);
        atomicAdd(&sa_blur_clock[2 * R + 1], __builtin_amdgcn_s_memrealtime() - clk_r0);
    }
#endif
}

// ------------------------------------------------- two blurs in one launch --
// Two successive blurs of an octave (image s-1 -> s -> s+1) in one launch, so that image s is
// written but never read back: 12 instead of 16 bytes per pixel.  A block of 1024 threads has two
// roles of 8 waves each.  The producer is the fused blur above (source rows from global memory, or
// the bilinearly oversampled input for the seed image; column pass in registers, row pass through
// LDS) and also leaves its output rows, 8 per iteration, in an LDS ring; the consumer runs the same
// two passes with radius RB on the rows it takes from the ring.  Both roles share one barrier per
// iteration: the producer fills ring slot j&1 behind barrier j, the consumer reads it behind barrier
// j+1 and uses it in iteration j+2, so the consumer trails by 16 + RB rows and warms up for
// 2 + ceil(RB/4) iterations before its window is complete.  Rows and columns outside the image are
// produced at their mirrored positions, which gives the bits of the mirrored result (the taps are
// symmetric and the pair sums commute).  A tile is 512 columns wide: the producer's results are
// valid on 512 - 2 HA of them, the consumer's on 512 - 2 (HA + HB), the columns a block stores
// (HA, HB: the radii rounded up to a multiple of 4).
#ifdef SA_PAIR_CLOCK
// Diagnostic build only: shader-clock ticks the waves of the two-blur launches spend per phase of an
// iteration, summed over all waves: [OVS][role][phase], phases = rows into the window (incl. the wait
// for their loads) / column pass / barrier / row pass and stores / iterations.
__device__ unsigned long long sa_pair_clock[2][2][5];
extern "C" void sift_amd_debug_pair_clock(unsigned long long* out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_pair_clock), sizeof(sa_pair_clock));
    if (reset) { unsigned long long z[20] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(sa_pair_clock), z, sizeof(z)); }
}
#define SA_PC_DECL unsigned long long pc_t = __builtin_amdgcn_s_memtime(), pc_acc[5] = {0, 0, 0, 0, 0}
#define SA_PC_MARK(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pc_acc[k] += n_ - pc_t; pc_t = n_; } while (0)
#define SA_PC_FLUSH(ovs, role) do { if ((tid & 63) == 0) for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&sa_pair_clock[ovs][role][k_], pc_acc[k_]); } while (0)
#else
#define SA_PC_DECL
#define SA_PC_MARK(k)
#define SA_PC_FLUSH(ovs, role)
#endif

#ifndef SA_PAIR_HELP_FROM
#define SA_PAIR_HELP_FROM 768
#endif
namespace pairk {
constexpr int G = 8, EW = 512;
struct Geo {
    int w, h;
    int X0;                 // global column of tile-local column 0
    int keep_lo, keep_hi;   // global columns this block stores
    int i0, i1;             // rows this block stores
    int n_iter;             // iterations of both roles
    int cells_w;            // marks per row of 8 x 64 cells
};

// 4 outputs from a window of 4 + 2H column-filtered samples starting at lrow (16-byte aligned)
template <int R, int H>
__device__ __forceinline__ void row_item(const float* lrow, const SaTapPairs& tp, float (&o4)[4])
{
    v2f hp[(4 + 2 * H) / 2];
#pragma unroll
    for (int q = 0; q < (4 + 2 * H) / 4; ++q) {
        const v4f v = ld4(lrow + 4 * q);
        hp[2 * q] = v2f{v.x, v.y};
        hp[2 * q + 1] = v2f{v.z, v.w};
    }
    v2f o01 = hp[H / 2] * tp.g[0], o23 = hp[H / 2 + 1] * tp.g[0];
#pragma unroll
    for (int k = 1; k <= R; ++k) {
        o01 = o01 + tp.g[k] * win_pair_sum(hp, H - k, H + k);
        o23 = o23 + tp.g[k] * win_pair_sum(hp, H + 2 - k, H + 2 + k);
    }
    o4[0] = o01.x; o4[1] = o01.y; o4[2] = o23.x; o4[3] = o23.y;
}

// column pass of one iteration: 8 outputs of column tid from the register window; tmp / prev point
// at the iteration's 8 x PITCH buffers
template <int R, int PITCH, bool PREV>
__device__ __forceinline__ void col_pass(const v2f* wp, const SaTapPairs& tp, float* tmp, float* prev, int tid)
{
#pragma unroll
    for (int g = 0; g < G; g += 2) {
        const v2f ctr = win_pair(wp, g + R);
        v2f s = ctr * tp.g[0];
#pragma unroll
        for (int k = 1; k <= R; ++k) s = s + tp.g[k] * win_pair_sum(wp, g + R - k, g + R + k);
        tmp[g * PITCH + tid] = s.x;
        tmp[(g + 1) * PITCH + tid] = s.y;
        if (PREV) { prev[g * PITCH + tid] = ctr.x; prev[(g + 1) * PITCH + tid] = ctr.y; }
    }
}

// stores 4 results; with PREV the contrast mark of their cell; with SUB the even samples of even
// rows into the next octave's first image (lib_discrete.c:180-191)
template <bool PREV, bool SUB>
__device__ __forceinline__ void emit(const Geo& g, __amdgpu_buffer_rsrc_t r_out, __amdgpu_buffer_rsrc_t r_flags,
                                     __amdgpu_buffer_rsrc_t r_sub, const float (&o4)[4], const float* prev4,
                                     float flag_thr, int row, int xo)
{
    buf_st4(r_out, (unsigned)(row * g.w + xo) * 4u, 0u, v4f{o4[0], o4[1], o4[2], o4[3]});
    if (PREV) {
        const v4f pv = ld4(prev4);
        const float m = fmaxf(fmaxf(fabsf(__fsub_rn(o4[0], pv.x)), fabsf(__fsub_rn(o4[1], pv.y))),
                              fmaxf(fabsf(__fsub_rn(o4[2], pv.z)), fabsf(__fsub_rn(o4[3], pv.w))));
        if (m > flag_thr)
            __builtin_amdgcn_raw_buffer_store_b32(1u, r_flags, (unsigned)((row >> 3) * g.cells_w + (xo >> 6)) * 4u, 0u, 0);
    }
    if (SUB && !(row & 1) && (row >> 1) < (g.h >> 1))
        buf_st2(r_sub, (unsigned)((row >> 1) * (g.w >> 1) + (xo >> 1)) * 4u, 0u, v2f{o4[0], o4[2]});
}

// CSA: the consumer writes the producer's image to memory (from the ring slot it reads anyway), so
// that the producer issues loads only: its wait for a set of rows never includes a store, and the
// role with the spare time does the storing.
// HELP_FROM: the consumer's row-pass items from this index on are computed by the producer's threads
// after their own (the consumer's column-pass rows lie in LDS for everybody); it balances the two
// roles where the consumer is the slower one.  G * NG of the consumer and more: no help.
template <int R, int HA, int HB, int PITCH, bool PREV, bool OVS, bool CSA, bool ODD, int RB, bool FB, bool SUB, int HELP_FROM>
__device__ __forceinline__ void producer(const float* in_img, int wi, int hi, float delta, float* out_img, int* flags_img,
                                         const SaTapPairs& tp, float flag_thr, const Geo& g, int r0, int tid,
                                         float* l_tmp, float* l_prev, float* ring,
                                         float* out_b_img, float* sub_img, const SaTapPairs& tp_b, int r0b, int warm,
                                         float* l_tmp_c, float* l_prev_c)
{
    constexpr int NG = (EW - 2 * HA) / 4, ITEMS = G * NG, NP = (G + 2 * R) / 2;
    constexpr unsigned OOB = 0x80000000u;       // lane offset past the descriptor's range: the access is dropped
    const __amdgpu_buffer_rsrc_t src = buf_rsrc(in_img), r_out = buf_rsrc(out_img), r_flags = buf_rsrc(flags_img);
    const int col = mirror_idx(g.X0 + tid, g.w);
    const unsigned ucol = (unsigned)col * 4u;
    OvsCol oc;
    if (OVS) {
        const float y = col * delta;
        oc.jm = (int)y; oc.jp = oc.jm + 1;
        if (oc.jp >= wi) oc.jp = 2 * wi - 1 - oc.jp;
        if (oc.jm >= wi) oc.jm = 2 * wi - 1 - oc.jm;
        oc.fy = y - floorf(y);
    }
    auto load = [&](int i) -> float {
        const int r = mirror_idx(i, g.h);
        if (OVS) return ovs_sample(src, wi, hi, delta, r, oc);
        return buf_ld(src, ucol, (unsigned)(r * g.w) * 4u);
    };
    v2f wp[NP];
    // The 8 source rows of an iteration are fetched two iterations ahead, into two register sets used
    // in turn: when an iteration waits for its rows, the loads of the following iteration are younger
    // than them, so the wait (vmcnt retires loads and stores in issue order) never extends to the
    // stores the wave has just issued.  Every iteration issues its loads; rows nobody will use get an
    // out-of-range lane offset.  With the oversampling at step 1/2, rows 2m and 2m+1 interpolate the
    // same two input rows: their horizontally interpolated rows are loaded once (raw samples issued
    // by issue(), interpolation in take() two iterations later); rows that need mirroring take the
    // per-sample path (same expressions as k_blur_fused).
    // The parity of the first row of a set is that of RA + RB (strips start at multiples of 8 rows): 8 rows of
    // an even start interpolate 5 input rows, of an odd start 6.  A thread's two input columns are neighbours
    // in memory (jp = jm + 1; in the last column both are the last sample), so one 8-byte load fetches both.
    constexpr int NROWS = G / 2 + 1 + (ODD ? 1 : 0);
    struct RowSet { float nx[G]; v2f raw[OVS ? NROWS : 1]; bool pending; };
    const bool half_step = OVS && delta == 0.5f && wi >= 2;
    const bool last_col = OVS && oc.jp != oc.jm + 1;
    const unsigned jm2 = OVS ? (unsigned)min(oc.jm, wi - 2) * 4u : 0u;
    auto issue = [&](RowSet& rs, int ifirst, bool live) {
        const bool shared = OVS && half_step && (ifirst & 1) == (ODD ? 1 : 0) && ifirst >= 0 && ifirst + G - 1 < g.h &&
                            ((ifirst + G - 1) >> 1) + 1 < hi;
        if (OVS && (shared || !live)) {
            const int m = shared ? ifirst >> 1 : 0;
            const unsigned jpair = live ? jm2 : OOB;
#pragma unroll
            for (int t = 0; t < NROWS; ++t) {
                const unsigned rp = (unsigned)(min(m + t, hi - 1) * wi) * 4u;
                rs.raw[OVS ? t : 0] = buf_ld2(src, jpair, rp);
            }
            rs.pending = true;
        } else if (OVS) {
            rs.pending = false;
#pragma unroll
            for (int q = 0; q < G; ++q) rs.nx[q] = load(ifirst + q);
        } else {
            rs.pending = false;
            const unsigned lane = live ? ucol : OOB;
#pragma unroll
            for (int q = 0; q < G; ++q) rs.nx[q] = buf_ld(src, lane, (unsigned)(mirror_idx(ifirst + q, g.h) * g.w) * 4u);
        }
    };
    // the set's rows into the top of the column window
    auto take = [&](RowSet& rs) {
        if (OVS && rs.pending) {
            float H[NROWS];
#pragma unroll
            for (int t = 0; t < NROWS; ++t) {
                const v2f in2 = rs.raw[OVS ? t : 0];          // samples jm2, jm2 + 1 of the row
                const float a = in2.y, b = last_col ? in2.y : in2.x;
                H[t] = __fadd_rn(__fmul_rn(oc.fy, a), __fmul_rn(1 - oc.fy, b));
            }
            // row ifirst + q: input rows (ifirst + q) / 2 and the next one, weight 1/2 on odd rows
            float v[G];
            if (ODD) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int t = (q + 1) >> 1;
                    const float fx = ((q + 1) & 1) ? 0.5f : 0.f;
                    v[q] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
                }
            } else {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int t = q >> 1;
                    const float fx = (q & 1) ? 0.5f : 0.f;
                    v[q] = __fadd_rn(__fmul_rn(fx, H[t + 1]), __fmul_rn(1 - fx, H[t]));
                }
            }
#pragma unroll
            for (int q = 0; q < G / 2; ++q) wp[R + q] = v2f{v[2 * q], v[2 * q + 1]};
        } else {
            // asm moves: the copies out of the load registers stay here, where only this set's loads
            // have to be complete, instead of becoming register copies at the loop's back edge
#pragma unroll
            for (int q = 0; q < G / 2; ++q) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(wp[R + q].x) : "v"(rs.nx[2 * q]));
                asm volatile("v_mov_b32 %0, %1" : "=v"(wp[R + q].y) : "v"(rs.nx[2 * q + 1]));
            }
        }
    };
    {
        float pro[2 * R];
#pragma unroll
        for (int t = 0; t < 2 * R; ++t) pro[t] = load(r0 - R + t);
#pragma unroll
        for (int t = 0; t < R; ++t) wp[t] = v2f{pro[2 * t], pro[2 * t + 1]};
    }
    // iteration j stores rows r0 + 8j ..., if its rows still reach the consumer before the consumer's
    // last iteration (j + 2 < n_iter); n_iter is even
    SA_PC_DECL;
    auto iteration = [&](int j, int buf, RowSet& rs) {
        const int ii = r0 + G * j;
        const bool work = j + 2 < g.n_iter;
        take(rs);
        issue(rs, ii + 2 * G + R, j + 4 < g.n_iter);
        SA_PC_MARK(0);
        float* tmp = l_tmp + buf * G * PITCH;
        float* prev = l_prev + (PREV ? buf * G * PITCH : 0);
        if (work) col_pass<R, PITCH, PREV>(wp, tp, tmp, prev, tid);
        SA_PC_MARK(1);
        __syncthreads();
        SA_PC_MARK(2);
        if (work) {
            float* slot = ring + buf * G * PITCH;
#pragma unroll
            for (int rnd = 0; rnd < (ITEMS + EW - 1) / EW; ++rnd) {
                const int item = rnd * EW + tid;
                const int hg = item / NG, hq = item - hg * NG;
                if (item < ITEMS) {
                    float o4[4];
                    row_item<R, HA>(tmp + hg * PITCH + 4 * hq, tp, o4);
                    const int cg = HA + 4 * hq, row = ii + hg, xo = g.X0 + cg;
                    *reinterpret_cast<v4f*>(slot + hg * PITCH + cg) = v4f{o4[0], o4[1], o4[2], o4[3]};
                    if (!CSA && row >= g.i0 && row < g.i1 && xo >= g.keep_lo && xo < g.keep_hi)
                        emit<PREV, false>(g, r_out, r_flags, r_out, o4, prev + (PREV ? hg * PITCH + cg : 0), flag_thr, row, xo);
                }
            }
        }
        constexpr int OFF_C = HA + HB, NG_C = (EW - 2 * OFF_C) / 4, ITEMS_C = G * NG_C;
        if (HELP_FROM < ITEMS_C && j >= warm) {
            static_assert(HELP_FROM >= ITEMS_C || ITEMS_C - HELP_FROM <= EW, "one helper item per thread");
            const float* tmp_c = l_tmp_c + buf * G * PITCH;
            const float* prev_c = l_prev_c + (FB ? buf * G * PITCH : 0);
            const int item = HELP_FROM + tid;
            const int hg = item / NG_C, hq = item - hg * NG_C;
            const int cg = OFF_C + 4 * hq, row = r0b + G * j + hg, xo = g.X0 + cg;
            if (item < ITEMS_C && row < g.i1 && xo < g.keep_hi) {
                const __amdgpu_buffer_rsrc_t r_out_b = buf_rsrc(out_b_img), r_sub = buf_rsrc(sub_img);
                float o4[4];
                row_item<RB, HB>(tmp_c + hg * PITCH + cg - HB, tp_b, o4);
                emit<FB, SUB>(g, r_out_b, r_flags, r_sub, o4, prev_c + (FB ? hg * PITCH + cg : 0), flag_thr, row, xo);
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) wp[t] = wp[t + G / 2];
        SA_PC_MARK(3);
#ifdef SA_PAIR_CLOCK
        pc_acc[4] += 1;
#endif
    };
    RowSet ra, rb;
    issue(ra, r0 + R, true);
    issue(rb, r0 + G + R, 3 < g.n_iter);
    for (int j = 0; j < g.n_iter; j += 2) {
        iteration(j, 0, ra);
        iteration(j + 1, 1, rb);
    }
    SA_PC_FLUSH(OVS ? 1 : 0, 0);
}

template <int R, int HA, int HB, int PITCH, bool PREV, bool SUB, bool CSA, int HELP_FROM>
__device__ __forceinline__ void consumer(float* out_img, int* flags_img, float* sub_img, const SaTapPairs& tp, float flag_thr,
                                         const Geo& g, int r0, int warm, int tid, float* l_tmp, float* l_prev, float* ring,
                                         float* out_a_img, int r0a)
{
    const __amdgpu_buffer_rsrc_t r_out_a = buf_rsrc(out_a_img);
    constexpr int OFF = HA + HB, NG = (EW - 2 * OFF) / 4, ITEMS = G * NG, NP = (G + 2 * R) / 2;
    const __amdgpu_buffer_rsrc_t r_out = buf_rsrc(out_img), r_flags = buf_rsrc(flags_img), r_sub = buf_rsrc(sub_img);
    v2f wp[NP];
    float nx[G];
#pragma unroll
    for (int t = 0; t < NP; ++t) wp[t] = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < G; ++q) nx[q] = 0.f;
    int buf = 0;
    SA_PC_DECL;
    for (int j = 0; j < g.n_iter; ++j, buf ^= 1) {
        const int ii = r0 + G * j;
        const bool work = j >= warm;         // the window holds produced rows only from here on
#pragma unroll
        for (int q = 0; q < G / 2; ++q) wp[R + q] = v2f{nx[2 * q], nx[2 * q + 1]};
        SA_PC_MARK(0);
        float* tmp = l_tmp + buf * G * PITCH;
        float* prev = l_prev + (PREV ? buf * G * PITCH : 0);
        if (work) col_pass<R, PITCH, PREV>(wp, tp, tmp, prev, tid);
        SA_PC_MARK(1);
        __syncthreads();
        SA_PC_MARK(2);
        // the rows the producer left in its previous iteration (slot (j-1)&1)
        const float* slot = ring + ((j + 1) & 1) * G * PITCH;
#pragma unroll
        for (int q = 0; q < G; ++q) nx[q] = slot[q * PITCH + tid];
        if (CSA && j >= 1 && j + 1 < g.n_iter) {
            // the slot holds rows r0a + 8 (j-1) ... of the producer's image, columns HA .. EW - HA of the tile
            constexpr int NGA = (EW - 2 * HA) / 4, ITEMS_A = G * NGA;
            const int ra = r0a + G * (j - 1);
#pragma unroll
            for (int rnd = 0; rnd < (ITEMS_A + EW - 1) / EW; ++rnd) {
                const int item = rnd * EW + tid;
                const int hg = item / NGA, hq = item - hg * NGA;
                const int cg = HA + 4 * hq, row = ra + hg, xo = g.X0 + cg;
                if (item < ITEMS_A && row >= g.i0 && row < g.i1 && xo >= g.keep_lo && xo < g.keep_hi)
                    buf_st4(r_out_a, (unsigned)(row * g.w + xo) * 4u, 0u, ld4(slot + hg * PITCH + cg));
            }
        }
        if (work) {
#pragma unroll
            for (int rnd = 0; rnd < (ITEMS + EW - 1) / EW; ++rnd) {
                const int item = rnd * EW + tid;
                const int hg = item / NG, hq = item - hg * NG;
                const int cg = OFF + 4 * hq, row = ii + hg, xo = g.X0 + cg;
                if (item < (HELP_FROM < ITEMS ? HELP_FROM : ITEMS) && row < g.i1 && xo < g.keep_hi) {
                    float o4[4];
                    row_item<R, HB>(tmp + hg * PITCH + cg - HB, tp, o4);
                    emit<PREV, SUB>(g, r_out, r_flags, r_sub, o4, prev + (PREV ? hg * PITCH + cg : 0), flag_thr, row, xo);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) wp[t] = wp[t + G / 2];
        SA_PC_MARK(3);
#ifdef SA_PAIR_CLOCK
        pc_acc[4] += 1;
#endif
    }
    SA_PC_FLUSH(SUB ? 0 : (HA == 8 && HB == 8 && !PREV ? 1 : 0), 1);
}
}  // namespace pairk

// FA / FB: the producer's / the consumer's results carry contrast marks; OVS: the producer's source
// is the bilinear oversampling of `in` (wi x hi, step delta) evaluated in its loads; SUB: the
// consumer's image also yields the next octave's first image.  With marks on both images all LDS
// rows are 512 floats (exactly the CU's 160 KB), otherwise 544 (conflict-free row-pass reads).
template <int RA, int RB, bool FA, bool FB, bool OVS, bool SUB>
__global__ void __launch_bounds__(1024)
k_blur_pair(const float* __restrict__ in, float* __restrict__ out_a, float* __restrict__ out_b, float* __restrict__ sub,
            int w, int h, long long s_in, long long s_a, long long s_b, long long s_sub, SaTapPairs tp_a, SaTapPairs tp_b,
            int strip, int nbx, int nby, int nblocks, int wi, int hi, float delta, int* __restrict__ flags, float flag_thr)
{
    using namespace pairk;
    constexpr int HA = (RA + 3) & ~3, HB = (RB + 3) & ~3, TWS = EW - 2 * (HA + HB);
    constexpr int PITCH = (FA && FB) ? EW : EW + 32;
    __shared__ __attribute__((aligned(16))) float l_tmp[2][2 * G * PITCH];
    __shared__ __attribute__((aligned(16))) float ring[2 * G * PITCH];
    __shared__ __attribute__((aligned(16))) float l_prev_a[FA ? 2 * G * PITCH : 4];
    __shared__ __attribute__((aligned(16))) float l_prev_b[FB ? 2 * G * PITCH : 4];
    const int role = threadIdx.x >> 9, tid = threadIdx.x & (EW - 1);
    const unsigned vid = xcd_remap(blockIdx.x, nblocks);
    const int bx = vid % nbx, by = (vid / nbx) % nby;
    const long long b = vid / (nbx * nby);
    Geo g;
    g.w = w; g.h = h;
    g.X0 = bx * TWS - HA - HB;
    g.keep_lo = bx * TWS; g.keep_hi = min(g.keep_lo + TWS, w);
    g.i0 = by * strip; g.i1 = min(g.i0 + strip, h);
    g.cells_w = (w + 63) >> 6;
    const int warm = 2 + (RB + 3) / 4;
    g.n_iter = (warm + (g.i1 - g.i0 + G - 1) / G + 1) & ~1;     // even: the producer walks its iterations in pairs
    const int r0b = g.i0 - G * warm, r0a = r0b + RB + 16;
    int* flags_img = flags ? flags + b * ((h + 7) >> 3) * (long long)g.cells_w : nullptr;
    // seed + first blur: the producer (loads, interpolation) is the slower role, so the consumer stores
    // both images (128 1080p images: 2.63 -> 2.51 ms; in the other pairs the consumer is the slower role)
    constexpr bool CSA = OVS && !FA;
    // the other pairs: the consumer (larger radius, marks, next octave's seed) is the slower role, and the
    // producer's threads take the tail of its row-pass items
    constexpr int ITEMS_C = G * (EW - 2 * (HA + HB)) / 4;
    constexpr int HELP_FROM = OVS ? ITEMS_C : SA_PAIR_HELP_FROM;
    float* sub_img = SUB ? sub + b * s_sub : nullptr;
    if (role == 0)
        producer<RA, HA, HB, PITCH, FA, OVS, CSA, ((RA + RB) & 1) != 0, RB, FB, SUB, HELP_FROM>(
            in + b * s_in, wi, hi, delta, out_a + b * s_a, flags_img, tp_a, flag_thr, g, r0a, tid, l_tmp[0], l_prev_a, ring,
            out_b + b * s_b, sub_img, tp_b, r0b, warm, l_tmp[1], l_prev_b);
    else
        consumer<RB, HA, HB, PITCH, FB, SUB, CSA, HELP_FROM>(out_b + b * s_b, flags_img, sub_img, tp_b, flag_thr, g, r0b,
                                                  warm, tid, l_tmp[1], l_prev_b, ring, out_a + b * s_a, r0a);
}

// Any radius (taps in global memory); one thread per output sample.  Only used for
// radii above SA_MAX_RADIUS_FAST, i.e. non-default parameters.
__global__ void __launch_bounds__(256)
k_blur_v_any(const float* __restrict__ in, float* __restrict__ out, int w, int h,
             long long s_in, long long s_out, const float* __restrict__ taps, int R)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    const float* src = in + (long long)blockIdx.z * s_in + j;
    float s = __fmul_rn(src[(long long)i * w], taps[0]);
    for (int k = 1; k <= R; ++k)
        s = __fadd_rn(s, __fmul_rn(taps[k], __fadd_rn(src[(long long)mirror_idx(i - k, h) * w],
                                                     src[(long long)mirror_idx(i + k, h) * w])));
    out[(long long)blockIdx.z * s_out + (long long)i * w + j] = s;
}

__global__ void __launch_bounds__(256)
k_blur_h_any(const float* __restrict__ tmp, const float* __restrict__ prev,
             float* __restrict__ out, float* __restrict__ dog, int w, int h,
             long long s_tmp, long long s_prev, long long s_out, long long s_dog,
             const float* __restrict__ taps, int R)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    const long long b = blockIdx.z;
    const float* srow = tmp + b * s_tmp + (long long)i * w;
    float s = __fmul_rn(srow[j], taps[0]);
    for (int k = 1; k <= R; ++k)
        s = __fadd_rn(s, __fmul_rn(taps[k], __fadd_rn(srow[mirror_idx(j - k, w)], srow[mirror_idx(j + k, w)])));
    const long long off = (long long)i * w + j;
    out[b * s_out + off] = s;
    if (dog) dog[b * s_dog + off] = __fsub_rn(s, prev[b * s_prev + off]);
}

// dog[s] = gss[s+1] - gss[s] over a whole stack (used by the stand-alone
// scalespace_compute entry point and when the blur ran without the epilogue).
__global__ void __launch_bounds__(256)
k_sub(const float* __restrict__ a, const float* __restrict__ bsrc, float* __restrict__ out,
      long long n, long long s_a, long long s_b, long long s_out)
{
    const long long z = blockIdx.z;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256)
        out[z * s_out + p] = __fsub_rn(a[z * s_a + p], bsrc[z * s_b + p]);
}

// Gradient images (lib_discrete.c:87-130), only materialised for the compatibility API.
__device__ __forceinline__ float grad_i(const float* im, int w, int h, int i, int j)
{
    if (i == 0) return __fsub_rn(im[(long long)w + j], im[j]);
    if (i == h - 1) return __fsub_rn(im[(long long)(h - 1) * w + j], im[(long long)(h - 2) * w + j]);
    return __fmul_rn(__fsub_rn(im[(long long)(i + 1) * w + j], im[(long long)(i - 1) * w + j]), 0.5f);
}
__device__ __forceinline__ float grad_j(const float* im, int w, int h, int i, int j)
{
    (void)h;
    const float* r = im + (long long)i * w;
    if (j == 0) return __fsub_rn(r[1], r[0]);
    if (j == w - 1) return __fsub_rn(r[w - 1], r[w - 2]);
    return __fmul_rn(__fsub_rn(r[j + 1], r[j - 1]), 0.5f);
}

__global__ void __launch_bounds__(256)
k_gradient(const float* __restrict__ im, float* __restrict__ gi, float* __restrict__ gj,
           int w, int h, int n_img)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    const long long off = (long long)blockIdx.z * w * h;
    gi[off + (long long)i * w + j] = grad_i(im + off, w, h, i, j);
    gj[off + (long long)i * w + j] = grad_j(im + off, w, h, i, j);
    (void)n_img;
}

// ------------------------------------------------------------ extrema scan --
// lib_sift_anatomy.c:240-313 fused with the first contrast test (:326-338).
// A wave owns 62 output columns (lanes 1..62; lanes 0 and 63 only supply neighbours) and
// walks down a strip of rows for all scales at once.  Per DoG image and row it keeps the
// horizontal 3-max / 3-min, so the 26-neighbour test of a sample is a handful of max3/min3.
// Hits are compacted with a wave ballot and one atomic per wave.
template <int NDOG>
struct ExtWin {
    float M3[NDOG][3], N3[NDOG][3];          // max / min of (left, centre, right), 3 rows
    float C[NDOG - 2][3], EM[NDOG - 2][3], Em[NDOG - 2][3];  // centre, max/min of (left,right)
};

// The DoG images are not read from memory: the scan fetches the NDOG+1 Gaussian samples of a
// row and forms dog[d] = gss[d+1] - gss[d] (lib_sift_anatomy.c:169-186, one float subtraction,
// same bits as a stored DoG image) when the row is consumed.
template <int NDOG>
__device__ __forceinline__ void ext_fetch(float (&raw)[NDOG + 1], const float* __restrict__ base,
                                          long long img_n, long long off, int j)
{
    // base/img_n/off are wave-uniform, j is the lane's column: scalar base + 32-bit lane offset
#pragma unroll
    for (int d = 0; d <= NDOG; ++d) raw[d] = (base + d * img_n + off)[j];
}

template <int NDOG, int SLOT>
__device__ __forceinline__ void ext_load(ExtWin<NDOG>& W, const float (&raw)[NDOG + 1])
{
#pragma unroll
    for (int d = 0; d < NDOG; ++d) {
        const float v = __fsub_rn(raw[d + 1], raw[d]);
        const float l = lane_prev(v), r = lane_next(v);
        W.M3[d][SLOT] = max3f(l, v, r);
        W.N3[d][SLOT] = min3f(l, v, r);
        if (d >= 1 && d <= NDOG - 2) {
            W.C[d - 1][SLOT] = v;
            W.EM[d - 1][SLOT] = fmaxf(l, r);
            W.Em[d - 1][SLOT] = fminf(l, r);
        }
    }
}

template <int NDOG, int TOP, int MID, int BOT>
__device__ __forceinline__ void ext_eval(const ExtWin<NDOG>& W, bool lane_ok, int o, int i, int j,
                                         float thr_pre, int emit_all,
                                         unsigned long long* __restrict__ ckey, float* __restrict__ cval,
                                         int* __restrict__ cnt, int cap, int& n_ext)
{
    float colM[NDOG], colm[NDOG];
#pragma unroll
    for (int d = 0; d < NDOG; ++d) {
        colM[d] = max3f(W.M3[d][TOP], W.M3[d][MID], W.M3[d][BOT]);
        colm[d] = min3f(W.N3[d][TOP], W.N3[d][MID], W.N3[d][BOT]);
    }
#pragma unroll
    for (int s = 1; s <= NDOG - 2; ++s) {
        const float c = W.C[s - 1][MID];
        const float nbM = max3f(colM[s - 1], colM[s + 1], max3f(W.M3[s][TOP], W.M3[s][BOT], W.EM[s - 1][MID]));
        const float nbm = min3f(colm[s - 1], colm[s + 1], min3f(W.N3[s][TOP], W.N3[s][BOT], W.Em[s - 1][MID]));
        const bool ext = lane_ok && (nbM < c || nbm > c);
        const unsigned long long mA = __ballot(ext);
        if (mA) {
            n_ext += __popcll(mA);
            const bool keep = ext && (emit_all || fabsf(c) > thr_pre);
            const unsigned long long mK = __ballot(keep);
            if (mK) {
                int base = 0;
                if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)mK) - 1)
                    base = atomicAdd(&cnt[SA_CNT_CAND], __popcll(mK));
                base = __shfl(base, __ffsll((long long)mK) - 1);
                if (keep) {
                    const int idx = base + __popcll(mK & lanemask_lt());
                    if (idx < cap) {
                        ckey[idx] = sa_pack_key(o, s, i, j);
                        cval[idx] = c;
                    } else {
                        cnt[SA_CNT_OVERFLOW] = 1;
                    }
                }
            }
        }
    }
}

// Walks output rows [ibeg, iend) of one wave's column segment.
template <int NDOG>
__device__ __forceinline__ void ext_walk(const float* __restrict__ base, long long img_n, int w, int h, int j,
                                         int jraw, bool lane_ok, int ibeg, int iend, int o, float thr_pre,
                                         int emit_all, unsigned long long* __restrict__ kb, float* __restrict__ vb,
                                         int* __restrict__ cnt, int cap, int& n_ext)
{
    ExtWin<NDOG> W;
    // rows are fetched two steps ahead of their use so that two rows of loads are in flight
    float rawA[NDOG + 1], rawB[NDOG + 1];
    ext_fetch<NDOG>(rawA, base, img_n, (long long)(ibeg - 1) * w, j);
    ext_fetch<NDOG>(rawB, base, img_n, (long long)ibeg * w, j);
    ext_load<NDOG, 0>(W, rawA);
    ext_fetch<NDOG>(rawA, base, img_n, (long long)min(ibeg + 1, h - 1) * w, j);
    ext_load<NDOG, 1>(W, rawB);
    int i = ibeg;
    // invariant at the top of each step: rawA/rawB (alternating) holds row i+1
    while (true) {
        ext_fetch<NDOG>(rawB, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 2>(W, rawA);
        ext_eval<NDOG, 0, 1, 2>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
        ext_fetch<NDOG>(rawA, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 0>(W, rawB);
        ext_eval<NDOG, 1, 2, 0>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
        ext_fetch<NDOG>(rawB, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 1>(W, rawA);
        ext_eval<NDOG, 2, 0, 1>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
        ext_fetch<NDOG>(rawA, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 2>(W, rawB);
        ext_eval<NDOG, 0, 1, 2>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
        ext_fetch<NDOG>(rawB, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 0>(W, rawA);
        ext_eval<NDOG, 1, 2, 0>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
        ext_fetch<NDOG>(rawA, base, img_n, (long long)min(i + 2, h - 1) * w, j);
        ext_load<NDOG, 1>(W, rawB);
        ext_eval<NDOG, 2, 0, 1>(W, lane_ok, o, i, jraw, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
        if (++i >= iend) break;
    }
}

// `flags` (optional): marks of the 8-row x 64-column cells that hold a sample of a centre scale
// passing the first contrast test (written by the blur).  A candidate's own sample must pass
// that test, so rows whose cells are all unmarked are skipped; only maximal runs of marked row
// groups are walked.  Without flags the whole strip is walked (needed when every discrete
// extremum is to be listed or counted).
template <int NDOG>
__global__ void __launch_bounds__(256)
k_extrema(const float* __restrict__ gss, long long s_gss, int w, int h, int o,
          float thr_pre, int emit_all, int rows_per_wave,
          unsigned long long* __restrict__ ckey, float* __restrict__ cval,
          int* __restrict__ counters, int cap, int nbx, int nby, int nblocks,
          const int* __restrict__ flags)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // column-adjacent blocks share cache lines at their boundary: keep them on one XCD
    const unsigned vid = xcd_remap(blockIdx.x, nblocks);
    const int bx = vid % nbx, by = (vid / nbx) % nby;
    const long long b = vid / (nbx * nby);
    const int seg = bx * 4 + wave;
    const int jraw = seg * 62 + lane;                // lane 1 of seg 0 is column 1
    const int ibeg = by * rows_per_wave + 1;          // first output row of the strip
    const int iend = min(ibeg + rows_per_wave, h - 1);  // one past the last output row
    if (seg * 62 + 1 > w - 2 || ibeg >= iend) return;  // wave-uniform
    const int j = min(jraw, w - 1);
    const bool lane_ok = lane >= 1 && lane <= 62 && jraw <= w - 2;
    const float* base = gss + b * s_gss;
    const long long img_n = (long long)w * h;
    unsigned long long* kb = ckey + b * cap;
    float* vb = cval + b * cap;
    int* cnt = counters + b * SA_CNT_STRIDE;
    int n_ext = 0;

    if (!flags) {
        ext_walk<NDOG>(base, img_n, w, h, j, jraw, lane_ok, ibeg, iend, o, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
    } else {
        // bit g of `marked`: row group (ibeg>>3)+g has a marked cell under this wave's output columns
        const int g0 = ibeg >> 3, ng = ((iend - 1) >> 3) - g0 + 1;
        const int fw = (w + 63) >> 6;
        const int c0 = (seg * 62 + 1) >> 6, c1 = min(seg * 62 + 62, w - 2) >> 6;
        const int* frow = flags + (b * ((h + 7) >> 3) + g0) * (long long)fw;
        int f = 0;                                  // lane g looks at row group g0+g (ng <= 33)
        if (lane < ng) f = frow[(long long)lane * fw + c0] | frow[(long long)lane * fw + c1];
        const unsigned long long marked = __ballot(f != 0);
        int g = 0;
        while (g < ng) {
            if (!((marked >> g) & 1ull)) { ++g; continue; }
            int ge = g;
            while (ge + 1 < ng && ((marked >> (ge + 1)) & 1ull)) ++ge;
            const int rb = max(ibeg, (g0 + g) << 3), re = min(iend, (g0 + ge + 1) << 3);
            ext_walk<NDOG>(base, img_n, w, h, j, jraw, lane_ok, rb, re, o, thr_pre, emit_all, kb, vb, cnt, cap, n_ext);
            g = ge + 1;
        }
    }
    if (lane == 0 && n_ext) {
        atomicAdd(&cnt[SA_CNT_EXT_OCT + o], n_ext);
        atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_A], n_ext);
    }
}

// ------------------------------------------------- refinement and filters --
// lib_sift_anatomy.c:358-405 (quadratic fit), :426-505 (iteration), :326-338 (contrast),
// :525-575 (edge), :685-703 (border).  One thread per candidate.
__device__ __forceinline__ void taylor3d(const float* __restrict__ st, int w, int h, int i, int j, int s,
                                         float& oX, float& oY, float& oS, float& val)
{
    const long long n = (long long)w * h;
    // dog[s] = gss[s+1] - gss[s], formed on the fly (st is the Gaussian stack)
#define D(ss, ii, jj) __fsub_rn(st[(long long)((ss) + 1) * n + (long long)(ii) * w + (jj)], \
                                st[(long long)(ss) * n + (long long)(ii) * w + (jj)])
    const float c = D(s, i, j);
    const float c2 = __fmul_rn(2.f, c);
    const float hXX = __fsub_rn(__fadd_rn(D(s, i - 1, j), D(s, i + 1, j)), c2);
    const float hYY = __fsub_rn(__fadd_rn(D(s, i, j + 1), D(s, i, j - 1)), c2);
    const float hSS = __fsub_rn(__fadd_rn(D(s + 1, i, j), D(s - 1, i, j)), c2);
    const float hXY = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(D(s, i + 1, j + 1), D(s, i + 1, j - 1)),
                                                __fsub_rn(D(s, i - 1, j + 1), D(s, i - 1, j - 1))));
    const float hXS = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(D(s + 1, i + 1, j), D(s + 1, i - 1, j)),
                                                __fsub_rn(D(s - 1, i + 1, j), D(s - 1, i - 1, j))));
    const float hYS = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(D(s + 1, i, j + 1), D(s + 1, i, j - 1)),
                                                __fsub_rn(D(s - 1, i, j + 1), D(s - 1, i, j - 1))));
    const float gX = __fmul_rn(0.5f, __fsub_rn(D(s, i + 1, j), D(s, i - 1, j)));
    const float gY = __fmul_rn(0.5f, __fsub_rn(D(s, i, j + 1), D(s, i, j - 1)));
    const float gS = __fmul_rn(0.5f, __fsub_rn(D(s + 1, i, j), D(s - 1, i, j)));
#undef D
    // det = hXX*hYY*hSS - hXX*hYS*hYS - hXY*hXY*hSS + 2*hXY*hXS*hYS - hXS*hXS*hYY
    float det = __fmul_rn(__fmul_rn(hXX, hYY), hSS);
    det = __fsub_rn(det, __fmul_rn(__fmul_rn(hXX, hYS), hYS));
    det = __fsub_rn(det, __fmul_rn(__fmul_rn(hXY, hXY), hSS));
    det = __fadd_rn(det, __fmul_rn(__fmul_rn(__fmul_rn(2.f, hXY), hXS), hYS));
    det = __fsub_rn(det, __fmul_rn(__fmul_rn(hXS, hXS), hYY));
    const float aa = __fdiv_rn(__fsub_rn(__fmul_rn(hYY, hSS), __fmul_rn(hYS, hYS)), det);
    const float ab = __fdiv_rn(__fsub_rn(__fmul_rn(hXS, hYS), __fmul_rn(hXY, hSS)), det);
    const float ac = __fdiv_rn(__fsub_rn(__fmul_rn(hXY, hYS), __fmul_rn(hXS, hYY)), det);
    const float bb = __fdiv_rn(__fsub_rn(__fmul_rn(hXX, hSS), __fmul_rn(hXS, hXS)), det);
    const float bc = __fdiv_rn(__fsub_rn(__fmul_rn(hXY, hXS), __fmul_rn(hXX, hYS)), det);
    const float cc = __fdiv_rn(__fsub_rn(__fmul_rn(hXX, hYY), __fmul_rn(hXY, hXY)), det);
    oX = __fsub_rn(__fsub_rn(__fmul_rn(-aa, gX), __fmul_rn(ab, gY)), __fmul_rn(ac, gS));
    oY = __fsub_rn(__fsub_rn(__fmul_rn(-ab, gX), __fmul_rn(bb, gY)), __fmul_rn(bc, gS));
    oS = __fsub_rn(__fsub_rn(__fmul_rn(-ac, gX), __fmul_rn(bc, gY)), __fmul_rn(cc, gS));
    const float oV = __fmul_rn(0.5f, __fadd_rn(__fadd_rn(__fmul_rn(gX, oX), __fmul_rn(gY, oY)), __fmul_rn(gS, oS)));
    val = __fadd_rn(c, oV);
}

__global__ void __launch_bounds__(256)
k_refine(SaPyramid P, SaScalars sc, const unsigned long long* __restrict__ ckey,
         const float* __restrict__ cval, int* __restrict__ counters, int cap,
         SaKey* __restrict__ ref, int* __restrict__ stage,
         unsigned long long* __restrict__ skey, int* __restrict__ sidx, int fcap)
{
    const long long b = blockIdx.y;
    int* cnt = counters + b * SA_CNT_STRIDE;
    const int n = min(cnt[SA_CNT_CAND], cap);
    int nB = 0, nC = 0, nD = 0, nE = 0, nF = 0;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
        const unsigned long long key = ckey[b * cap + c];
        const int o = (int)(key >> 44), s0 = (int)((key >> 40) & 0xf);
        const int i0 = (int)((key >> 20) & 0xfffff), j0 = (int)(key & 0xfffff);
        const SaOctave& oc = P.oct[o];
        const int w = oc.w, h = oc.h, ns = P.n_sca - 1;
        const float delta = oc.delta;
        const float* st = oc.gss + b * oc.gss_stride;      // Gaussian stack; DoG formed on the fly
        float val = cval[b * cap + c];
        SaKey k;
        k.o = o; k.s = s0; k.i = i0; k.j = j0;
        k.x = __fmul_rn(delta, (float)i0);
        k.y = __fmul_rn(delta, (float)j0);
        k.sigma = oc.sigmas[s0];
        k.theta = 0.f; k.val = val; k.edge = 0.f;
        int stg = SA_ST_A;
        if (fabsf(val) > sc.thr_pre) {
            stg = SA_ST_B; ++nB;
            int ic = i0, jc = j0, scur = s0, it = 0;
            bool conv = false;
            float oX = 0.f, oY = 0.f, oS = 0.f;
            while (it < sc.itermax) {
                if (0 < ic && ic < h - 1 && 0 < jc && jc < w - 1) {
                    taylor3d(st, w, h, ic, jc, scur, oX, oY, oS, val);
                } else {
                    oX = oY = oS = 5.0f;
                }
                if (fabsf(oX) < 0.6f && fabsf(oY) < 0.6f && fabsf(oS) < 0.6f) { conv = true; break; }
                if (oX > +0.6f && ic + 1 < h - 1) ic += 1;
                if (oX < -0.6f && ic - 1 > 0) ic -= 1;
                if (oY > +0.6f && jc + 1 < w - 1) jc += 1;
                if (oY < -0.6f && jc - 1 > 0) jc -= 1;
                if (oS > +0.6f && scur + 1 < ns - 1) scur += 1;
                if (oS < -0.6f && scur - 1 > 0) scur -= 1;
                ++it;
            }
            if (conv) {
                stg = SA_ST_C; ++nC;
                k.x = __fmul_rn(__fadd_rn((float)ic, oX), delta);
                k.y = __fmul_rn(__fadd_rn((float)jc, oY), delta);
                k.i = ic; k.j = jc; k.s = scur;
                k.sigma = (float)((double)oc.sigmas[scur] * pow((double)sc.sigma_ratio, (double)oS));
                k.val = val;
                if (fabsf(val) > sc.thr) {
                    stg = SA_ST_D; ++nD;
                    const float* g0 = st + (long long)scur * w * h;
                    const float* g1 = g0 + (long long)w * h;
                    auto dg = [&](int ii, int jj) { return __fsub_rn(g1[(long long)ii * w + jj], g0[(long long)ii * w + jj]); };
                    const float c0 = dg(ic, jc);
                    const float c2 = __fmul_rn(2.f, c0);
                    const float hXX = __fsub_rn(__fadd_rn(dg(ic - 1, jc), dg(ic + 1, jc)), c2);
                    const float hYY = __fsub_rn(__fadd_rn(dg(ic, jc + 1), dg(ic, jc - 1)), c2);
                    const float hXY = __fmul_rn(0.25f, __fsub_rn(__fsub_rn(dg(ic + 1, jc + 1), dg(ic + 1, jc - 1)),
                                                                __fsub_rn(dg(ic - 1, jc + 1), dg(ic - 1, jc - 1))));
                    const float tr = __fadd_rn(hXX, hYY);
                    k.edge = __fdiv_rn(__fmul_rn(tr, tr), __fsub_rn(__fmul_rn(hXX, hYY), __fmul_rn(hXY, hXY)));
                    if (fabsf(k.edge) <= sc.thr_edge) {
                        stg = SA_ST_E; ++nE;
                        const float ls = k.sigma;  // lambda = 1
                        const bool in = (__fsub_rn(k.x, ls) > 0.f) && (__fadd_rn(k.x, ls) < (float)P.im_h)
                                     && (__fsub_rn(k.y, ls) > 0.f) && (__fadd_rn(k.y, ls) < (float)P.im_w);
                        if (in) {
                            stg = SA_ST_F; ++nF;
                            // survivors are few: each appends its seed key and candidate index to a
                            // compact list, from which k_rank derives the reference's list order
                            const int slot = atomicAdd(&cnt[SA_CNT_SURV], 1);
                            if (slot < fcap) { skey[b * fcap + slot] = key; sidx[b * fcap + slot] = c; }
                            else cnt[SA_CNT_OVERFLOW] = 1;
                        }
                    }
                }
            }
        }
        ref[b * cap + c] = k;
        stage[b * cap + c] = stg;
    }
    // stage sizes (integer atomics: order independent)
    if (nB) atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_B], nB);
    if (nC) atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_C], nC);
    if (nD) atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_D], nD);
    if (nE) atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_E], nE);
    if (nF) atomicAdd(&cnt[SA_CNT_STAGE + SA_ST_F], nF);
}

// ----------------------------------------------------- canonical ordering --
// The reference lists are in lexicographic (o,s,i,j) order of the discrete extremum that
// seeded each keypoint.  The survivors of all filters (compacted, in no particular order, by
// k_refine) are few, so each one simply counts how many survivors have a smaller seed key
// (keys are unique) and stores its candidate index at that rank.
__global__ void __launch_bounds__(256)
k_rank(const unsigned long long* __restrict__ skey, const int* __restrict__ sidx,
       int* __restrict__ counters, int* __restrict__ flist, int fcap)
{
    __shared__ unsigned long long tile[1024];
    const long long b = blockIdx.y;
    int* cnt = counters + b * SA_CNT_STRIDE;
    const int n = min(cnt[SA_CNT_SURV], fcap);
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        const int q = base + threadIdx.x;
        const bool mine = q < n;
        const unsigned long long key = mine ? skey[b * fcap + q] : 0ull;
        int rank = 0;
        for (int t0 = 0; t0 < n; t0 += 1024) {
            __syncthreads();
            for (int t = threadIdx.x; t < 1024; t += 256) tile[t] = t0 + t < n ? skey[b * fcap + t0 + t] : ~0ull;
            __syncthreads();
            const int lim = min(1024, n - t0);
            if (mine)
                for (int t = 0; t < lim; ++t) rank += tile[t] < key;
        }
        if (mine) flist[b * fcap + rank] = sidx[b * fcap + q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[SA_CNT_NF] = n;
}

// RN(a / b) for a denominator b whose correctly rounded reciprocal y is at hand three instructions
// instead of a full division.  With y = RN(1/b), q = RN(a*y), r = a - b*q (exact in an FMA),
// RN(q + r*y) is RN(a/b) (Markstein's theorem; the operands here are far from the overflow and
// underflow ranges).
__device__ __forceinline__ float div_by_uniform(float a, float b, float y)
{
    const float q = __fmul_rn(a, y);
    return __fmaf_rn(__fmaf_rn(-b, q, a), y, q);
}
__device__ __forceinline__ double div_by_uniform(double a, double b, double y)
{
    const double q = __dmul_rn(a, y);
    return __fma_rn(__fma_rn(-b, q, a), y, q);
}

// --------------------------------------------------- orientation histogram --
__device__ __forceinline__ float modulus_2pi(float x)   // lib_util.c:187-198 with y = 2*pi
{
    constexpr float y = (float)(2 * SA_PI), yr = 1.0f / y;
    float z = x;
    int n;
    if (z < 0) {
        n = (int)div_by_uniform(-z, y, yr) + 1;
        z = __fadd_rn(z, __fmul_rn((float)n, y));
    }
    n = (int)div_by_uniform(z, y, yr);
    z = __fsub_rn(z, __fmul_rn((float)n, y));
    return z;
}

// (k + n) % n (lib_description.c:331) without the division for the values that occur
// (k in [-n, 2n)); anything else takes the modulo.
__device__ __forceinline__ int wrap_bin(int k, int n)
{
    if ((unsigned)(k + n) < (unsigned)(3 * n)) return k < 0 ? k + n : (k >= n ? k - n : k);
    return (k + n) % n;
}

__device__ __forceinline__ float bin_to_ori(float bin, int nbins)   // lib_description.c:88-94
{
    float ori = (float)(((double)bin + 0.5) * 2 * SA_PI / (double)(float)nbins);
    if ((double)ori > SA_PI) ori = (float)((double)ori - 2 * SA_PI);
    return ori;
}

// lib_description.c:122-169 (accumulation), :406-423 (smoothing), :201-227 (peaks),
// driver loops lib_sift_anatomy.c:599-681.  One wave per keypoint; lane p of a chunk handles
// the p-th sample of the patch in raster order, lane b (< n_bins) owns histogram bin b and
// adds the chunk's contributions to it in sample order (per-bin lists in LDS).
// mode 0: every peak above t*max becomes a keypoint; mode 1: only the global maximum.
__global__ void __launch_bounds__(64)
k_orientation(SaPyramid P, SaScalars sc, const SaKey* __restrict__ keys, const int* __restrict__ index,
              const int* __restrict__ counters, int cnt_slot, int kcap, int icap, int mode,
              int* __restrict__ ori_n, float* __restrict__ ori_theta, float* __restrict__ ori_hist)
{
    const int lane = threadIdx.x;
    const long long b = blockIdx.y;
    const int n = min(counters[b * SA_CNT_STRIDE + cnt_slot], icap);
    const int nbins = sc.n_bins;
    extern __shared__ float ori_lds[];             // nbins * 65 list slots, then 64 counts
    float* list_l = ori_lds;
    int* cnt_l = reinterpret_cast<int*>(ori_lds + nbins * 65);
    const unsigned long long lt = lanemask_lt();
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
        const int src = index ? index[b * icap + q] : q;
        const SaKey key = keys[b * kcap + src];
        const SaOctave& oc = P.oct[key.o];
        const int w = oc.w, h = oc.h;
        const float* im = oc.gss + b * oc.gss_stride + (long long)key.s * w * h;
        const float xk = __fdiv_rn(key.x, oc.delta), yk = __fdiv_rn(key.y, oc.delta);
        const float sk = __fdiv_rn(key.sigma, oc.delta);
        const float R = __fmul_rn(__fmul_rn(3.f, sc.lambda_ori), sk);
        const int i0 = max(0, (int)((double)__fsub_rn(xk, R) + 0.5));
        const int j0 = max(0, (int)((double)__fsub_rn(yk, R) + 0.5));
        const int i1 = min((int)((double)__fadd_rn(xk, R) + 0.5), h - 1);
        const int j1 = min((int)((double)__fadd_rn(yk, R) + 0.5), w - 1);
        const int nw = j1 - j0 + 1, nh = i1 - i0 + 1;
        const int npix = (nw > 0 && nh > 0) ? nw * nh : 0;
        const float denom = __fmul_rn(__fmul_rn(2.f, sc.lambda_ori), sc.lambda_ori);
        const float sk_rcp = __fdiv_rn(1.f, sk), denom_rcp = __fdiv_rn(1.f, denom);
        float hb = 0.f;
        // (row, column) of raster position p0 + lane, advanced by 64 positions per chunk
        int fi = 0, fj = 0;
        if (npix > 0) { fi = lane / nw; fj = lane - fi * nw; }
        for (int p0 = 0; p0 < npix; p0 += 64) {
            const int p = p0 + lane;
            int bin = -1;
            float M = 0.f;
            if (p < npix) {
                const int si = i0 + fi, sj = j0 + fj;
                const float sX = div_by_uniform(__fsub_rn((float)si, xk), sk, sk_rcp);
                const float sY = div_by_uniform(__fsub_rn((float)sj, yk), sk, sk_rcp);
                const float dx = grad_i(im, w, h, si, sj);
                const float dy = grad_j(im, w, h, si, sj);
                float ori = modulus_2pi((float)atan2((double)dy, (double)dx));
                const float r2 = __fadd_rn(__fmul_rn(sX, sX), __fmul_rn(sY, sY));
                // (dx, dy are floats: their squares are exact in double, so the root of the sum is
                // within an ulp of hypot() without its scaling steps)
                M = (float)(sqrt((double)dx * (double)dx + (double)dy * (double)dy) *
                            exp((double)div_by_uniform(-r2, denom, denom_rcp)));
                if (ori < 0) ori = (float)((double)ori + 2 * SA_PI);
                bin = (int)(div_by_uniform((double)ori, 2 * SA_PI, 1.0 / (2 * SA_PI)) * nbins + 0.5);
                bin = (unsigned)bin < 2u * (unsigned)nbins ? (bin >= nbins ? bin - nbins : bin) : bin % nbins;
            }
            fj += 64;
            while (fj >= nw) { fj -= nw; ++fi; }
            // The chunk's magnitudes go to LDS, one list per bin in sample order, and every bin
            // owner adds its list.  A sample's place in its list is the number of earlier lanes
            // with the same bin: the mask of those lanes is the intersection, over the six bits of
            // the bin number, of the ballot of that bit or its complement.
            const bool valid = bin >= 0;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool bit = (bin >> i) & 1;
                const unsigned long long bi = __ballot(bit);
                same &= bit ? bi : ~bi;
            }
            cnt_l[lane] = 0;                       // (my own bin's count; one wave, LDS operations stay in order)
            __syncthreads();
            if (valid) {
                list_l[bin * 65 + __popcll(same & lt)] = M;     // pitch 65: the owners' reads below hit 64 banks
                cnt_l[bin] = __popcll(same);                    // the same value from every lane of the bin
            }
            __syncthreads();
            const int mine = cnt_l[lane];
            for (int r = 0; __ballot(r < mine); ++r)
                if (r < mine) hb = __fadd_rn(hb, list_l[lane * 65 + r]);
            __syncthreads();
        }
        // six circular box passes
        for (int it = 0; it < 6; ++it) {
            const float pv = __shfl(hb, (lane - 1 + nbins) % nbins);
            const float nv = __shfl(hb, (lane + 1) % nbins);
            hb = (float)((double)__fadd_rn(__fadd_rn(pv, hb), nv) / 3.);
        }
        float mx = lane < nbins ? hb : -3.402823466e+38f;
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
        const float h1 = __shfl(hb, (lane - 1 + nbins) % nbins);
        const float h3 = __shfl(hb, (lane + 1) % nbins);
        bool peak;
        if (mode == 0) {
            peak = lane < nbins && hb > __fmul_rn(sc.t_peak, mx) && hb > h1 && hb > h3;
        } else {
            const unsigned long long ismax = __ballot(lane < nbins && hb == mx);
            peak = lane == __ffsll((long long)ismax) - 1;   // first bin holding the maximum
        }
        const unsigned long long pm = __ballot(peak);
        const int np = min(__popcll(pm), SA_MAX_PEAKS);
        if (peak) {
            const int r = __popcll(pm & lanemask_lt());
            if (r < SA_MAX_PEAKS) {
                const float off = __fdiv_rn(__fsub_rn(h1, h3),
                                            __fmul_rn(2.f, __fsub_rn(__fadd_rn(h1, h3), __fmul_rn(2.f, hb))));
                ori_theta[(b * icap + q) * SA_MAX_PEAKS + r] = bin_to_ori(__fadd_rn((float)lane, off), nbins);
            }
        }
        if (lane == 0) ori_n[b * icap + q] = np;
        if (lane < nbins) ori_hist[(b * icap + q) * SA_MAX_BINS + lane] = hb;
    }
}

// Exclusive scan of the per-keypoint orientation counts (in list order) and expansion
// into the final keypoint list: one entry per (keypoint, orientation), orientations of a
// keypoint adjacent and in ascending bin order (lib_sift_anatomy.c:636-640).
__global__ void __launch_bounds__(1024)
k_expand(const SaKey* __restrict__ keys, const int* __restrict__ index, int* __restrict__ counters,
         int cnt_slot, int kcap, int icap, const int* __restrict__ ori_n,
         const float* __restrict__ ori_theta, SaKey* __restrict__ out, int* __restrict__ out_src,
         int ocap)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const long long b = blockIdx.x;
    int* cnt = counters + b * SA_CNT_STRIDE;
    const int n = min(cnt[cnt_slot], icap);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int q = base + threadIdx.x;
        const int v = q < n ? ori_n[b * icap + q] : 0;
        int inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        int total = 0;
        for (int k = 0; k < 16; ++k) total += wsum[k];
        const int carry = carry_s;
        const int off = carry + woff + inc - v;
        if (q < n) {
            const int src = index ? index[b * icap + q] : q;
            SaKey k = keys[b * kcap + src];
            for (int r = 0; r < v; ++r) {
                if (off + r < ocap) {
                    k.theta = ori_theta[(b * icap + q) * SA_MAX_PEAKS + r];
                    out[b * ocap + off + r] = k;
                    out_src[b * ocap + off + r] = q;
                } else {
                    cnt[SA_CNT_OVERFLOW] = 1;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[SA_CNT_KEYS] = min(carry_s, ocap);
}

// ------------------------------------------------------------- descriptor --
// lib_description.c:278-347 (accumulation) and :374-389 (clip, renormalise, quantise),
// driver loop lib_sift_anatomy.c:724-765.
//
// The reference adds every sample's (up to 2x2 cells) x (2 orientation bins) contributions
// into float bins in raster order of the samples.  A bin receives at most one contribution per
// sample, so keeping the contributions of a bin in sample order and adding them one after the
// other gives the reference's bits without serialising the whole patch.
#define SA_MAX_CELLS 64

// One 64-lane wave owns a keypoint, so no block-wide barriers are needed, and the samples that
// fall inside the rotated window are first queued (in raster order) so that the expensive
// double-precision part always runs on full waves.  Lane l owns the BPL descriptor bins
// l, l+64, l+128, ...: a batch of 64 raster-consecutive samples touches only one or two rows of
// cells, i.e. a contiguous range of bins, and this spreads that range over all lanes.
// array_l2norm (lib_util.c:156-164) over LDS by one lane: sequential float sum, four values
// per LDS read.
__device__ __forceinline__ float seq_l2norm(const float* d, int n)
{
    float l2 = 0.f;
    int k = 0;
#pragma unroll 8
    for (; k + 4 <= n; k += 4) {
        const v4f v = ld4(d + k);
        l2 = __fadd_rn(l2, __fmul_rn(v.x, v.x));
        l2 = __fadd_rn(l2, __fmul_rn(v.y, v.y));
        l2 = __fadd_rn(l2, __fmul_rn(v.z, v.z));
        l2 = __fadd_rn(l2, __fmul_rn(v.w, v.w));
    }
    for (; k < n; ++k) l2 = __fadd_rn(l2, __fmul_rn(d[k], d[k]));
    return (float)sqrt((double)l2);
}

// Ballots into lanes: lane L of (lo, hi) receives mask m[L].  A ballot lives in an SGPR pair and
// v_writelane_b32 drops each half into one lane of a VGPR.  Two hazards that the compiler does not
// handle inside asm statements (measured with tools/membench/writelane_test.hip): v_writelane must
// not read an SGPR in the cycle after the v_cmp that wrote it, hence all compares first, then wait
// states, then the writes; and a DPP read of the VGPR needs two wait states after its last write.
#define SA_WL(lo, hi, m, L)                                                                           \
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((int)(unsigned)(m)), "n"(L));          \
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((int)(unsigned)((m) >> 32)), "n"(L));
__device__ __forceinline__ void masks_to_lanes8(int key, int& lo, int& hi)
{
    unsigned long long m[8];
#pragma unroll
    for (int L = 0; L < 8; ++L) m[L] = __ballot(key == L);
    asm volatile("s_nop 4" :: "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(m[4]), "s"(m[5]), "s"(m[6]), "s"(m[7]));
    SA_WL(lo, hi, m[0], 0) SA_WL(lo, hi, m[1], 1) SA_WL(lo, hi, m[2], 2) SA_WL(lo, hi, m[3], 3)
    SA_WL(lo, hi, m[4], 4) SA_WL(lo, hi, m[5], 5) SA_WL(lo, hi, m[6], 6) SA_WL(lo, hi, m[7], 7)
    asm volatile("s_nop 1" : "+v"(lo), "+v"(hi));
}
// the same for keys -1..3 of two variables at once (lane L: key == L - 1)
__device__ __forceinline__ void masks_to_lanes5x2(int ka, int kb, int& alo, int& ahi, int& blo, int& bhi)
{
    unsigned long long ma[5], mb[5];
#pragma unroll
    for (int L = 0; L < 5; ++L) { ma[L] = __ballot(ka == L - 1); mb[L] = __ballot(kb == L - 1); }
    asm volatile("s_nop 4" :: "s"(ma[0]), "s"(ma[1]), "s"(ma[2]), "s"(ma[3]), "s"(ma[4]),
                 "s"(mb[0]), "s"(mb[1]), "s"(mb[2]), "s"(mb[3]), "s"(mb[4]));
    SA_WL(alo, ahi, ma[0], 0) SA_WL(alo, ahi, ma[1], 1) SA_WL(alo, ahi, ma[2], 2) SA_WL(alo, ahi, ma[3], 3)
    SA_WL(alo, ahi, ma[4], 4)
    SA_WL(blo, bhi, mb[0], 0) SA_WL(blo, bhi, mb[1], 1) SA_WL(blo, bhi, mb[2], 2) SA_WL(blo, bhi, mb[3], 3)
    SA_WL(blo, bhi, mb[4], 4)
    asm volatile("s_nop 1" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi));
}
#undef SA_WL

#ifdef SA_DESCR_PROF
__device__ unsigned long long sa_prof[16];
extern "C" void sift_amd_debug_prof(unsigned long long* out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(sa_prof), sizeof(sa_prof));
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(sa_prof), z, sizeof(z)); }
}
#define PROF_T(k) { const long long t_ = clock64(); pacc[k] += t_ - tprof; tprof = clock64(); }
#else
#define PROF_T(k)
#endif
// NH, NB: descriptor shape known at compile time (0 = taken from the parameters at run time);
// the mask loops over cell rows/columns and orientation bins then unroll.
template <int BPL, int NH = 0, int NB = 0>
__device__ __forceinline__ void
descriptor_wave_body(const SaPyramid& P, const SaScalars& sc, const SaKey* __restrict__ keys, const int* __restrict__ counters,
                     int cnt_slot, int kcap, SaStd* __restrict__ out_std, float* __restrict__ out_descr)
{
    constexpr int POOL = 256;
    // Contribution pool, used in one of two ways.  Per-bin lists: up to 512 contributions, every
    // list padded to an even length (64*BPL more slots), plus slack because the walk reads one
    // pair ahead.  Per-cell lists: POOL+1 entries of two contributions (bins k0 / k0+1) and the
    // orientation bin k0 of each.
    constexpr int POOL_LISTS = (512 + 64 * BPL + 4) * 8, POOL_CELLS = (POOL + 1) * 16 + (POOL + 1) * 4 + 4;
    __shared__ __attribute__((aligned(16))) unsigned char pool_raw[POOL_LISTS > POOL_CELLS ? POOL_LISTS : POOL_CELLS];
    double (*p_c)[2] = reinterpret_cast<double(*)[2]>(pool_raw);
    int* p_k = reinterpret_cast<int*>(pool_raw + (POOL + 1) * 16);
    __shared__ int q_pix[128];
    __shared__ int cstart[SA_MAX_CELLS + 1];
    // per-bin lists (n_ori <= 64): list starts, and per batch the lane masks of every pair of
    // adjacent orientation bins, cell rows and cell columns
    __shared__ int bstart_s[SA_MAX_DESCR];
    __shared__ unsigned long long kk_s[64], pa_s[16], pb_s[16];
    __shared__ __attribute__((aligned(16))) float d_s[SA_MAX_DESCR];
    __shared__ float norm_s;
    const int lane = threadIdx.x;
    const long long b = blockIdx.y;
    const int n = min(counters[b * SA_CNT_STRIDE + cnt_slot], kcap);
    const int Nh = NH ? NH : sc.n_hist, Nb = NB ? NB : sc.n_ori, nd = Nh * Nh * Nb, ncell = Nh * Nh;
    const unsigned long long lt = lanemask_lt();
    int my_t[BPL], my_cell[BPL], my_ci[BPL], my_cj[BPL], my_k[BPL], my_kprev[BPL];
#pragma unroll
    for (int u = 0; u < BPL; ++u) {
        // Default shape: a lane's second bin lies two cell rows further down and four orientation
        // bins away.  Orientations are measured from the keypoint's dominant one, so the bins around
        // zero collect the longest lists in every cell; pairing bin k with bin k+4 evens out what the
        // lanes have to add.
        const int t = (NB == 8 && BPL == 2 && u == 1) ? 64 + (lane & ~7) + ((lane + 4) & 7) : lane + 64 * u;
        my_t[u] = t;
        my_cell[u] = t < nd ? t / Nb : -1;
        my_ci[u] = t < nd ? my_cell[u] / Nh : 0;
        my_cj[u] = t < nd ? my_cell[u] - my_ci[u] * Nh : 0;
        my_k[u] = t % Nb;
        my_kprev[u] = my_k[u] == 0 ? Nb - 1 : my_k[u] - 1;
    }
    // all bins of a lane lie in one cell
    const bool one_cell = BPL == 1 && !(sc.debug & 1);
    // One list per descriptor bin instead of one per cell: a lane then touches only the
    // contributions of its own bins.  Needs a lane per orientation bin for the bookkeeping.
    const bool bin_lists = NB ? true : (Nb <= 64 && !(sc.debug & 2));   // (fixed shapes: launched without debug switches)
    double* pool = reinterpret_cast<double*>(pool_raw);
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
#ifdef SA_DESCR_PROF
        long long pacc[12] = {0};
        long long tprof = clock64();
#endif
        const SaKey key = keys[b * kcap + q];
        const SaOctave& oc = P.oct[key.o];
        const int w = oc.w, h = oc.h;
        const float* im = oc.gss + b * oc.gss_stride + (long long)key.s * w * h;
        const float xk = __fdiv_rn(key.x, oc.delta), yk = __fdiv_rn(key.y, oc.delta);
        const float sk = __fdiv_rn(key.sigma, oc.delta);
        const float theta = key.theta;
        const float R = __fmul_rn(__fmul_rn(__fadd_rn(1.f, __fdiv_rn(1.f, (float)Nh)), sc.lambda_descr), sk);
        const float Rp = (float)(SA_SQRT2 * (double)R);
        const int i0 = max(0, (int)((double)__fsub_rn(xk, Rp) + 0.5));
        const int j0 = max(0, (int)((double)__fsub_rn(yk, Rp) + 0.5));
        const int i1 = min((int)((double)__fadd_rn(xk, Rp) + 0.5), h - 1);
        const int j1 = min((int)((double)__fadd_rn(yk, Rp) + 0.5), w - 1);
        const int nw = j1 - j0, nh = i1 - i0;                 // upper bounds are exclusive
        const int npix = (nw > 0 && nh > 0) ? nw * nh : 0;
        const float ma = -theta;
        const float ct = (float)cos((double)ma), st = (float)sin((double)ma);
        const double tt = (double)__fmul_rn(sc.lambda_descr, sk);
        const double two_t2 = 2 * tt * tt;
        const float cell = __fdiv_rn(__fmul_rn(__fmul_rn(2.f, sc.lambda_descr), sk), (float)Nh);
        // Correctly rounded quotients by a denominator that is the same for the whole keypoint:
        // with y = RN(1/b), q = RN(a*y), r = a - b*q (exact in an FMA), RN(q + r*y) is RN(a/b)
        // (Markstein's theorem; operands here are far from the overflow and underflow ranges).
        const float cell_rcp = __fdiv_rn(1.f, cell);
        const double two_t2_rcp = 1.0 / two_t2;
        const double half = ((double)Nh - 1.0) / 2.0;
        float acc[BPL];
#pragma unroll
        for (int u = 0; u < BPL; ++u) acc[u] = 0.f;

        int qn = 0, p0 = 0, qhead = 0;     // q_pix is a ring of 128 entries
        // (row, column) inside the bounding box of raster position p0 + lane, advanced by 64
        // positions per round without divisions
        int fi = 0, fj = 0;
        if (npix > 0) { fi = lane / nw; fj = lane - fi * nw; }
        PROF_T(0)
        while (p0 < npix || qn > 0) {
            // ---- queue the samples inside the rotated window, in raster order
            while (qn < 64 && p0 < npix) {
                bool in_win = false;
                if (p0 + lane < npix) {
                    const float X0 = __fsub_rn((float)(i0 + fi), xk), Y0 = __fsub_rn((float)(j0 + fj), yk);
                    const float X = __fsub_rn(__fmul_rn(ct, X0), __fmul_rn(st, Y0));
                    const float Y = __fadd_rn(__fmul_rn(st, X0), __fmul_rn(ct, Y0));
                    in_win = fmaxf(fabsf(X), fabsf(Y)) < R;
                }
                const unsigned long long m = __ballot(in_win);
                if (in_win) q_pix[(qhead + qn + __popcll(m & lt)) & 127] = (fi << 16) | fj;
                qn += __popcll(m);
                p0 += 64;
                fj += 64;
                while (fj >= nw) { fj -= nw; ++fi; }
            }
            __syncthreads();
            PROF_T(1)
            const int nproc = min(qn, 64);
            const bool act = lane < nproc;
            // ---- geometry of my sample, cell ranks
            int si = 0, sj = 0, a0 = -2, b0 = -2;
            float X = 0.f, Y = 0.f, al = 0.f, be = 0.f;
            if (act) {
                const int p = q_pix[(qhead + lane) & 127];
                si = i0 + (p >> 16); sj = j0 + (p & 0xffff);
                const float X0 = __fsub_rn((float)si, xk), Y0 = __fsub_rn((float)sj, yk);
                X = __fsub_rn(__fmul_rn(ct, X0), __fmul_rn(st, Y0));
                Y = __fadd_rn(__fmul_rn(st, X0), __fmul_rn(ct, Y0));
                al = (float)((double)div_by_uniform(X, cell, cell_rcp) + half);
                be = (float)((double)div_by_uniform(Y, cell, cell_rcp) + half);
                a0 = (int)floor((double)al);
                b0 = (int)floor((double)be);
            }
            PROF_T(2)
            if (bin_lists) {
                // ---- my sample's orientation bin pair, weights and magnitude
                int k0 = -1, k1 = 0;
                double wl = 0., wr = 0., M = 0.;
                if (act) {
                    const double dx = (double)grad_i(im, w, h, si, sj);
                    const double dy = (double)grad_j(im, w, h, si, sj);
                    float ori = (float)(atan2(dy, dx) - (double)theta);
                    ori = modulus_2pi(ori);
                    const float r2 = __fadd_rn(__fmul_rn(X, X), __fmul_rn(Y, Y));
                    // dx, dy are floats: their squares are exact in double, so the root of the sum
                    // is within an ulp of hypot() without its scaling steps
                    M = sqrt(dx * dx + dy * dy) * exp(div_by_uniform((double)(-r2), two_t2, two_t2_rcp));
                    const float ga = (float)(div_by_uniform((double)ori, 2 * SA_PI, 1.0 / (2 * SA_PI)) * Nb);
                    const double gf = floor((double)ga);
                    k0 = wrap_bin((int)ga, Nb);
                    k1 = k0 + 1 == Nb ? 0 : k0 + 1;
                    wl = 1. - ((double)ga - gf);
                    wr = 1.0 - (gf + 1 - (double)ga);
                }
                PROF_T(3)
                // ---- kk_s[k]: lanes whose sample adds to orientation bin k (its own bin k0 == k,
                // or k0 == k-1 through the second weight); lane k collects the mask of k0 == k
                unsigned long long myK;
                {
                    // (inactive lanes hold k0 = -1; a ballot lives in an SGPR pair and v_writelane
                    // drops it into lane k)
                    int lo = 0, hi = 0;
                    if (NB == 8) {
                        masks_to_lanes8(k0, lo, hi);
                    } else {
                        for (int k = 0; k < Nb; ++k) {
                            const unsigned long long m = __ballot(k0 == k);
                            if (lane == k) { lo = (int)(unsigned)m; hi = (int)(unsigned)(m >> 32); }
                        }
                    }
                    myK = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
                }
                {
                    // mask of the bin before mine (bin Nb-1 for bin 0)
                    unsigned long long prevK = lane_prev_u64(myK);
                    const unsigned lastLo = __builtin_amdgcn_readlane((int)(unsigned)myK, Nb - 1);
                    const unsigned lastHi = __builtin_amdgcn_readlane((int)(unsigned)(myK >> 32), Nb - 1);
                    if (lane == 0) prevK = ((unsigned long long)lastHi << 32) | lastLo;
                    if (lane < Nb) kk_s[lane] = myK | prevK;
                }
                PROF_T(4)
                // ---- which samples touch which cell: a sample with corner (a0, b0) touches the cell
                // rows a0, a0+1 and columns b0, b0+1, so the mask of cell (ci, cj) is
                // (samples with a0 in {ci-1, ci}) & (samples with b0 in {cj-1, cj}).  Lane p collects
                // the mask of a0 == p-1 (p = 0..Nh), pa_s[p] is the pair mask of cell row p-1.
                {
                    int alo = 0, ahi = 0, blo = 0, bhi = 0;
                    if (NH == 4) {
                        // (inactive lanes hold a0 = b0 = -2)
                        masks_to_lanes5x2(a0, b0, alo, ahi, blo, bhi);
                    } else {
                        for (int a = -1; a < Nh; ++a) {
                            const unsigned long long ma_ = __ballot(a0 == a), mb_ = __ballot(b0 == a);
                            if (lane == a + 1) {
                                alo = (int)(unsigned)ma_; ahi = (int)(unsigned)(ma_ >> 32);
                                blo = (int)(unsigned)mb_; bhi = (int)(unsigned)(mb_ >> 32);
                            }
                        }
                    }
                    const unsigned long long tA = ((unsigned long long)(unsigned)ahi << 32) | (unsigned)alo;
                    const unsigned long long tB = ((unsigned long long)(unsigned)bhi << 32) | (unsigned)blo;
                    const unsigned long long pA = lane_prev_u64(tA), pB = lane_prev_u64(tB);
                    if (lane <= Nh + 1) { pa_s[lane] = tA | pA; pb_s[lane] = tB | pB; }
                }
                __syncthreads();
                PROF_T(5)
                // ranks of my two contributions inside every cell I touch (8 bits per slot)
                int rankL = 0, rankR = 0;
                if (act) {
                    const unsigned long long mA = kk_s[k0], mB = kk_s[k1];
                    const unsigned long long ra0 = pa_s[a0 + 1], ra1 = pa_s[a0 + 2];
                    const unsigned long long rb0 = pb_s[b0 + 1], rb1 = pb_s[b0 + 2];
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const unsigned long long m = ((sl >> 1) ? ra1 : ra0) & ((sl & 1) ? rb1 : rb0) & lt;
                        rankL |= __popcll(m & mA) << (8 * sl);
                        rankR |= __popcll(m & mB) << (8 * sl);
                    }
                }
                PROF_T(6)
                // ---- list length of my bins, list starts by a scan in bin order
                int cnt[BPL], start[BPL], tot = 0;
#pragma unroll
                for (int u = 0; u < BPL; ++u) {
                    cnt[u] = my_cell[u] >= 0 ? __popcll(pa_s[my_ci[u] + 1] & pb_s[my_cj[u] + 1] & kk_s[my_k[u]]) : 0;
                    start[u] = tot;
                    tot += (cnt[u] + 1) & ~1;      // lists start at even slots
                }
                {
                    const int excl = wave_scan_inclusive(tot) - tot;
#pragma unroll
                    for (int u = 0; u < BPL; ++u) {
                        start[u] += excl;
                        if (my_cell[u] >= 0) bstart_s[my_t[u]] = start[u];
                        // odd lists end with +0.0, which leaves the non-negative accumulator as it is
                        if (cnt[u] & 1) pool[start[u] + cnt[u]] = 0.0;
                    }
                }
                __syncthreads();
                PROF_T(7)
                // ---- my contributions go to their places in the lists
                if (act) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int ci = a0 + (sl >> 1), cj = b0 + (sl & 1);
                        if ((unsigned)ci < (unsigned)Nh && (unsigned)cj < (unsigned)Nh) {
                            const float da = fabsf(__fsub_rn((float)ci, al));
                            const float db = fabsf(__fsub_rn((float)cj, be));
                            const double wa = 1.0 - (double)da, wb = 1.0 - (double)db;
                            const int base = (ci * Nh + cj) * Nb;
                            pool[bstart_s[base + k0] + ((rankL >> (8 * sl)) & 0xff)] = wl * wa * wb * M;
                            pool[bstart_s[base + k1] + ((rankR >> (8 * sl)) & 0xff)] = wr * wa * wb * M;
                        }
                    }
                }
                __syncthreads();
                PROF_T(8)
                // ---- every lane adds the lists of its bins, in order.  A lane's lists lie one after
                // the other in the pool, so they are walked as one sequence of pairs (two entries per
                // 16-byte read, the next pair loaded while the current one is added) whose
                // accumulator changes at the list boundaries: the wave is busy for the longest
                // per-lane total, not for the sum of the longest lists.
                {
                    typedef double v2d __attribute__((ext_vector_type(2)));
                    const v2d* pl = reinterpret_cast<const v2d*>(pool) + (start[0] >> 1);
                    int bound[BPL];
                    int total = 0;
#pragma unroll
                    for (int u = 0; u < BPL; ++u) { total += (cnt[u] + 1) >> 1; bound[u] = total; }
#ifdef SA_DESCR_PROF
                    {   // walk statistics: batches, longest and summed per-lane walks, samples
                        int mx = total, sm = total;
                        for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, __shfl_xor(mx, d)); sm += __shfl_xor(sm, d); }
                        if (lane == 0) {
                            atomicAdd(&sa_prof[12], 1ull); atomicAdd(&sa_prof[13], (unsigned long long)mx);
                            atomicAdd(&sa_prof[14], (unsigned long long)sm); atomicAdd(&sa_prof[15], (unsigned long long)nproc);
                        }
                    }
#endif
                    v2d v = pl[0];
                    for (int e = 0; e < total; ++e) {
                        const v2d nv = pl[e + 1];
                        int sel = 0;
#pragma unroll
                        for (int u = 0; u + 1 < BPL; ++u) sel += e >= bound[u];
                        float a = acc[0];
#pragma unroll
                        for (int u = 1; u < BPL; ++u) a = sel == u ? acc[u] : a;
                        a = (float)((double)a + v.x);
                        a = (float)((double)a + v.y);
#pragma unroll
                        for (int u = 0; u < BPL; ++u) acc[u] = sel == u ? a : acc[u];
                        v = nv;
                    }
                }
            } else {
                // Rank of my sample in the list of each cell it touches (8 bits per slot, ranks < 64)
                // and the start of every list: cells are visited in order, lane c remembers where
                // list c starts, the running total is wave-uniform.
                int rank_pack = 0, my_start = 0, run = 0;
                for (int c = 0, ci = 0, cj = 0; c < ncell; ++c, ++cj) {
                    if (cj == Nh) { cj = 0; ++ci; }
                    const int di = ci - a0, dj = cj - b0;
                    const bool f = act && (unsigned)di < 2u && (unsigned)dj < 2u;
                    const unsigned long long m = __ballot(f);
                    if (lane == c) my_start = run;
                    if (f) rank_pack |= __popcll(m & lt) << (8 * (di * 2 + dj));
                    run += __popcll(m);
                }
                if (lane < ncell) cstart[lane] = my_start;
                if (lane == 0) cstart[ncell] = run;
                __syncthreads();
                // ---- contributions of my sample, appended to the lists of the cells it touches
                if (act) {
                    const double dx = (double)grad_i(im, w, h, si, sj);
                    const double dy = (double)grad_j(im, w, h, si, sj);
                    float ori = (float)(atan2(dy, dx) - (double)theta);
                    ori = modulus_2pi(ori);
                    const float r2 = __fadd_rn(__fmul_rn(X, X), __fmul_rn(Y, Y));
                    const double M = hypot(dx, dy) * exp((double)(-r2) / two_t2);
                    const float ga = (float)((double)ori / (2 * SA_PI) * Nb);
                    const double gf = floor((double)ga);
                    const int k0 = wrap_bin((int)ga, Nb);
                    const double wl = 1. - ((double)ga - gf);
                    const double wr = 1.0 - (gf + 1 - (double)ga);
    #pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int ci = a0 + (sl >> 1), cj = b0 + (sl & 1);
                        if ((unsigned)ci < (unsigned)Nh && (unsigned)cj < (unsigned)Nh) {
                            const float da = fabsf(__fsub_rn((float)ci, al));
                            const float db = fabsf(__fsub_rn((float)cj, be));
                            const double wa = 1.0 - (double)da, wb = 1.0 - (double)db;
                            const int pos = cstart[ci * Nh + cj] + ((rank_pack >> (8 * sl)) & 0xff);
                            p_k[pos] = k0;
                            p_c[pos][0] = wl * wa * wb * M;
                            p_c[pos][1] = wr * wa * wb * M;
                        }
                    }
                }
                __syncthreads();
                // ---- bin owners consume their cells' lists in order
                if (one_cell) {
                    // Branch-free walk, next entry loaded while the current one is added.  An entry
                    // adds to at most one of a bin's two sources; the others add +0.0, which leaves
                    // the (non-negative) float accumulator unchanged bit for bit.
                    const int c = my_cell[0];
                    int e = 0, e1 = 0;
                    if (c >= 0) { e = cstart[c]; e1 = cstart[c + 1]; }
                    int kk = p_k[e];
                    double vL = p_c[e][0], vR = p_c[e][1];
                    while (e < e1) {
                        const int nk = p_k[e + 1];
                        const double nL = p_c[e + 1][0], nR = p_c[e + 1][1];
    #pragma unroll
                        for (int v = 0; v < BPL; ++v) {
                            const double add = kk == my_k[v] ? vL : (kk == my_kprev[v] ? vR : 0.0);
                            acc[v] = (float)((double)acc[v] + add);
                        }
                        kk = nk; vL = nL; vR = nR;
                        ++e;
                    }
                } else {
    #pragma unroll
                    for (int u = 0; u < BPL; ++u) {
                        if (u > 0 && my_cell[u] == my_cell[u - 1]) continue;   // handled with bin u-1
                        if (my_cell[u] < 0) continue;
                        const int e1 = cstart[my_cell[u] + 1];
                        for (int e = cstart[my_cell[u]]; e < e1; ++e) {
                            const int kk = p_k[e];
                            const double vL = p_c[e][0], vR = p_c[e][1];
    #pragma unroll
                            for (int v = u; v < BPL; ++v) {
                                if (my_cell[v] != my_cell[u]) break;
                                if (kk == my_k[v]) acc[v] = (float)((double)acc[v] + vL);
                                else if (kk == my_kprev[v]) acc[v] = (float)((double)acc[v] + vR);
                            }
                        }
                    }
                }
            }
            PROF_T(9)
            // ---- pop the processed samples from the queue
            qhead = (qhead + nproc) & 127;
            qn -= nproc;
            __syncthreads();
            PROF_T(10)
        }
#pragma unroll
        for (int u = 0; u < BPL; ++u)
            if (my_t[u] < nd) d_s[my_t[u]] = acc[u];
        __syncthreads();
        // clip at 0.2*norm, renormalise; norms are sequential float sums as in array_l2norm
        if (lane == 0) norm_s = seq_l2norm(d_s, nd);
        __syncthreads();
        const float cap = __fmul_rn(0.2f, norm_s);
#pragma unroll
        for (int u = 0; u < BPL; ++u) {
            acc[u] = fminf(acc[u], cap);
            if (my_t[u] < nd) d_s[my_t[u]] = acc[u];
        }
        __syncthreads();
        if (lane == 0) norm_s = seq_l2norm(d_s, nd);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BPL; ++u) {
            const int t = my_t[u];
            if (t < nd) {
                float v = (float)(int)((double)acc[u] * 512.0 / (double)norm_s);
                v = v < 255.f ? v : 255.f;
                if (out_descr) out_descr[(b * kcap + q) * SA_MAX_DESCR + t] = v;
                if (out_std && t < 128) out_std[b * kcap + q].descriptor[t] = (unsigned char)v;
            }
        }
        // the 144-byte record has room for 128 bins: shorter descriptors are zero-padded, longer
        // ones are cut there (the float descriptors hold every bin)
        if (out_std && nd < 128)
            for (int t = nd + lane; t < 128; t += 64) out_std[b * kcap + q].descriptor[t] = 0;
        if (out_std && lane == 0) {
            SaStd* o = &out_std[b * kcap + q];
            o->x = key.x; o->y = key.y; o->scale = key.sigma; o->orientation = key.theta;
        }
        __syncthreads();
        PROF_T(11)
#ifdef SA_DESCR_PROF
        if (threadIdx.x < 12) {
            long long v = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) if (threadIdx.x == k) v = pacc[k];
            atomicAdd(&sa_prof[threadIdx.x], (unsigned long long)v);
        }
#endif
    }
}

template <int BPL>
__global__ void __launch_bounds__(64)
k_descriptor_wave(SaPyramid P, SaScalars sc, const SaKey* __restrict__ keys, const int* __restrict__ counters,
                  int cnt_slot, int kcap, SaStd* __restrict__ out_std, float* __restrict__ out_descr)
{
    descriptor_wave_body<BPL>(P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr);
}

// The default 4 x 4 x 8 shape.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_descriptor_wave_448(SaPyramid P, SaScalars sc, const SaKey* __restrict__ keys, const int* __restrict__ counters,
                      int cnt_slot, int kcap, SaStd* __restrict__ out_std, float* __restrict__ out_descr)
{
    descriptor_wave_body<2, 4, 8>(P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr);
}

// Output records of a batch, image after image without gaps, so that the host gather is one
// copy.  Block (x, b) copies a slice of image b's records to its offset (sum of the counts of
// the images before it); 144-byte records move as 36 dwords.
__global__ void __launch_bounds__(256)
k_pack_std(const SaStd* __restrict__ std_recs, const int* __restrict__ counters, int cnt_slot, int kcap,
           SaStd* __restrict__ packed)
{
    const int b = blockIdx.y;
    int off = 0;
    for (int k = 0; k < b; ++k) off += min(counters[k * SA_CNT_STRIDE + cnt_slot], kcap);
    const int n = min(counters[b * SA_CNT_STRIDE + cnt_slot], kcap);
    const unsigned* src = reinterpret_cast<const unsigned*>(std_recs + (long long)b * kcap);
    unsigned* dst = reinterpret_cast<unsigned*>(packed + off);
    const long long words = (long long)n * (sizeof(SaStd) / 4);
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < words; t += (long long)gridDim.x * 256)
        dst[t] = src[t];
}

// ================================================================ launchers ==
#define HIPK(...) do { hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace sa {

// Which route the blur launches of this process took (tests assert that the route they mean to
// exercise really ran; a hipGraph replay launches nothing and counts nothing).
static long long g_launch_stats[SIFT_AMD_LS_COUNT];
static void ls_add(int k, long long v = 1) { __atomic_fetch_add(&g_launch_stats[k], v, __ATOMIC_RELAXED); }
static void ls_max(int k, long long v)
{
    long long cur = __atomic_load_n(&g_launch_stats[k], __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(&g_launch_stats[k], &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
void launch_stats(long long* out, int reset)
{
    for (int k = 0; k < SIFT_AMD_LS_COUNT; ++k)
        out[k] = reset ? __atomic_exchange_n(&g_launch_stats[k], 0, __ATOMIC_RELAXED)
                       : __atomic_load_n(&g_launch_stats[k], __ATOMIC_RELAXED);
}

#define SA_R_CASES(F) \
    case 1: F(1); break; case 2: F(2); break; case 3: F(3); break; case 4: F(4); break; \
    case 5: F(5); break; case 6: F(6); break; case 7: F(7); break; case 8: F(8); break; \
    case 9: F(9); break; case 10: F(10); break; case 11: F(11); break; case 12: F(12); break; \
    case 13: F(13); break; case 14: F(14); break; case 15: F(15); break; case 16: F(16); break;

template <int R, int EW, int NBUF>
static void blur_fused_launch(hipStream_t st, const BlurArgs& a)
{
    constexpr int TW = EW - 32;
    // strip height: tall strips amortise the 2R-row prologue, but every block walks its strip
    // sequentially, so small octaves get short strips to keep enough blocks in flight
    const char* fs = getenv("SIFT_AMD_STRIP");          // forces a strip height (tests, A/B runs)
    const int forced = fs ? atoi(fs) : 0;
    const bool sparse = a.need_cells && a.ovs_w == 0 && !a.dog && !a.flags && (a.w & 3) == 0;
    // sparse launches: a block walks only from its first to its last needed group of rows, so tall
    // strips cost no prologue where nothing is needed
    int strip = EW == 512 ? 256 : (sparse ? 256 : 64);
    const long long cols = (a.w + TW - 1) / TW;
    while (strip > 16 && cols * ((a.h + strip - 1) / strip) * a.batch * (EW / 256) < 2048) strip >>= 1;
    if (EW == 512 && a.ovs_w == 0) {
        // plain wide launches: the image cut into equal strips of about 512 rows (measured for 128 1080p
        // images, R = 10 / 13: 1.93 / 2.19 ms with 128-row strips, 1.87 / 2.14 ms with 544), more of them
        // while there are too few blocks
        int n = (a.h + 256) / 512 > 0 ? (a.h + 256) / 512 : 1;
        auto rows = [&](int parts) { return (((a.h + parts - 1) / parts) + 7) & ~7; };
        if (sparse) while (rows(n) > 512) ++n;        // one bit per 8-row group of a strip
        while (rows(n) > 16 && cols * ((a.h + rows(n) - 1) / rows(n)) * a.batch * 2 < 2048) n *= 2;
        strip = rows(n);
    }
    if (forced > 0) strip = (forced + 7) & ~7;     // the kernel relies on strips of whole 8-row groups
    if (sparse) {
        if (strip > 512) strip = 512;
        ls_add(SIFT_AMD_LS_SPARSE);
    }
    const int nbx = (int)cols, nby = (a.h + strip - 1) / strip, nblocks = nbx * nby * a.batch;
    if (EW == 512 && a.ovs_w == 0 && strip >= 64) {      // the kernel's two-iterations-ahead branch
        ls_add(SIFT_AMD_LS_WIDE_AHEAD2);
        ls_max(SIFT_AMD_LS_WIDE_AHEAD2_STRIP_MAX, strip);
    } else {
        ls_add(EW == 512 ? SIFT_AMD_LS_WIDE_OTHER : SIFT_AMD_LS_NARROW);
    }
    float* out = a.out;
    SaTapPairs tp2;
    for (int k = 0; k <= SA_MAX_RADIUS_FAST; ++k) { tp2.g[k].x = a.taps.g[k]; tp2.g[k].y = a.taps.g[k]; }
#define SA_BLUR_LAUNCH_S(OVS_, VW_, PREV_, SPARSE_, ...)                                                      \
    HIPK((k_blur_fused<R, OVS_, EW, NBUF, VW_, PREV_, SPARSE_>), dim3(nblocks), dim3(EW), (OVS_ || PREV_ ? 0 : pad_np), st, a.in, out, a.dog, a.sub, a.w, a.h, \
         a.s_in, a.s_out, a.s_dog, a.s_sub, tp2, strip, nbx, nby, nblocks, __VA_ARGS__, a.flags, a.flag_thr, a.need_cells)
#define SA_BLUR_LAUNCH(OVS_, VW_, PREV_, ...) SA_BLUR_LAUNCH_S(OVS_, VW_, PREV_, false, __VA_ARGS__)
    const bool vw = (a.w & 3) == 0;
    const bool prev = a.dog || a.flags;
    // Occupancy (measured, 128 1080p images): the seed launch, whose loads interpolate, runs 12 % faster
    // with three blocks per CU, which it gets without the centre-sample buffer (1.48 -> 1.29 ms);
    // the plain launches are fastest with two (R = 5: 1.62 ms, 1.69 with three, 1.76 with four,
    // 1.68 with one), so those without the buffer reserve its size as unused dynamic LDS.
    constexpr int pad_np = NBUF * 8 * (EW + 32) * (int)sizeof(float);
    if (a.ovs_w > 0 && !prev) {         // (the seed image has no DoG)
        if (vw) SA_BLUR_LAUNCH(true, true, false, a.ovs_w, a.ovs_h, a.ovs_delta);
        else SA_BLUR_LAUNCH(true, false, false, a.ovs_w, a.ovs_h, a.ovs_delta);
    } else if (prev) {
        if (vw) SA_BLUR_LAUNCH(false, true, true, 0, 0, 0.f);
        else SA_BLUR_LAUNCH(false, false, true, 0, 0, 0.f);
    } else {
        if (vw && sparse) SA_BLUR_LAUNCH_S(false, true, false, true, 0, 0, 0.f);
        else if (vw) SA_BLUR_LAUNCH(false, true, false, 0, 0, 0.f);
        else SA_BLUR_LAUNCH(false, false, false, 0, 0, 0.f);
    }
#undef SA_BLUR_LAUNCH
#undef SA_BLUR_LAUNCH_S
}

template <int R>
static void blur_fused_t(hipStream_t st, const BlurArgs& a)
{
    // Tiles of 480 outputs (512 columns with the halo, 8 waves) or 224 outputs (256 columns,
    // 4 waves): whichever covers the image width with fewer columns in total.  The wide tile has
    // the smaller halo share and fewer idle row-pass slots (~5 % on octave 0 of a 1080p image) but
    // wastes most of a tile on narrow octaves.  SIFT_AMD_WIDE_MIN=<width> forces the choice by width.
    const int wide_min = getenv("SIFT_AMD_WIDE_MIN") ? atoi(getenv("SIFT_AMD_WIDE_MIN")) : 0;
    bool wide = wide_min > 0 ? a.w >= wide_min
                             : ((a.w + 479) / 480) * 512 < ((a.w + 223) / 224) * 256;
    // Launches that compute only the needed parts decide per 8-row group of a tile, so narrow tiles skip
    // more: on the wide octaves, where an image has regions without contrast, they are the better choice
    // (128 natural-like 1080p images, last image of octave 0: 1.33 ms against 1.59 with wide tiles, 2.2
    // computed everywhere; white noise, nothing to skip: 2.5 against 2.45); the small octaves are needed
    // almost everywhere on every workload.  SIFT_AMD_SPARSE_WIDE=1/0 forces wide/narrow tiles for them.
    if (a.need_cells) {
        if (const char* m = getenv("SIFT_AMD_SPARSE_WIDE")) wide = atoi(m) != 0;
        else if (a.w >= 1500) wide = false;
    }
    if (wide) blur_fused_launch<R, 512, 2>(st, a);
    else blur_fused_launch<R, 256, 2>(st, a);
}

// Single-kernel blur (column + row pass, DoG and optional subsample epilogue).
// Returns false if the radius has no register-window instantiation.
bool launch_blur_fused(hipStream_t st, const BlurArgs& a)
{
    if (a.radius < 1 || a.radius > SA_MAX_RADIUS_FAST) return false;
    switch (a.radius) {
#define F(R) blur_fused_t<R>(st, a)
        SA_R_CASES(F)
#undef F
    }
    return true;
}

// Two blurs in one launch (k_blur_pair).  Returns false when this pair of radii / image shape has
// no instantiation (`probe`: only that answer); the caller then issues the two blurs one after the other.
// Instantiations: the pairs of the default parameters that arithmetic does not bound and whose tiles
// keep 480 of 512 columns (seed 5 + 5, 5 + 7, 7 + 8).  Measured, 128 1080p images: radii 5 and 7 in
// one launch 2.73 ms against 1.64 + 1.74 ms apart; radii 10 and 13 (456 columns per tile) 4.97 ms
// against 2.09 + 2.38 apart, so that pair stays two launches.
bool launch_blur_pair(hipStream_t st, const BlurPairArgs& a, bool probe)
{
    if ((a.w & 3) || a.w < 900) return false;
    const bool ovs = a.ovs_w > 0;
    if (ovs && (a.flag_on_a || a.flag_on_b || a.sub)) return false;
    // Tall strips: every strip repeats the consumer's warm-up (32 rows for radius 7).  Measured for 128
    // 1080p images (octave 0 3840 x 2160): 1080 rows 2.72 ms, 544 rows 2.82, 272 rows 2.83, 136 rows
    // 3.02.  Small batches would leave CUs without a block: they take the two-launch route.
    // SIFT_AMD_PAIR_STRIP=<rows> forces the two-blur launch with that strip height (tests, A/B runs).
    const char* fs = getenv("SIFT_AMD_PAIR_STRIP");
    const int forced = fs ? atoi(fs) : 0;
    int strip = forced > 0 ? ((forced + 7) & ~7) : 1080;
    if (forced <= 0) {
        auto blocks = [&](int st_) { return (long long)((a.w + 479) / 480) * ((a.h + st_ - 1) / st_) * a.batch; };
        while (strip > 270 && blocks(strip) < 512) strip = ((strip / 2) + 7) & ~7;
        if (blocks(strip) < 256) return false;
    }
    SaTapPairs ta, tb;
    for (int k = 0; k <= SA_MAX_RADIUS_FAST; ++k) { ta.g[k] = v2f{a.taps_a.g[k], a.taps_a.g[k]}; tb.g[k] = v2f{a.taps_b.g[k], a.taps_b.g[k]}; }
    const bool fa = a.flags && a.flag_on_a, fb = a.flags && a.flag_on_b, sub = a.sub != nullptr;
#define SA_PAIR(RA, RB, FA, FB, OVS, SUB)                                                                     \
    do {                                                                                                      \
        if (probe) return true;                                                                               \
        constexpr int TWS = 512 - 2 * (((RA + 3) & ~3) + ((RB + 3) & ~3));                                    \
        const int nbx = (a.w + TWS - 1) / TWS, nby = (a.h + strip - 1) / strip, nblocks = nbx * nby * a.batch; \
        ls_add(OVS ? SIFT_AMD_LS_PAIR_SEED : SIFT_AMD_LS_PAIR);                                               \
        ls_max(SIFT_AMD_LS_PAIR_STRIP_MAX, strip);                                                            \
        HIPK((k_blur_pair<RA, RB, FA, FB, OVS, SUB>), dim3(nblocks), dim3(1024), 0, st, a.in, a.out_a, a.out_b, a.sub, \
             a.w, a.h, a.s_in, a.s_a, a.s_b, a.s_sub, ta, tb, strip, nbx, nby, nblocks, a.ovs_w, a.ovs_h,      \
             a.ovs_delta, a.flags, a.flag_thr);                                                               \
        return true;                                                                                          \
    } while (0)
    if (ovs) {
        if (a.radius_a == 5 && a.radius_b == 5) SA_PAIR(5, 5, false, false, true, false);
        return false;
    }
    if (a.radius_a == 5 && a.radius_b == 7 && !fa && !sub) {
        if (fb) SA_PAIR(5, 7, false, true, false, false);
        SA_PAIR(5, 7, false, false, false, false);
    }
    if (a.radius_a == 7 && a.radius_b == 8 && fa == fb) {
        if (fa && sub) SA_PAIR(7, 8, true, true, false, true);
        if (fa) SA_PAIR(7, 8, true, true, false, false);
        if (sub) SA_PAIR(7, 8, false, false, false, true);
        SA_PAIR(7, 8, false, false, false, false);
    }
#undef SA_PAIR
    return false;
}

// Any-radius form, two launches.  pass 0: column pass (in -> tmp); pass 1: row pass (tmp -> out
// [, dog = out - in]).
void launch_blur_pass(hipStream_t st, const BlurArgs& a, int pass)
{
    dim3 grid((a.w + 255) / 256, a.h, a.batch);
    if (pass == 0)
        HIPK(k_blur_v_any, grid, dim3(256), 0, st, a.in, a.tmp, a.w, a.h, a.s_in, a.s_tmp, a.taps_dev, a.radius);
    else
        HIPK(k_blur_h_any, grid, dim3(256), 0, st, a.tmp, a.in, a.out, a.dog, a.w, a.h,
             a.s_tmp, a.s_in, a.s_out, a.s_dog, a.taps_dev, a.radius);
}

void launch_clear_words(hipStream_t st, void* p, long long n_words)
{
    const long long n = (n_words + 3) & ~3ll;          // allocations are rounded up by the caller
    if (n <= 0) return;
    HIPK(k_clear_words, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (unsigned*)p, n);
}

void launch_oversample(hipStream_t st, const float* in, int wi, int hi, long long s_in,
                       float* out, int wo, int ho, long long s_out, float delta, int batch)
{
    dim3 grid((wo + 255) / 256, ho, batch);
    HIPK(k_oversample, grid, dim3(256), 0, st, in, wi, hi, s_in, out, wo, ho, s_out, delta);
}

void launch_ingest_u8(hipStream_t st, const unsigned char* in, int channels, float* out, long long n_px)
{
    const dim3 grid((unsigned)((n_px + 1023) / 1024)), block(256);
    if (channels == 1) hipLaunchKernelGGL(k_ingest_u8<1>, grid, block, 0, st, in, out, n_px);
    else if (channels == 3) hipLaunchKernelGGL(k_ingest_u8<3>, grid, block, 0, st, in, out, n_px);
    else hipLaunchKernelGGL(k_ingest_u8<4>, grid, block, 0, st, in, out, n_px);
}

void launch_subsample(hipStream_t st, const float* in, int wi, long long s_in,
                      float* out, int wo, int ho, long long s_out, int batch)
{
    dim3 grid((wo + 255) / 256, ho, batch);
    HIPK(k_subsample2, grid, dim3(256), 0, st, in, wi, s_in, out, wo, ho, s_out);
}

void launch_sub(hipStream_t st, const float* a, const float* b, float* out, long long n,
                long long s_a, long long s_b, long long s_out, int batch)
{
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks, 1, batch);
    HIPK(k_sub, grid, dim3(256), 0, st, a, b, out, n, s_a, s_b, s_out);
}

void launch_gradient(hipStream_t st, const float* im, float* gi, float* gj, int w, int h, int n_img)
{
    dim3 grid((w + 255) / 256, h, n_img);
    HIPK(k_gradient, grid, dim3(256), 0, st, im, gi, gj, w, h, n_img);
}

int launch_extrema(hipStream_t st, const float* gss, long long s_gss, int w, int h, int n_dog, int o,
                   float thr_pre, int emit_all, unsigned long long* ckey, float* cval,
                   int* counters, int cap, int batch, const int* flags)
{
    if (w < 3 || h < 3) return 0;
    static const int forced = getenv("SIFT_AMD_EXT_RPW") ? atoi(getenv("SIFT_AMD_EXT_RPW")) : 0;
    const int nseg = (w - 2 + 61) / 62;
    // tall strips amortise the two halo rows and keep a wave's loads streaming; small octaves
    // need short strips to have enough waves in flight
    int rpw = 128;
    while (rpw > 16 && (long long)nseg * ((h - 2 + rpw - 1) / rpw) * batch < 2500) rpw >>= 1;
    if (forced > 0) rpw = forced;
    if (rpw > 256) rpw = 256;   // the marked-cell mask of a strip has 64 bits (8 rows each)
    const int nbx = (nseg + 3) / 4, nby = (h - 2 + rpw - 1) / rpw, nblocks = nbx * nby * batch;
    dim3 grid(nblocks);
#define EXT(N) HIPK(k_extrema<N>, grid, dim3(256), 0, st, gss, s_gss, w, h, o, thr_pre, emit_all, rpw, ckey, cval, counters, cap, nbx, nby, nblocks, flags)
    switch (n_dog) {
        case 3: EXT(3); break; case 4: EXT(4); break; case 5: EXT(5); break;
        case 6: EXT(6); break; case 7: EXT(7); break; case 8: EXT(8); break;
        case 9: EXT(9); break; case 10: EXT(10); break; case 11: EXT(11); break;
        default: return -1;
    }
#undef EXT
    return 0;
}

void launch_refine(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const unsigned long long* ckey,
                   const float* cval, int* counters, int cap, SaKey* ref, int* stage,
                   unsigned long long* skey, int* sidx, int fcap, int batch)
{
    HIPK(k_refine, dim3(64, batch), dim3(256), 0, st, P, sc, ckey, cval, counters, cap, ref, stage, skey, sidx, fcap);
}

void launch_rank(hipStream_t st, const unsigned long long* skey, const int* sidx, int* counters,
                 int* flist, int fcap, int batch)
{
    HIPK(k_rank, dim3(8, batch), dim3(256), 0, st, skey, sidx, counters, flist, fcap);
}

void launch_orientation(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const SaKey* keys,
                        const int* index, const int* counters, int cnt_slot, int kcap, int icap, int mode,
                        int* ori_n, float* ori_theta, float* ori_hist, int batch)
{
    HIPK(k_orientation, dim3(2048, batch), dim3(64), (size_t)(sc.n_bins * 65 + 64) * sizeof(float), st, P, sc, keys, index, counters, cnt_slot,
         kcap, icap, mode, ori_n, ori_theta, ori_hist);
}

void launch_expand(hipStream_t st, const SaKey* keys, const int* index, int* counters, int cnt_slot,
                   int kcap, int icap, const int* ori_n, const float* ori_theta, SaKey* out,
                   int* out_src, int ocap, int batch)
{
    HIPK(k_expand, dim3(batch), dim3(1024), 0, st, keys, index, counters, cnt_slot, kcap, icap,
         ori_n, ori_theta, out, out_src, ocap);
}

void launch_descriptor(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const SaKey* keys,
                       const int* counters, int cnt_slot, int kcap, SaStd* out_std, float* out_descr,
                       int batch)
{
    const int nd = sc.n_hist * sc.n_hist * sc.n_ori;
    // one wave per keypoint, further keypoints of an image in a grid-stride loop; most images
    // have fewer keypoints than this and empty blocks are not free to dispatch
    const dim3 grid(1024, batch), blk(64);
    switch ((nd + 63) / 64) {
        case 1: HIPK(k_descriptor_wave<1>, grid, blk, 0, st, P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr); break;
        case 2:
            if (sc.n_hist == 4 && sc.n_ori == 8 && !sc.debug)
                HIPK(k_descriptor_wave_448, grid, blk, 0, st, P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr);
            else
                HIPK(k_descriptor_wave<2>, grid, blk, 0, st, P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr);
            break;
        case 3: HIPK(k_descriptor_wave<3>, grid, blk, 0, st, P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr); break;
        default: HIPK(k_descriptor_wave<4>, grid, blk, 0, st, P, sc, keys, counters, cnt_slot, kcap, out_std, out_descr); break;
    }
}

void launch_pack_std(hipStream_t st, const SaStd* std_recs, const int* counters, int cnt_slot, int kcap,
                     SaStd* packed, int batch)
{
    HIPK(k_pack_std, dim3(32, batch), dim3(256), 0, st, std_recs, counters, cnt_slot, kcap, packed);
}

}  // namespace sa
