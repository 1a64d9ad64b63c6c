// Host side of the batched pipeline: context, HBM pool, launch sequence, transfers.
// Implements Part 2 of include/sift_amd.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/sift_amd.h"
#include "sift_ctx.h"
#include "sift_launch.h"

static_assert(sizeof(sift_amd_key) == sizeof(SaKey), "key record layout");
static_assert(sizeof(sift_keypoint_std) == sizeof(SaStd) && sizeof(SaStd) == 144, "std record layout");
static_assert(SIFT_AMD_MAX_OCT == SA_MAX_OCT && SIFT_AMD_MAX_SCA == SA_MAX_SCA, "limits");

static thread_local char g_err[512] = "";

void sa_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* sift_amd_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            sa_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                     \
        }                                                                                  \
    } while (0)

// Makes `dev` the calling thread's current HIP device for the guard's lifetime and puts the
// caller's device back afterwards, so a host thread that drives several GPUs (or shares the
// thread with another HIP user) does not find its current device changed by a library call.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        else prev = -1;                      // nothing to restore
        if (!ok) sa_set_error("hipSetDevice(%d) failed", dev);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define SA_ON_DEVICE(c)                     \
    DeviceGuard dev_guard_((c)->device);    \
    if (!dev_guard_.ok) return -1

extern "C" int sift_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------- parameters --
// Scalar conversions done exactly as the reference does them on the host.

static const double kLn2 = 0.693147180559945309417;

void sa_default_parameters(sift_parameters* p)   // lib_sift_anatomy.c:771-789
{
    p->n_oct = 8; p->n_spo = 3;
    p->sigma_min = 0.8; p->delta_min = 0.5; p->sigma_in = 0.5;
    p->C_DoG = 0.013333333; p->C_edge = 10;
    p->n_bins = 36; p->lambda_ori = 1.5; p->t = 0.80;
    p->n_hist = 4; p->n_ori = 8; p->lambda_descr = 6; p->itermax = 5;
}

int sa_number_of_octaves(int w, int h, const sift_parameters* p)   // lib_sift_anatomy.c:793-802
{
    int hmin = 12;
    int h0 = (int)((float)std::min(w, h) / p->delta_min);
    if (h0 / hmin < 1) return 0;
    int n = (int)(log((double)(h0 / hmin)) / kLn2) + 1;
    return std::min(p->n_oct, n);
}

float sa_convert_threshold(const sift_parameters* p)   // lib_sift_anatomy.c:807-814
{
    float k_nspo = (float)exp(kLn2 / (double)((float)p->n_spo));
    float k_3 = (float)exp(kLn2 / (double)((float)3));
    return (k_nspo - 1) / (k_3 - 1) * p->C_DoG;
}

// lib_discrete.c:145-164,172-174
static int gaussian_taps(float sigma, std::vector<float>& g)
{
    int rad = (int)ceil((double)(4 * sigma));
    if (rad < 0) rad = 0;
    g.assign(rad + 1, 0.f);
    g[0] = 1.f;
    if (sigma > 0) {
        float sum = g[0];
        for (int i = 1; i <= rad; i++) {
            g[i] = (float)exp(-0.5 * (double)(float)i * (double)(float)i / (double)sigma / (double)sigma);
            sum += 2 * g[i];
        }
        for (int i = 0; i <= rad; i++) g[i] /= sum;
    }
    return rad;
}

void sa_layout_lowe(SaLayout* L, int n_oct, int n_spo, int im_w, int im_h, float delta_min,
                    float sigma_min)   // lib_scalespace.c:216-252
{
    L->n_oct = n_oct;
    L->n_sca = n_spo + 3;
    L->delta[0] = delta_min;
    L->h[0] = (int)((float)im_h / delta_min);
    L->w[0] = (int)((float)im_w / delta_min);
    for (int o = 1; o < n_oct; o++) {
        L->w[o] = L->w[o - 1] / 2;
        L->h[o] = L->h[o - 1] / 2;
        L->delta[o] = (float)(L->delta[o - 1] * 2.0);
    }
    for (int o = 0; o < n_oct; o++)
        for (int s = 0; s < n_spo + 3; s++)
            L->sigmas[o][s] = (float)((double)(L->delta[o] / L->delta[0] * sigma_min)
                                      * pow(2.0, (double)((float)s / (float)n_spo)));
}

// ----------------------------------------------------------------- context --

struct BlurSpec {
    float sigma;
    int radius;
    SaTaps taps;
    float* taps_dev;
};

struct ProfEvent { hipEvent_t a, b; int cls, oct, sca; };

struct sift_amd_ctx {
    int device = 0;
    int im_w = 0, im_h = 0, batch = 0, flags = 0;
    sift_parameters p;
    SaLayout L;
    float sigma_in = 0.f;
    SaPyramid pyr;
    SaScalars sc;
    BlurSpec seed;
    BlurSpec step[SA_MAX_OCT][SA_MAX_SCA];
    hipStream_t own_stream = nullptr;
    // Recorded behind every run on the stream the run was queued on.  The next run and every
    // reader wait for it on their own stream (hipStreamWaitEvent: ordering on the device, no host
    // block), so a caller's stream is never touched after the call that received it has returned;
    // the caller may destroy it.
    hipEvent_t run_done = nullptr;
    // second chain of a whole-pipeline run (see run_scalespace_and_detect_forked): its stream and the
    // events between the chains, one per octave and one for the join
    hipStream_t side_stream = nullptr;
    hipEvent_t fork_ev[SA_MAX_OCT + 1] = {};
    int use_fork = -1;              // SIFT_AMD_FORK=0/1 forces one chain / two chains; -1: by batch size
    int last_n = 0;

    // d_seed (oversampled input) and d_tmp (column-pass image) only exist when some blur radius is
    // outside the register-window range and takes the any-radius two-launch form
    float *d_in = nullptr, *d_seed = nullptr, *d_tmp = nullptr;
    // marks of the 8-row x 64-column DoG cells that hold a sample passing the first contrast
    // test, per octave [batch][ceil(h/8)][ceil(w/64)], one allocation (d_cells[0] is its start)
    int* d_cells[SA_MAX_OCT] = {};
    size_t cells_total = 0;             // ints
    bool cells_valid[SA_MAX_OCT] = {};  // set by the scale-space run when every centre-scale blur marked
    unsigned char* d_u8 = nullptr;      // staging of 8-bit host images, grown on demand
    size_t d_u8_bytes = 0;
    float* d_gss[SA_MAX_OCT] = {};
    float* d_dog[SA_MAX_OCT] = {};
    int cap = 0, fcap = 0, kcap = 0;
    unsigned long long* d_ckey = nullptr;
    float* d_cval = nullptr;
    SaKey* d_ref = nullptr;
    int* d_stage = nullptr;
    int* d_counters = nullptr;
    int* d_flist = nullptr;
    unsigned long long* d_skey = nullptr;   // survivors of all filters, unordered: seed key, candidate index
    int* d_sidx = nullptr;
    SaKey* d_given = nullptr;       // keypoints supplied by the caller (description-only entry points)
    int* d_ori_n = nullptr;
    float *d_ori_theta = nullptr, *d_ori_hist = nullptr;
    SaKey* d_keys = nullptr;
    int* d_key_src = nullptr;
    SaStd* d_std = nullptr;
    SaStd* d_packed = nullptr;       // the batch's records without gaps between images (host gather)
    float* d_descr = nullptr;
    int* h_counters = nullptr;      // pinned
    bool counters_valid = false;
    bool overflowed = false;        // a list of the last run ran out of room (sticky until the next run)
    std::vector<void*> allocs;

    // cached hipGraphs of the whole launch sequence, keyed by (input pointer, batch, timers)
    struct GraphEntry {
        const float* d_images; int n; unsigned prof_mask; int what;
        hipGraphExec_t exec; std::vector<ProfEvent> events;
        hipGraph_t graph = nullptr;     // kept for the lifetime of exec
        ~GraphEntry()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
        }
    };
    bool use_graph = true;          // SIFT_AMD_GRAPH=0 launches kernel by kernel
    bool use_cells = true;          // SIFT_AMD_CELLS=0: the extrema scan visits every sample
    bool use_pairs = true;          // SIFT_AMD_PAIR=0: every blur is a launch of its own
    bool use_sparse = true;         // SIFT_AMD_SPARSE=0: the last image of every stack is computed everywhere
    std::vector<GraphEntry*> graphs;
    // keys launched eagerly once: a graph is built when a key comes back, so callers that pass a
    // new buffer every time do not pay a capture per call
    struct SeenKey { const float* d_images; int n; int what; };
    std::vector<SeenKey> seen_once;
    GraphEntry* capturing = nullptr;
    hipStream_t capture_stream = nullptr;

    unsigned prof_mask = 0;
    std::vector<ProfEvent> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[SIFT_AMD_K_COUNT][SA_MAX_OCT + 1] = {};
    long long prof_n[SIFT_AMD_K_COUNT][SA_MAX_OCT + 1] = {};
    // the blur launches once more per (octave, image index of the stack they write)
    double prof_blur_ms[SA_MAX_OCT][SA_MAX_SCA] = {};
    long long prof_blur_n[SA_MAX_OCT][SA_MAX_SCA] = {};
};

// Blocking copy on the context's own (non-blocking) stream.  The library never uses the legacy
// default stream: a plain hipMemcpy from one host thread fails with "operation would make the legacy
// stream depend on a capturing blocking stream" whenever another host thread happens to be capturing a
// hipGraph at that moment, even in thread-local capture mode and on non-blocking streams.
static hipError_t sa_copy(sift_amd_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    if (!bytes) return hipSuccess;
    DeviceGuard dev_guard_(c->device);
    if (!dev_guard_.ok) return hipErrorInvalidDevice;
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->own_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(c->own_stream);
}

// The context's own stream, ordered behind the last run (whatever stream that was queued on).
static hipError_t own_stream_after_run(sift_amd_ctx* c)
{
    return hipStreamWaitEvent(c->own_stream, c->run_done, 0);
}

// Called behind the last launch of a run.  Inside a capture nothing is recorded: the event is
// recorded behind the graph's launch instead.
static hipError_t mark_run_end(sift_amd_ctx* c, hipStream_t st)
{
    if (c->capturing) return hipSuccess;
    return hipEventRecord(c->run_done, st);
}

template <typename T>
static int dev_alloc(sift_amd_ctx* c, T** p, size_t count)
{
    void* v = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&v, bytes);
    if (e != hipSuccess) {
        sa_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return -1;
    }
    c->allocs.push_back(v);
    *p = (T*)v;
    return 0;
}

static int make_blur(sift_amd_ctx* c, BlurSpec* b, float sigma)
{
    std::vector<float> g;
    b->sigma = sigma;
    b->radius = gaussian_taps(sigma, g);
    memset(&b->taps, 0, sizeof b->taps);
    b->taps_dev = nullptr;
    if (b->radius <= SA_MAX_RADIUS_FAST) {
        for (int k = 0; k <= b->radius; k++) b->taps.g[k] = g[k];
    }
    if (b->radius < 1 || b->radius > SA_MAX_RADIUS_FAST) {
        if (b->radius >= SA_MAX_TAPS) {
            sa_set_error("blur radius %d (sigma %g) exceeds the supported maximum %d", b->radius, sigma, SA_MAX_TAPS - 1);
            return -1;
        }
        if (dev_alloc(c, &b->taps_dev, g.size())) return -1;
        HIP_TRY(sa_copy(c, b->taps_dev, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

static long long total_pixels(const SaLayout& L)
{
    long long n = 0;
    for (int o = 0; o < L.n_oct; o++) n += (long long)L.w[o] * L.h[o];
    return n;
}

// cap_req / fcap_req: capacities of the candidate and survivor lists per image (0 = defaults
// derived from the pyramid size).
sift_amd_ctx* sa_create_from_layout(int device, int im_w, int im_h, int max_batch, const sift_parameters* p,
                                    const SaLayout* L, float sigma_in, int flags, int cap_req, int fcap_req)
{
    if (max_batch < 1 || im_w < 1 || im_h < 1) { sa_set_error("invalid size or batch"); return nullptr; }
    if (L->n_oct < 1 || L->n_oct > SA_MAX_OCT) {
        sa_set_error("number of octaves %d outside 1..%d (image %dx%d too small or n_oct too large)",
                     L->n_oct, SA_MAX_OCT, im_w, im_h);
        return nullptr;
    }
    if (L->n_sca < 4 || L->n_sca > SA_MAX_SCA) { sa_set_error("n_spo must be in 1..%d", SA_MAX_SCA - 3); return nullptr; }
    if (p->n_bins < 3 || p->n_bins > SA_MAX_BINS) { sa_set_error("n_bins must be in 3..%d", SA_MAX_BINS); return nullptr; }
    if (p->n_hist < 1 || p->n_hist > 8 || p->n_ori < 2 || p->n_hist * p->n_hist * p->n_ori > SA_MAX_DESCR) {
        sa_set_error("descriptor shape n_hist=%d n_ori=%d unsupported (n_hist<=8, n_ori>=2, n_hist^2*n_ori<=%d)",
                     p->n_hist, p->n_ori, SA_MAX_DESCR);
        return nullptr;
    }
    if (!(p->delta_min <= 1.f) || !(p->delta_min > 0.f)) { sa_set_error("delta_min must be in (0,1]"); return nullptr; }
    if (!(L->sigmas[0][0] >= sigma_in)) { sa_set_error("sigma_min must be >= sigma_in"); return nullptr; }
    // the blur kernels address one image with 32-bit byte offsets
    if ((long long)L->w[0] * L->h[0] * 4 >= (1ll << 31) || (long long)im_w * im_h * 4 >= (1ll << 31)) {
        sa_set_error("image too large: %dx%d (first octave %dx%d) exceeds 2 GiB per scale-space image",
                     im_w, im_h, L->w[0], L->h[0]);
        return nullptr;
    }
    DeviceGuard dev_guard_(device);
    if (!dev_guard_.ok) return nullptr;

    sift_amd_ctx* c = new sift_amd_ctx();
    c->device = device; c->im_w = im_w; c->im_h = im_h; c->batch = max_batch; c->flags = flags;
    c->p = *p; c->L = *L; c->sigma_in = sigma_in;
    auto fail = [&]() -> sift_amd_ctx* { sift_amd_destroy(c); return nullptr; };
    if (const char* m = getenv("SIFT_AMD_GRAPH")) c->use_graph = atoi(m) != 0;
    if (const char* m = getenv("SIFT_AMD_CELLS")) c->use_cells = atoi(m) != 0;
    if (const char* m = getenv("SIFT_AMD_PAIR")) c->use_pairs = atoi(m) != 0;
    if (const char* m = getenv("SIFT_AMD_SPARSE")) c->use_sparse = atoi(m) != 0;
    if (const char* m = getenv("SIFT_AMD_FORK")) c->use_fork = atoi(m) != 0 ? 1 : 0;

    for (int o = 1; o < L->n_oct; o++)
        if (L->w[o] != L->w[o - 1] / 2 || L->h[o] != L->h[o - 1] / 2) {
            sa_set_error("octave %d is not the previous octave halved", o);
            return fail();
        }

    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->run_done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(c->run_done, c->own_stream) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) {
        sa_set_error("hipStreamCreate / hipEventCreate failed");
        return fail();
    }
    for (auto& e : c->fork_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { sa_set_error("hipEventCreate failed"); return fail(); }

    // blur schedule: lib_sift_anatomy.c:132 (seed) and :155-158 (steps)
    {
        float smin = L->sigmas[0][0], dmin = L->delta[0];
        float se = (float)(sqrt((double)(smin * smin - sigma_in * sigma_in)) / (double)dmin);
        if (make_blur(c, &c->seed, se)) return fail();
        for (int o = 0; o < L->n_oct; o++)
            for (int s = 1; s < L->n_sca; s++) {
                float sp = L->sigmas[o][s - 1], sn = L->sigmas[o][s];
                float sx = (float)(sqrt((double)(sn * sn - sp * sp)) / (double)L->delta[o]);
                if (make_blur(c, &c->step[o][s], sx)) return fail();
            }
    }

    // scalars
    float thresh = sa_convert_threshold(p);
    c->sc.thr = thresh;
    c->sc.thr_pre = (float)(0.8 * (double)thresh);                      // lib_sift_anatomy.c:855
    c->sc.thr_edge = (p->C_edge + 1) * (p->C_edge + 1) / p->C_edge;     // lib_sift_anatomy.c:859
    c->sc.sigma_ratio = L->sigmas[0][1] / L->sigmas[0][0];              // lib_sift_anatomy.c:436
    c->sc.lambda_ori = p->lambda_ori; c->sc.t_peak = p->t; c->sc.lambda_descr = p->lambda_descr;
    c->sc.itermax = p->itermax; c->sc.n_bins = p->n_bins; c->sc.n_hist = p->n_hist; c->sc.n_ori = p->n_ori;
    c->sc.debug = getenv("SIFT_AMD_DEBUG") ? atoi(getenv("SIFT_AMD_DEBUG")) : 0;

    // HBM pool
    const long long n0 = (long long)L->w[0] * L->h[0];
    const long long ntot = total_pixels(*L);
    const long long B = max_batch;
    if (dev_alloc(c, &c->d_in, (size_t)(B * im_w * im_h))) return fail();
    {
        auto slow = [](const BlurSpec& b) { return b.radius < 1 || b.radius > SA_MAX_RADIUS_FAST; };
        bool any_slow = slow(c->seed);
        for (int o = 0; o < L->n_oct; o++)
            for (int s = 1; s < L->n_sca; s++) any_slow = any_slow || slow(c->step[o][s]);
        if (any_slow) {
            if (dev_alloc(c, &c->d_seed, (size_t)(B * n0))) return fail();
            if (dev_alloc(c, &c->d_tmp, (size_t)(B * n0))) return fail();
        }
    }
    memset(&c->pyr, 0, sizeof c->pyr);
    c->pyr.n_oct = L->n_oct; c->pyr.n_sca = L->n_sca; c->pyr.im_w = im_w; c->pyr.im_h = im_h;
    for (int o = 0; o < L->n_oct; o++) {
        const long long n = (long long)L->w[o] * L->h[o];
        if (dev_alloc(c, &c->d_gss[o], (size_t)(B * L->n_sca * n))) return fail();
        // DoG images exist in memory only for callers that download them; the pipeline itself
        // forms DoG values from the Gaussian stack where it needs them
        if ((flags & SIFT_AMD_KEEP_SCALESPACE) && dev_alloc(c, &c->d_dog[o], (size_t)(B * (L->n_sca - 1) * n))) return fail();
        SaOctave& oc = c->pyr.oct[o];
        oc.gss = c->d_gss[o]; oc.dog = c->d_dog[o];
        oc.gss_stride = L->n_sca * n; oc.dog_stride = (L->n_sca - 1) * n;
        oc.w = L->w[o]; oc.h = L->h[o]; oc.delta = L->delta[o];
        for (int s = 0; s < L->n_sca; s++) oc.sigmas[s] = L->sigmas[o][s];
    }
    long long cap = (flags & SIFT_AMD_KEEP_ALL_EXTREMA) ? ntot / 8 : ntot / 64;
    cap = std::max<long long>(cap, 16384);
    cap = std::min<long long>(cap, 1 << 24);
    if (cap_req > 0) cap = std::min<long long>(std::max<long long>(cap, cap_req), 1 << 26);
    c->cap = (int)cap;
    c->fcap = (int)std::max<long long>(4096, cap / 4);
    if (fcap_req > 0) c->fcap = std::min(std::max(c->fcap, fcap_req), 1 << 26);
    // SIFT_AMD_CAP=<n>: list capacities forced small, to exercise the overflow report
    if (const char* m = (cap_req > 0 || fcap_req > 0) ? nullptr : getenv("SIFT_AMD_CAP")) {
        c->cap = std::max(64, atoi(m));
        c->fcap = std::max(16, c->cap / 4);
    }
    // room for two orientations per survivor on average; four when the caller sized the lists
    c->kcap = (fcap_req > 0 ? 4 : 2) * c->fcap;
    if (dev_alloc(c, &c->d_ckey, (size_t)(B * c->cap))) return fail();
    if (dev_alloc(c, &c->d_cval, (size_t)(B * c->cap))) return fail();
    if (dev_alloc(c, &c->d_ref, (size_t)(B * c->cap))) return fail();
    if (dev_alloc(c, &c->d_stage, (size_t)(B * c->cap))) return fail();
    if (dev_alloc(c, &c->d_counters, (size_t)(B * SA_CNT_STRIDE))) return fail();
    {
        size_t off[SA_MAX_OCT + 1] = {0};
        for (int o = 0; o < L->n_oct; o++)
            off[o + 1] = off[o] + (size_t)B * ((L->h[o] + 7) / 8) * ((L->w[o] + 63) / 64);
        c->cells_total = off[L->n_oct];
        int* cells = nullptr;
        if (dev_alloc(c, &cells, (c->cells_total + 3) & ~(size_t)3)) return fail();    // cleared in 16-byte words
        for (int o = 0; o < L->n_oct; o++) c->d_cells[o] = cells + off[o];
    }
    if (dev_alloc(c, &c->d_flist, (size_t)(B * c->fcap))) return fail();
    if (dev_alloc(c, &c->d_skey, (size_t)(B * c->fcap))) return fail();
    if (dev_alloc(c, &c->d_sidx, (size_t)(B * c->fcap))) return fail();
    if (dev_alloc(c, &c->d_given, (size_t)(B * c->fcap))) return fail();
    if (dev_alloc(c, &c->d_ori_n, (size_t)(B * c->fcap))) return fail();
    if (dev_alloc(c, &c->d_ori_theta, (size_t)(B * c->fcap * SA_MAX_PEAKS))) return fail();
    if (dev_alloc(c, &c->d_ori_hist, (size_t)(B * c->fcap * SA_MAX_BINS))) return fail();
    if (dev_alloc(c, &c->d_keys, (size_t)(B * c->kcap))) return fail();
    if (dev_alloc(c, &c->d_key_src, (size_t)(B * c->kcap))) return fail();
    if (dev_alloc(c, &c->d_std, (size_t)(B * c->kcap))) return fail();
    if (B > 1 && dev_alloc(c, &c->d_packed, (size_t)(B * c->kcap))) return fail();
    if (dev_alloc(c, &c->d_descr, (size_t)(B * c->kcap * SA_MAX_DESCR))) return fail();
    if (hipHostMalloc((void**)&c->h_counters, (size_t)B * SA_CNT_STRIDE * sizeof(int)) != hipSuccess) {
        sa_set_error("hipHostMalloc failed");
        return fail();
    }
    return c;
}

extern "C" sift_amd_ctx* sift_amd_create(int device, int w, int h, int max_batch,
                                         const sift_parameters* p_in, int flags)
{
    sift_parameters p;
    if (p_in) p = *p_in; else sa_default_parameters(&p);
    if (p.n_spo < 1 || p.n_spo + 3 > SA_MAX_SCA) { sa_set_error("n_spo must be in 1..%d", SA_MAX_SCA - 3); return nullptr; }
    SaLayout L;
    memset(&L, 0, sizeof L);
    int n_oct = sa_number_of_octaves(w, h, &p);
    if (n_oct < 1 || n_oct > SA_MAX_OCT) {
        sa_set_error("image %dx%d gives %d octaves (supported 1..%d)", w, h, n_oct, SA_MAX_OCT);
        return nullptr;
    }
    sa_layout_lowe(&L, n_oct, p.n_spo, w, h, p.delta_min, p.sigma_min);
    return sa_create_from_layout(device, w, h, max_batch, &p, &L, p.sigma_in, flags, 0, 0);
}

extern "C" void sift_amd_destroy(sift_amd_ctx* c)
{
    if (!c) return;
    DeviceGuard dev_guard_(c->device);
    if (c->run_done) (void)hipEventSynchronize(c->run_done);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (auto& e : c->prof_pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : c->prof_pool) hipEventDestroy(e);
    for (auto* g : c->graphs) {
        for (auto& e : g->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        delete g;
    }
    if (c->capture_stream) hipStreamDestroy(c->capture_stream);
    if (c->run_done) hipEventDestroy(c->run_done);
    for (auto& e : c->fork_ev) if (e) hipEventDestroy(e);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    for (void* v : c->allocs) hipFree(v);
    if (c->d_u8) hipFree(c->d_u8);
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int sift_amd_get_geometry(const sift_amd_ctx* c, sift_amd_geometry* g)
{
    memset(g, 0, sizeof *g);
    g->n_oct = c->L.n_oct; g->n_sca = c->L.n_sca;
    for (int o = 0; o < c->L.n_oct; o++) {
        g->w[o] = c->L.w[o]; g->h[o] = c->L.h[o]; g->delta[o] = c->L.delta[o];
        for (int s = 0; s < c->L.n_sca; s++) {
            g->sigmas[o][s] = c->L.sigmas[o][s];
            g->blur_radius[o][s] = s ? c->step[o][s].radius : (o == 0 ? c->seed.radius : 0);
        }
    }
    g->thresh_pre = c->sc.thr_pre; g->thresh = c->sc.thr; g->thresh_edge = c->sc.thr_edge;
    return 0;
}

// --------------------------------------------------------------- profiling --

struct ProfScope {
    sift_amd_ctx* c; hipStream_t st; ProfEvent ev; bool on;
    ProfScope(sift_amd_ctx* c_, hipStream_t st_, int cls, int oct, int sca = -1) : c(c_), st(st_), on(false)
    {
        if (!(c->prof_mask & (1u << cls))) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else hipEventCreate(&e);
            return e;
        };
        ev.a = get(); ev.b = get(); ev.cls = cls; ev.oct = oct; ev.sca = sca;
        hipEventRecord(ev.a, st);
        on = true;
    }
    ~ProfScope()
    {
        if (!on) return;
        hipEventRecord(ev.b, st);
        c->prof_pending.push_back(ev);
    }
};

static void prof_collect(sift_amd_ctx* c)
{

    for (auto& e : c->prof_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.cls][e.oct] += ms; c->prof_n[e.cls][e.oct] += 1;
            if (e.sca >= 0 && e.sca < SA_MAX_SCA) { c->prof_blur_ms[e.oct][e.sca] += ms; c->prof_blur_n[e.oct][e.sca] += 1; }
        }
        c->prof_pool.push_back(e.a); c->prof_pool.push_back(e.b);
    }
    c->prof_pending.clear();
}

extern "C" int sift_amd_profile_enable(sift_amd_ctx* c, unsigned mask) { c->prof_mask = mask; return 0; }

extern "C" int sift_amd_profile_read(sift_amd_ctx* c, int cls, int octave, double* ms, long long* launches, int reset)
{
    if (cls < 0 || cls >= SIFT_AMD_K_COUNT || octave >= SA_MAX_OCT) { sa_set_error("bad class/octave"); return -1; }
    SA_ON_DEVICE(c);
    prof_collect(c);
    double t = 0; long long n = 0;
    for (int o = 0; o < SA_MAX_OCT; o++)
        if (octave < 0 || octave == o) {
            t += c->prof_ms[cls][o]; n += c->prof_n[cls][o];
            if (reset) { c->prof_ms[cls][o] = 0; c->prof_n[cls][o] = 0; }
        }
    if (ms) *ms = t;
    if (launches) *launches = n;
    return 0;
}

extern "C" int sift_amd_profile_read_blur(sift_amd_ctx* c, int octave, int scale, double* ms, long long* launches, int reset)
{
    if (octave < 0 || octave >= SA_MAX_OCT || scale < 0 || scale >= SA_MAX_SCA) { sa_set_error("bad octave/scale"); return -1; }
    SA_ON_DEVICE(c);
    prof_collect(c);
    if (ms) *ms = c->prof_blur_ms[octave][scale];
    if (launches) *launches = c->prof_blur_n[octave][scale];
    if (reset) { c->prof_blur_ms[octave][scale] = 0; c->prof_blur_n[octave][scale] = 0; }
    return 0;
}

// ---------------------------------------------------------------- pipeline --

// sca: index, within the octave's Gaussian stack, of the image this blur writes
static void run_blur(sift_amd_ctx* c, hipStream_t st, int o, int sca, const BlurSpec& b, const float* in, long long s_in,
                     float* out, long long s_out, float* dog, long long s_dog, float* sub, long long s_sub, int n,
                     int ovs_w = 0, int ovs_h = 0, float ovs_delta = 0.f, int* cells = nullptr, const int* need_cells = nullptr)
{
    sa::BlurArgs a;
    a.flags = cells; a.flag_thr = c->sc.thr_pre; a.need_cells = need_cells;
    a.ovs_w = ovs_w; a.ovs_h = ovs_h; a.ovs_delta = ovs_delta;
    a.in = in; a.tmp = c->d_tmp; a.out = out; a.dog = dog; a.sub = sub;
    a.s_in = s_in; a.s_tmp = (long long)c->L.w[0] * c->L.h[0]; a.s_out = s_out; a.s_dog = s_dog; a.s_sub = s_sub;
    a.w = c->L.w[o]; a.h = c->L.h[o]; a.batch = n;
    a.radius = b.radius; a.taps = b.taps; a.taps_dev = b.taps_dev;
    if (b.radius >= 1 && b.radius <= SA_MAX_RADIUS_FAST) {
        // the seed launch that also evaluates the oversampling is timed as the oversampling class
        ProfScope ps(c, st, ovs_w > 0 ? SIFT_AMD_K_OVERSAMPLE : SIFT_AMD_K_BLUR_FUSED, o, sca);
        sa::launch_blur_fused(st, a);
        return;
    }
    { ProfScope ps(c, st, SIFT_AMD_K_BLUR_V, o); sa::launch_blur_pass(st, a, 0); }
    { ProfScope ps(c, st, SIFT_AMD_K_BLUR_H, o); sa::launch_blur_pass(st, a, 1); }
    if (sub) {
        ProfScope ps(c, st, SIFT_AMD_K_SUBSAMPLE, o + 1);
        sa::launch_subsample(st, out, a.w, s_out, sub, a.w / 2, a.h / 2, s_sub, n);
    }
}

// The blur launches that write images s_from .. s_to of octave o's Gaussian stack, in order, on `st`
// (s_from == 0, octave 0: the seed image first).  Pairs of images that have a two-blur launch are
// taken together when both lie in the range.
static void octave_blurs(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st, int o, int s_from, int s_to,
                         bool mark, bool sparse_last)
{
    const SaLayout& L = c->L;
    const int ns = L.n_sca;
    const long long np = (long long)L.w[o] * L.h[o];
    const long long sg = ns * np, sd = (ns - 1) * np;
    float* g = c->d_gss[o];
    float* d = c->d_dog[o];
    // DoG image s-1 is a centre scale of the extrema scan for s-1 in 1..ns-3
    auto marked = [&](int sc) { return mark && sc - 1 >= 1 && sc - 1 <= ns - 3; };
    auto emits_next_octave = [&](int sc) { return sc == ns - 3 && o + 1 < L.n_oct; };
    // Images sa and sa+1 in one launch where an instantiation exists (image sa is then written but
    // not read back); sa = 0: the seed image, whose source is the oversampled input.  Not for
    // contexts that store DoG images.
    auto try_pair = [&](int sa) -> bool {
        if (!c->use_pairs || d || sa + 1 > s_to || emits_next_octave(sa)) return false;
        sa::BlurPairArgs pa;
        const BlurSpec& ba = sa == 0 ? c->seed : c->step[o][sa];
        const BlurSpec& bb = c->step[o][sa + 1];
        if (sa == 0) {
            if (o != 0 || !(L.delta[0] < 1.f)) return false;
            pa.in = d_images; pa.s_in = (long long)c->im_w * c->im_h;
            pa.ovs_w = c->im_w; pa.ovs_h = c->im_h; pa.ovs_delta = L.delta[0];
        } else {
            pa.in = g + (sa - 1) * np; pa.s_in = sg;
        }
        pa.out_a = g + sa * np; pa.out_b = g + (sa + 1) * np;
        pa.s_a = pa.s_b = sg;
        pa.w = L.w[o]; pa.h = L.h[o]; pa.batch = n;
        pa.radius_a = ba.radius; pa.radius_b = bb.radius;
        pa.taps_a = ba.taps; pa.taps_b = bb.taps;
        pa.flag_on_a = marked(sa); pa.flag_on_b = marked(sa + 1);
        pa.flags = (pa.flag_on_a || pa.flag_on_b) ? c->d_cells[o] : nullptr;
        pa.flag_thr = c->sc.thr_pre;
        if (emits_next_octave(sa + 1)) {
            pa.sub = c->d_gss[o + 1];
            pa.s_sub = (long long)ns * L.w[o + 1] * L.h[o + 1];
        }
        if (!sa::launch_blur_pair(st, pa, true)) return false;
        ProfScope ps(c, st, SIFT_AMD_K_BLUR_FUSED, o, sa + 1);     // (timed under the second image of the pair)
        sa::launch_blur_pair(st, pa);
        return true;
    };
    int s_first = std::max(s_from, 1);
    if (o == 0 && s_from == 0) {
        const bool fuse_ovs = c->seed.radius >= 1 && c->seed.radius <= SA_MAX_RADIUS_FAST;
        if (L.delta[0] < 1.f && fuse_ovs && try_pair(0)) {
            s_first = 2;
        } else if (L.delta[0] < 1.f && fuse_ovs) {
            // oversampling evaluated inside the seed blur's loads: no oversampled image in memory
            run_blur(c, st, 0, 0, c->seed, d_images, (long long)c->im_w * c->im_h, g, sg, nullptr, 0, nullptr, 0, n,
                     c->im_w, c->im_h, L.delta[0]);
        } else if (L.delta[0] < 1.f) {
            { ProfScope ps(c, st, SIFT_AMD_K_OVERSAMPLE, 0);
              sa::launch_oversample(st, d_images, c->im_w, c->im_h, (long long)c->im_w * c->im_h,
                                    c->d_seed, L.w[0], L.h[0], np, L.delta[0], n); }
            run_blur(c, st, 0, 0, c->seed, c->d_seed, np, g, sg, nullptr, 0, nullptr, 0, n);
        } else {
            run_blur(c, st, 0, 0, c->seed, d_images, (long long)c->im_w * c->im_h, g, sg, nullptr, 0, nullptr, 0, n);
        }
    }
    // octaves o > 0 start from the even samples of image n_spo of the octave below
    // (lib_sift_anatomy.c:142-149), written by that image's blur
    for (int s = s_first; s <= s_to; s++) {
        if (try_pair(s)) { ++s; continue; }
        float* sub = nullptr;
        long long s_sub = 0;
        if (emits_next_octave(s)) {
            sub = c->d_gss[o + 1];
            s_sub = (long long)ns * L.w[o + 1] * L.h[o + 1];
        }
        int* cells = nullptr;
        if (marked(s)) {
            if (c->step[o][s].radius >= 1 && c->step[o][s].radius <= SA_MAX_RADIUS_FAST) cells = c->d_cells[o];
            else c->cells_valid[o] = false;
        }
        // the marks of all centre scales are complete when the last image's launch starts (same stream)
        const int* need = (sparse_last && s == ns - 1 && c->cells_valid[o] && !sub && !cells) ? c->d_cells[o] : nullptr;
        run_blur(c, st, o, s, c->step[o][s], g + (s - 1) * np, sg, g + s * np, sg, d ? d + (s - 1) * np : nullptr, sd, sub, s_sub, n,
                 0, 0, 0.f, cells, need);
    }
}

// Cell marks let the extrema scan skip regions without any contrast-passing sample.  They are of no
// use when every discrete extremum has to be listed.
static bool uses_marks(const sift_amd_ctx* c) { return !(c->flags & SIFT_AMD_KEEP_ALL_EXTREMA) && c->use_cells; }
// the last image of a stack only where it can be read (see k_blur_fused, SPARSE)
static bool sparse_last_image(const sift_amd_ctx* c, bool whole_stack)
{
    return uses_marks(c) && !whole_stack && c->use_sparse && !(c->flags & SIFT_AMD_KEEP_SCALESPACE);
}

int sa_run_scalespace(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st, bool whole_stack)
{
    SA_ON_DEVICE(c);
    const SaLayout& L = c->L;
    const bool mark = uses_marks(c);
    if (mark) sa::launch_clear_words(st, c->d_cells[0], (long long)c->cells_total);
    for (int o = 0; o < L.n_oct; o++) {
        c->cells_valid[o] = mark;
        octave_blurs(c, d_images, n, st, o, 0, L.n_sca - 1, mark, sparse_last_image(c, whole_stack));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sa_set_error("scale-space launch failed: %s", hipGetErrorString(e)); return -1; }
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

static int scan_octave(sift_amd_ctx* c, int n, hipStream_t st, int o)
{
    const SaLayout& L = c->L;
    const int emit_all = (c->flags & SIFT_AMD_KEEP_ALL_EXTREMA) ? 1 : 0;
    const long long np = (long long)L.w[o] * L.h[o];
    ProfScope ps(c, st, SIFT_AMD_K_EXTREMA, o);
    if (sa::launch_extrema(st, c->d_gss[o], L.n_sca * np, L.w[o], L.h[o], L.n_sca - 1, o,
                           c->sc.thr_pre, emit_all, c->d_ckey, c->d_cval, c->d_counters, c->cap, n,
                           c->cells_valid[o] && !emit_all ? c->d_cells[o] : nullptr)) {
        sa_set_error("unsupported number of DoG images %d", L.n_sca - 1);
        return -1;
    }
    return 0;
}

static int refine_and_rank(sift_amd_ctx* c, int n, hipStream_t st)
{
    { ProfScope ps(c, st, SIFT_AMD_K_REFINE, 0);
      sa::launch_refine(st, c->pyr, c->sc, c->d_ckey, c->d_cval, c->d_counters, c->cap, c->d_ref, c->d_stage,
                        c->d_skey, c->d_sidx, c->fcap, n); }
    { ProfScope ps(c, st, SIFT_AMD_K_RANK, 0);
      sa::launch_rank(st, c->d_skey, c->d_sidx, c->d_counters, c->d_flist, c->fcap, n); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sa_set_error("detection launch failed: %s", hipGetErrorString(e)); return -1; }
    return 0;
}

// Scale-space and detection of a whole-pipeline run as two interleaved chains.  Octave o+1 needs only
// the image of octave o that is subsampled (index n_spo), not its last images nor its extrema scan:
// the octaves alternate between the run's stream and a second stream of the context, octave o+1
// starting as soon as octave o has written that image, so the short launches of the small octaves (a
// few blocks each) run beside the remainder of the octave before instead of after it.  Candidates of
// different octaves are appended to the per-image lists concurrently; the lists are unordered anyway
// (k_rank orders the survivors).  Captured into a hipGraph the two chains become parallel branches.
static int run_scalespace_and_detect_forked(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st)
{
    SA_ON_DEVICE(c);
    const SaLayout& L = c->L;
    const int ns = L.n_sca;
    const bool mark = uses_marks(c), sparse = sparse_last_image(c, false);
    if (mark) sa::launch_clear_words(st, c->d_cells[0], (long long)c->cells_total);
    sa::launch_clear_words(st, c->d_counters, (long long)n * SA_CNT_STRIDE);
    hipStream_t lane[2] = {st, c->side_stream};
    for (int o = 0; o < L.n_oct; o++) {
        hipStream_t mine = lane[o & 1], other = lane[(o + 1) & 1];
        c->cells_valid[o] = mark;
        const bool last = o + 1 == L.n_oct;
        // up to the image the next octave starts from; the first use of the second stream waits for
        // what the run's stream has queued so far (octave 0's event)
        octave_blurs(c, d_images, n, mine, o, 0, last ? ns - 1 : ns - 3, mark, sparse);
        if (!last) {
            HIP_TRY(hipEventRecord(c->fork_ev[o], mine));
            HIP_TRY(hipStreamWaitEvent(other, c->fork_ev[o], 0));
            octave_blurs(c, d_images, n, mine, o, ns - 2, ns - 1, mark, sparse);
        }
        if (scan_octave(c, n, mine, o)) return -1;
    }
    if (L.n_oct > 1) {      // the second stream joins the run's stream
        HIP_TRY(hipEventRecord(c->fork_ev[SA_MAX_OCT], c->side_stream));
        HIP_TRY(hipStreamWaitEvent(st, c->fork_ev[SA_MAX_OCT], 0));
    }
    if (refine_and_rank(c, n, st)) return -1;
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

int sa_run_detect(sift_amd_ctx* c, int n, hipStream_t st)
{
    SA_ON_DEVICE(c);
    sa::launch_clear_words(st, c->d_counters, (long long)n * SA_CNT_STRIDE);
    for (int o = 0; o < c->L.n_oct; o++)
        if (scan_octave(c, n, st, o)) return -1;
    if (refine_and_rank(c, n, st)) return -1;
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

// keys/index/cnt_slot select the keypoints to orient: detection output (d_ref through
// d_flist) or caller-supplied keys (d_given).
int sa_run_orient(sift_amd_ctx* c, int n, hipStream_t st, bool given, int mode)
{
    SA_ON_DEVICE(c);
    const SaKey* keys = given ? c->d_given : c->d_ref;
    const int* index = given ? nullptr : c->d_flist;
    const int kcap = given ? c->fcap : c->cap;
    { ProfScope ps(c, st, SIFT_AMD_K_ORIENTATION, 0);
      sa::launch_orientation(st, c->pyr, c->sc, keys, index, c->d_counters, SA_CNT_NF, kcap, c->fcap, mode,
                             c->d_ori_n, c->d_ori_theta, c->d_ori_hist, n); }
    { ProfScope ps(c, st, SIFT_AMD_K_EXPAND, 0);
      sa::launch_expand(st, keys, index, c->d_counters, SA_CNT_NF, kcap, c->fcap, c->d_ori_n, c->d_ori_theta,
                        c->d_keys, c->d_key_src, c->kcap, n); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sa_set_error("orientation launch failed: %s", hipGetErrorString(e)); return -1; }
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

int sa_run_describe(sift_amd_ctx* c, int n, hipStream_t st)
{
    SA_ON_DEVICE(c);
    ProfScope ps(c, st, SIFT_AMD_K_DESCRIPTOR, 0);
    sa::launch_descriptor(st, c->pyr, c->sc, c->d_keys, c->d_counters, SA_CNT_KEYS, c->kcap, c->d_std, c->d_descr, n);
    if (n > 1) sa::launch_pack_std(st, c->d_std, c->d_counters, SA_CNT_KEYS, c->kcap, c->d_packed, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sa_set_error("descriptor launch failed: %s", hipGetErrorString(e)); return -1; }
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

static hipStream_t pick_stream(sift_amd_ctx* c, void* stream) { return stream ? (hipStream_t)stream : c->own_stream; }

static int begin_run(sift_amd_ctx* c, int n, hipStream_t st)
{
    if (n < 1 || n > c->batch) { sa_set_error("batch %d outside 1..%d", n, c->batch); return -1; }
    // runs of one context share its pool: a run waits, on the device, for the previous one,
    // whatever stream that was queued on
    HIP_TRY(hipStreamWaitEvent(st, c->run_done, 0));
    c->last_n = n; c->counters_valid = false; c->overflowed = false;
    return 0;
}

// what: 0 = scale-space only, 1 = whole pipeline
static int run_sequence(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st, int what)
{
    // (timed runs stay one chain: the timers bracket launches on one stream)
    // Two chains shorten a run of a few images, whose launches leave most of the GPU idle (one 1080p
    // image: 1.00 -> 0.88 ms from the host call to the records, a batch of 8: 1.51 -> 1.34 ms); large
    // batches fill the GPU with every launch and callers that keep several contexts in flight overlap
    // their batches anyway (128 images, four contexts: no gain to -1 %), so they stay one chain.
    const long long px0 = (long long)n * c->L.w[0] * c->L.h[0];
    const bool fork = c->use_fork >= 0 ? c->use_fork != 0 : px0 <= 8ll * 3840 * 2160;
    // (the any-radius blur form keeps its column-pass image in one buffer of the context: one chain)
    if (what == 1 && fork && !c->prof_mask && !c->d_tmp) {
        if (run_scalespace_and_detect_forked(c, d_images, n, st)) return -1;
    } else {
        if (sa_run_scalespace(c, d_images, n, st, what == 0)) return -1;
        if (what == 0) return 0;
        if (sa_run_detect(c, n, st)) return -1;
    }
    if (sa_run_orient(c, n, st, false, 0)) return -1;
    return sa_run_describe(c, n, st);
}

// The launch sequence of a batch is ~70 short dependent launches; replaying it as a hipGraph
// removes the host launch cost between them.  Graphs are cached per (input pointer, batch size,
// timer mask) from the second use of a key on; anything that cannot be captured falls back to
// plain launches.
static int run_graphed(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st, int what)
{
    // event timers cannot be read back from captured event nodes: timed runs launch eagerly
    if (!c->use_graph || c->prof_mask) return run_sequence(c, d_images, n, st, what);
    sift_amd_ctx::GraphEntry* g = nullptr;
    for (auto* e : c->graphs)
        if (e->d_images == d_images && e->n == n && e->prof_mask == c->prof_mask && e->what == what) { g = e; break; }
    if (!g) {
        bool seen = false;
        for (auto& k : c->seen_once) seen = seen || (k.d_images == d_images && k.n == n && k.what == what);
        if (!seen) {
            if (c->seen_once.size() >= 16) c->seen_once.erase(c->seen_once.begin());
            c->seen_once.push_back({d_images, n, what});
            return run_sequence(c, d_images, n, st, what);
        }
        if (c->graphs.size() >= 8) {   // drop the oldest entry
            auto* old = c->graphs.front();
            c->graphs.erase(c->graphs.begin());
            for (auto& e : old->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
            delete old;
        }
        g = new sift_amd_ctx::GraphEntry{d_images, n, c->prof_mask, what, nullptr, {}};
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(c->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            c->capturing = g;
            const int rc = run_sequence(c, d_images, n, c->capture_stream, what);
            c->capturing = nullptr;
            ok = hipStreamEndCapture(c->capture_stream, &graph) == hipSuccess && rc == 0 && graph;
        }
        if (ok) ok = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        g->graph = graph;
        if (!ok) {
            (void)hipGetLastError();
            for (auto& e : g->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
            delete g;
            c->use_graph = false;
            return run_sequence(c, d_images, n, st, what);
        }
        c->graphs.push_back(g);
    }
    HIP_TRY(hipGraphLaunch(g->exec, st));
    HIP_TRY(mark_run_end(c, st));
    return 0;
}

extern "C" int sift_amd_scalespace_device(sift_amd_ctx* c, const float* d_images, int n, void* stream)
{
    SA_ON_DEVICE(c);
    hipStream_t st = pick_stream(c, stream);
    if (begin_run(c, n, st)) return -1;
    return run_graphed(c, d_images, n, st, 0);
}

extern "C" int sift_amd_detect_device(sift_amd_ctx* c, const float* d_images, int n, void* stream)
{
    SA_ON_DEVICE(c);
    hipStream_t st = pick_stream(c, stream);
    if (begin_run(c, n, st)) return -1;
    return run_graphed(c, d_images, n, st, 1);
}

extern "C" int sift_amd_detect_host(sift_amd_ctx* c, const float* images, int n)
{
    if (n < 1 || n > c->batch) { sa_set_error("batch %d outside 1..%d", n, c->batch); return -1; }
    SA_ON_DEVICE(c);
    HIP_TRY(own_stream_after_run(c));          // the previous run may still be reading the input buffer
    HIP_TRY(hipMemcpyAsync(c->d_in, images, (size_t)n * c->im_w * c->im_h * sizeof(float),
                           hipMemcpyHostToDevice, c->own_stream));
    return sift_amd_detect_device(c, c->d_in, n, nullptr);
}

// 8-bit input: convert into the context's float input buffer on the run's stream, then run.
// The device pointer must be 4-byte aligned (hipMalloc'd buffers are).
extern "C" int sift_amd_detect_device_u8(sift_amd_ctx* c, const unsigned char* d_images, int n, int channels,
                                         void* stream)
{
    if (channels != 1 && channels != 3 && channels != 4) { sa_set_error("channels must be 1, 3 or 4 (got %d)", channels); return -1; }
    if (((uintptr_t)d_images & 3) != 0) { sa_set_error("8-bit device images must be 4-byte aligned"); return -1; }
    SA_ON_DEVICE(c);
    hipStream_t st = pick_stream(c, stream);
    if (begin_run(c, n, st)) return -1;
    sa::launch_ingest_u8(st, d_images, channels, c->d_in, (long long)n * c->im_w * c->im_h);
    HIP_TRY(hipGetLastError());
    return run_graphed(c, c->d_in, n, st, 1);
}

extern "C" int sift_amd_detect_host_u8(sift_amd_ctx* c, const unsigned char* images, int n, int channels)
{
    if (n < 1 || n > c->batch) { sa_set_error("batch %d outside 1..%d", n, c->batch); return -1; }
    if (channels != 1 && channels != 3 && channels != 4) { sa_set_error("channels must be 1, 3 or 4 (got %d)", channels); return -1; }
    SA_ON_DEVICE(c);
    const size_t bytes = (size_t)n * c->im_w * c->im_h * channels;
    HIP_TRY(own_stream_after_run(c));          // the previous run may still be reading the staging buffer
    if (bytes > c->d_u8_bytes) {
        HIP_TRY(hipStreamSynchronize(c->own_stream));
        if (c->d_u8) HIP_TRY(hipFree(c->d_u8));
        c->d_u8 = nullptr; c->d_u8_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->d_u8, bytes));
        c->d_u8_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->d_u8, images, bytes, hipMemcpyHostToDevice, c->own_stream));
    return sift_amd_detect_device_u8(c, c->d_u8, n, channels, nullptr);
}

extern "C" int sift_amd_copy_ingested(sift_amd_ctx* c, int n, float* dst)
{
    if (n < 1 || n > c->batch) { sa_set_error("batch %d outside 1..%d", n, c->batch); return -1; }
    SA_ON_DEVICE(c);
    HIP_TRY(own_stream_after_run(c));
    HIP_TRY(sa_copy(c, dst, c->d_in, (size_t)n * c->im_w * c->im_h * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int sa_ctx_image_keys(sift_amd_ctx* c, int img, const void** d_std, const float** d_descr, int* n, int* n_descr)
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    *n = c->h_counters[img * SA_CNT_STRIDE + SA_CNT_KEYS];
    *d_std = c->d_std + (size_t)img * c->kcap;
    *d_descr = c->d_descr + (size_t)img * c->kcap * SA_MAX_DESCR;
    *n_descr = c->p.n_hist * c->p.n_hist * c->p.n_ori;
    return 0;
}

int sa_sync_counters(sift_amd_ctx* c)
{
    SA_ON_DEVICE(c);
    if (!c->counters_valid) {
        hipStream_t st = c->own_stream;
        HIP_TRY(own_stream_after_run(c));
        HIP_TRY(hipMemcpyAsync(c->h_counters, c->d_counters, (size_t)c->batch * SA_CNT_STRIDE * sizeof(int),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        c->counters_valid = true;
        c->overflowed = false;
        for (int b = 0; b < c->last_n; b++)
            if (c->h_counters[b * SA_CNT_STRIDE + SA_CNT_OVERFLOW]) c->overflowed = true;
    }
    // every reader of the run's results gets the error, not only the first one: the lists are
    // truncated until the next run
    if (c->overflowed) {
        for (int b = 0; b < c->last_n; b++)
            if (c->h_counters[b * SA_CNT_STRIDE + SA_CNT_OVERFLOW]) {
                const int* q = c->h_counters + b * SA_CNT_STRIDE;
                sa_set_error("candidate/keypoint list overflow on image %d: %d candidates, %d survivors, %d keypoints "
                             "(capacity %d/%d/%d)", b, q[SA_CNT_CAND], q[SA_CNT_SURV], q[SA_CNT_KEYS], c->cap, c->fcap, c->kcap);
                break;
            }
        return -2;
    }
    return 0;
}

extern "C" int sift_amd_keypoint_counts(sift_amd_ctx* c, int* counts)
{
    int r = sa_sync_counters(c);
    if (r) return r;
    for (int b = 0; b < c->last_n; b++) counts[b] = c->h_counters[b * SA_CNT_STRIDE + SA_CNT_KEYS];
    return 0;
}

extern "C" int sift_amd_fetch_keypoints(sift_amd_ctx* c, int img, sift_keypoint_std* out_std,
                                        sift_amd_key* out_keys, int max_keys)
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    int n = std::min(c->h_counters[img * SA_CNT_STRIDE + SA_CNT_KEYS], max_keys);
    if (n > 0 && out_std)
        HIP_TRY(sa_copy(c, out_std, c->d_std + (size_t)img * c->kcap, (size_t)n * sizeof(SaStd), hipMemcpyDeviceToHost));
    if (n > 0 && out_keys)
        HIP_TRY(sa_copy(c, out_keys, c->d_keys + (size_t)img * c->kcap, (size_t)n * sizeof(SaKey), hipMemcpyDeviceToHost));
    return n;
}

extern "C" int sift_amd_gather_keypoints(sift_amd_ctx* c, sift_keypoint_std** out, int* counts)
{
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    size_t total = 0;
    for (int b = 0; b < c->last_n; b++) {
        counts[b] = c->h_counters[b * SA_CNT_STRIDE + SA_CNT_KEYS];
        total += counts[b];
    }
    sift_keypoint_std* buf = (sift_keypoint_std*)malloc(std::max<size_t>(total, 1) * sizeof(sift_keypoint_std));
    if (!buf) { sa_set_error("out of host memory"); return -1; }
    SA_ON_DEVICE(c);
    hipStream_t st = c->own_stream;            // already ordered behind the run by sa_sync_counters
    // one copy: the records were packed image after image on the device (k_pack_std)
    const SaStd* src = c->last_n > 1 ? c->d_packed : c->d_std;
    if (total) HIP_TRY(hipMemcpyAsync(buf, src, total * sizeof(SaStd), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = buf;
    return 0;
}

extern "C" int sift_amd_gather_keypoints_into(sift_amd_ctx* c, sift_keypoint_std* out, int max_records, int* counts)
{
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    SA_ON_DEVICE(c);
    size_t total = 0;
    for (int b = 0; b < c->last_n; b++) {
        counts[b] = c->h_counters[b * SA_CNT_STRIDE + SA_CNT_KEYS];
        total += counts[b];
    }
    if (total > (size_t)max_records) {
        sa_set_error("gather buffer holds %d records, the batch has %zu", max_records, total);
        return -1;
    }
    hipStream_t st = c->own_stream;            // already ordered behind the run by sa_sync_counters
    const SaStd* src = c->last_n > 1 ? c->d_packed : c->d_std;
    if (total) HIP_TRY(hipMemcpyAsync(out, src, total * sizeof(SaStd), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return (int)total;
}

// ---------------------------------------------------------- copy ceiling --
// What a read-one-write-one stream reaches on this GPU: every block copies one contiguous
// range with eight 16-byte loads in flight per thread, nontemporal accesses (the fastest of
// the forms tried: grid-stride or block-contiguous, 1..8 loads in flight, plain or nontemporal).
typedef float sa_v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_ceiling(const sa_v4f* __restrict__ in, sa_v4f* __restrict__ out, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + 7 * 256 < b1; i += 8 * 256) {
        sa_v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&in[i + u * 256]);
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(v[u], &out[i + u * 256]);
    }
    for (; i < b1; i += 256) out[i] = in[i];
}

// The rate at which this GPU executes non-fused float32 multiplies and adds when nothing else
// limits them: every thread runs 8 independent chains of packed multiply + packed add
// (v_pk_mul_f32 / v_pk_add_f32, operands in vector registers), the form the blur's arithmetic
// takes (each product and each sum rounded separately, as the reference prescribes) and the
// fastest one on gfx950: a packed instruction does two lane-operations in ~1.7x the issue time
// of a plain one, and a plain one with a scalar-register operand takes as long as a packed one.
typedef float sa_v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_valu_ceiling(float* out, int iters)
{
    sa_v2f acc[8];
    // multiplier and addend live in vector registers (values depend on the lane)
    const sa_v2f a = {0.999f + 1e-6f * (threadIdx.x & 3), 0.998f + 1e-6f * (threadIdx.x & 1)};
    const sa_v2f b = {0.001f + 1e-6f * (threadIdx.x & 7), 0.002f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = sa_v2f{(float)(threadIdx.x + k), (float)k};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] * a + b;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
    if (s == 12345.678f) out[0] = s;
}

// Both probes run on a non-blocking stream of their own (never the legacy default stream, see sa_copy).
struct ProbeStream {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = false;
    ProbeStream()
    {
        ok = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
             hipEventCreate(&e1) == hipSuccess;
    }
    ~ProbeStream()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
    }
};

extern "C" int sift_amd_valu_ceiling(int device, int reps, double* lane_gops)
{
    DeviceGuard dev_guard_(device);
    if (!dev_guard_.ok) return -1;
    float* d = nullptr;
    ProbeStream ps;
    int rc = -1;
    float ms = 0.f;
    const int iters = 4096, blocks = 256 * 16;     // 16 waves per SIMD in total, 4 resident at a time
    do {
        if (!ps.ok || hipMalloc((void**)&d, 256) != hipSuccess) break;
        hipLaunchKernelGGL(k_valu_ceiling, dim3(blocks), dim3(256), 0, ps.s, d, iters);
        if (hipStreamSynchronize(ps.s) != hipSuccess) break;
        (void)hipEventRecord(ps.e0, ps.s);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_valu_ceiling, dim3(blocks), dim3(256), 0, ps.s, d, iters);
        (void)hipEventRecord(ps.e1, ps.s);
        if (hipEventSynchronize(ps.e1) != hipSuccess || hipEventElapsedTime(&ms, ps.e0, ps.e1) != hipSuccess) break;
        *lane_gops = (double)blocks * 256 * iters * 32.0 * reps / (ms * 1e-3) / 1e9;
        rc = 0;
    } while (0);
    if (rc) sa_set_error("valu ceiling: %s", hipGetErrorString(hipGetLastError()));
    if (d) (void)hipFree(d);
    return rc;
}

extern "C" int sift_amd_copy_ceiling(int device, size_t bytes, int reps, double* gb_per_s)
{
    DeviceGuard dev_guard_(device);
    if (!dev_guard_.ok) return -1;
    if (bytes < (1u << 20) || reps < 1) { sa_set_error("copy ceiling: need >= 1 MiB and >= 1 repetition"); return -1; }
    const size_t n = bytes / 16;
    void *a = nullptr, *b = nullptr;
    ProbeStream ps;
    int rc = -1;
    float ms = 0.f;
    g_err[0] = 0;
    do {
        if (!ps.ok) break;
        if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) { sa_set_error("copy ceiling: hipMalloc failed"); break; }
        if (hipMemsetAsync(a, 1, n * 16, ps.s) != hipSuccess || hipMemsetAsync(b, 0, n * 16, ps.s) != hipSuccess) break;
        hipLaunchKernelGGL(k_copy_ceiling, dim3(16384), dim3(256), 0, ps.s, (const sa_v4f*)a, (sa_v4f*)b, n);
        if (hipStreamSynchronize(ps.s) != hipSuccess) break;
        (void)hipEventRecord(ps.e0, ps.s);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_copy_ceiling, dim3(16384), dim3(256), 0, ps.s, (const sa_v4f*)a, (sa_v4f*)b, n);
        (void)hipEventRecord(ps.e1, ps.s);
        if (hipEventSynchronize(ps.e1) != hipSuccess || hipEventElapsedTime(&ms, ps.e0, ps.e1) != hipSuccess) break;
        *gb_per_s = 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9;
        rc = 0;
    } while (0);
    if (rc && !g_err[0]) sa_set_error("copy ceiling: %s", hipGetErrorString(hipGetLastError()));
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

extern "C" int sift_amd_launch_stats(long long* out, int reset)
{
    sa::launch_stats(out, reset);
    return 0;
}

extern "C" int sift_amd_stage_counts(sift_amd_ctx* c, int img, int stage_sizes[6], int ext_oct[SIFT_AMD_MAX_OCT])
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    int r = sa_sync_counters(c);
    if (r) return r;
    const int* cnt = c->h_counters + img * SA_CNT_STRIDE;
    for (int k = 0; k < 6; k++) stage_sizes[k] = cnt[SA_CNT_STAGE + k];
    for (int o = 0; o < SA_MAX_OCT; o++) ext_oct[o] = cnt[SA_CNT_EXT_OCT + o];
    // where the scan skipped cells without a contrast-passing sample, extrema below the
    // contrast threshold were not counted
    for (int o = 0; o < c->L.n_oct; o++)
        if (c->cells_valid[o]) { stage_sizes[0] = -1; ext_oct[o] = -1; }
    return 0;
}

extern "C" int sift_amd_fetch_candidates(sift_amd_ctx* c, int img, sift_amd_key* seed, sift_amd_key* refined,
                                         int* stage, int max_n)
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    const int n = std::min(c->h_counters[img * SA_CNT_STRIDE + SA_CNT_CAND], c->cap);
    if (!seed && !refined && !stage) return n;
    std::vector<unsigned long long> key(n);
    std::vector<float> val(n);
    std::vector<SaKey> ref(n);
    std::vector<int> stg(n);
    if (n) {
        HIP_TRY(sa_copy(c, key.data(), c->d_ckey + (size_t)img * c->cap, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(sa_copy(c, val.data(), c->d_cval + (size_t)img * c->cap, n * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(sa_copy(c, ref.data(), c->d_ref + (size_t)img * c->cap, n * sizeof(SaKey), hipMemcpyDeviceToHost));
        HIP_TRY(sa_copy(c, stg.data(), c->d_stage + (size_t)img * c->cap, n * sizeof(int), hipMemcpyDeviceToHost));
    }
    // canonical order = ascending packed (o,s,i,j) of the seed; keys are unique
    std::vector<int> order(n);
    for (int k = 0; k < n; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    const int m = std::min(n, max_n);
    for (int r = 0; r < m; r++) {
        const int k = order[r];
        if (seed) {
            const int o = (int)(key[k] >> 44), s = (int)((key[k] >> 40) & 0xf);
            const int i = (int)((key[k] >> 20) & 0xfffff), j = (int)(key[k] & 0xfffff);
            sift_amd_key& q = seed[r];
            q.o = o; q.s = s; q.i = i; q.j = j;
            q.x = c->L.delta[o] * i; q.y = c->L.delta[o] * j;     // lib_sift_anatomy.c:303-304
            q.sigma = c->L.sigmas[o][s]; q.theta = 0.f; q.val = val[k]; q.edge = 0.f;
        }
        if (refined) memcpy(&refined[r], &ref[k], sizeof(SaKey));
        if (stage) stage[r] = stg[k];
    }
    return m;
}

extern "C" int sift_amd_copy_scalespace(sift_amd_ctx* c, int img, int which, int o, float* dst)
{
    if (img < 0 || img >= c->batch || o < 0 || o >= c->L.n_oct || which < 0 || which > 1) {
        sa_set_error("bad image/octave/which");
        return -1;
    }
    if (which == 1 && !(c->flags & SIFT_AMD_KEEP_SCALESPACE)) {
        sa_set_error("DoG images are only stored in a context created with SIFT_AMD_KEEP_SCALESPACE");
        return -1;
    }
    SA_ON_DEVICE(c);
    HIP_TRY(own_stream_after_run(c));
    const long long np = (long long)c->L.w[o] * c->L.h[o];
    const int nimg = which == 0 ? c->L.n_sca : c->L.n_sca - 1;
    const float* src = (which == 0 ? c->d_gss[o] : c->d_dog[o]) + (long long)img * nimg * np;
    HIP_TRY(sa_copy(c, dst, src, (size_t)(nimg * np) * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sift_amd_fetch_extras(sift_amd_ctx* c, int img, float* orihist, float* descr, int* src_index,
                                     int max_keys)
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    const int n = std::min(c->h_counters[img * SA_CNT_STRIDE + SA_CNT_KEYS], max_keys);
    const int nb = c->p.n_bins, nd = c->p.n_hist * c->p.n_hist * c->p.n_ori;
    if (n && descr) {
        std::vector<float> tmp((size_t)n * SA_MAX_DESCR);
        HIP_TRY(sa_copy(c, tmp.data(), c->d_descr + (size_t)img * c->kcap * SA_MAX_DESCR, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) memcpy(descr + (size_t)k * nd, tmp.data() + (size_t)k * SA_MAX_DESCR, nd * sizeof(float));
    }
    if (n && (orihist || src_index)) {
        const int nf = std::min(c->h_counters[img * SA_CNT_STRIDE + SA_CNT_NF], c->fcap);
        std::vector<float> hist((size_t)std::max(nf, 1) * SA_MAX_BINS);
        std::vector<int> src(n);
        HIP_TRY(sa_copy(c, hist.data(), c->d_ori_hist + (size_t)img * c->fcap * SA_MAX_BINS, (size_t)nf * SA_MAX_BINS * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(sa_copy(c, src.data(), c->d_key_src + (size_t)img * c->kcap, n * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) {
            if (orihist) memcpy(orihist + (size_t)k * nb, hist.data() + (size_t)src[k] * SA_MAX_BINS, nb * sizeof(float));
            if (src_index) src_index[k] = src[k];
        }
    }
    return n;
}

extern "C" int sift_amd_fetch_stage_f_histograms(sift_amd_ctx* c, int img, float* orihist, int max_n)
{
    if (img < 0 || img >= c->last_n) { sa_set_error("image index %d out of range", img); return -1; }
    if (const int rc_ = sa_sync_counters(c)) return rc_;     // -2: a list of the run overflowed
    const int nf = std::min(std::min(c->h_counters[img * SA_CNT_STRIDE + SA_CNT_NF], c->fcap), max_n);
    const int nb = c->p.n_bins;
    if (nf && orihist) {
        std::vector<float> hist((size_t)nf * SA_MAX_BINS);
        HIP_TRY(sa_copy(c, hist.data(), c->d_ori_hist + (size_t)img * c->fcap * SA_MAX_BINS, hist.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int k = 0; k < nf; k++) memcpy(orihist + (size_t)k * nb, hist.data() + (size_t)k * SA_MAX_BINS, nb * sizeof(float));
    }
    return nf;
}

// ---------------------------------------- entry points for caller-given keys --

// Uploads `n` keypoints of image 0 and runs orientation (mode 1: one orientation each) and/or
// description on them.  The scale-space of the last run must be resident.
int sa_describe_given(sift_amd_ctx* c, const SaKey* keys, int n, bool find_orientation)
{
    if (n > c->fcap) { sa_set_error("too many keypoints (%d > %d)", n, c->fcap); return -1; }
    SA_ON_DEVICE(c);
    hipStream_t st = c->own_stream;
    c->last_n = 1; c->counters_valid = false; c->overflowed = false;
    int cnt[SA_CNT_STRIDE] = {0};
    cnt[SA_CNT_NF] = n; cnt[SA_CNT_KEYS] = n;
    HIP_TRY(own_stream_after_run(c));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(sa_copy(c, c->d_counters, cnt, sizeof cnt, hipMemcpyHostToDevice));
    if (find_orientation) {
        if (n) HIP_TRY(sa_copy(c, c->d_given, keys, (size_t)n * sizeof(SaKey), hipMemcpyHostToDevice));
        if (sa_run_orient(c, 1, st, true, 1)) return -1;
    } else {
        if (n) HIP_TRY(sa_copy(c, c->d_keys, keys, (size_t)n * sizeof(SaKey), hipMemcpyHostToDevice));
    }
    if (sa_run_describe(c, 1, st)) return -1;
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// largest number of candidates the extrema scan met on one image of the last run (it counts
// also those that found no room in the list)
int sa_ctx_candidates_seen(const sift_amd_ctx* c)
{
    int m = 0;
    for (int b = 0; b < c->last_n; b++) m = std::max(m, c->h_counters[b * SA_CNT_STRIDE + SA_CNT_CAND]);
    return m;
}

int sa_ctx_limits(const sift_amd_ctx* c, int* cap, int* fcap, int* kcap)
{
    *cap = c->cap; *fcap = c->fcap; *kcap = c->kcap;
    return 0;
}
const SaLayout* sa_ctx_layout(const sift_amd_ctx* c) { return &c->L; }
const sift_parameters* sa_ctx_params(const sift_amd_ctx* c) { return &c->p; }
float* sa_ctx_input_buffer(sift_amd_ctx* c) { return c->d_in; }
hipStream_t sa_ctx_stream(sift_amd_ctx* c) { return c->own_stream; }
const float* sa_ctx_stack(sift_amd_ctx* c, int which, int o)
{
    if (o < 0 || o >= c->L.n_oct || which < 0 || which > 1 || (which == 1 && !c->d_dog[o])) return nullptr;
    DeviceGuard dev_guard_(c->device);
    if (!dev_guard_.ok || hipEventSynchronize(c->run_done) != hipSuccess) return nullptr;
    return which == 0 ? c->d_gss[o] : c->d_dog[o];
}
