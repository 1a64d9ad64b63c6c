// Part 1 of include/sift_amd.h: the reference-named entry points, implemented on top of
// the batched context.  This layer only moves data: it runs the HIP pipeline and then
// materialises the malloc-based structs that existing host code walks and frees itself.
// Errors follow the reference's convention (message on stderr, abort/exit), because the
// reference signatures have no error channel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/sift_amd.h"
#include "sift_ctx.h"
#include "sift_launch.h"

[[noreturn]] static void die(const char* what)
{
    fprintf(stderr, "sift_amd: %s: %s\n", what, sift_amd_last_error());
    abort();
}

static int env_device()
{
    const char* s = getenv("SIFT_AMD_DEVICE");
    return s ? atoi(s) : 0;
}

// ------------------------------------------------------ per-thread context --
// The reference API is stateless; allocating the HBM pool on every call would dominate
// small calls, so each host thread keeps the contexts of its last few (size, parameters,
// mode) combinations: a host that alternates sift_anatomy() with the flat API, or
// scalespace_compute() with anything, does not rebuild a pool per call.
struct CachedCtx {
    sift_amd_ctx* ctx = nullptr;
    int w = 0, h = 0, flags = 0, device = -1;
    sift_parameters p;
    bool lowe = true;          // geometry from the parameters (else: the caller's scale-space)
    SaLayout L;
    float sigma_in = 0;
    int cap = 0, fcap = 0;     // list capacities the context was created with (0 = defaults)
    unsigned long used = 0;
};
struct ThreadCache {
    static constexpr int N = 4;
    CachedCtx e[N];
    unsigned long tick = 0;
    // device scratch of scalespace_compute_gradient: one stack in, two out
    float* d_grad = nullptr;
    size_t grad_floats = 0;
    int grad_device = -1;
    // pinned staging of the scale-space downloads (two halves used in turn) and their stream
    char* pin = nullptr;
    hipStream_t copy_stream = nullptr;
    int copy_device = -1;
    void release()
    {
        for (auto& c : e)
            if (c.ctx) { sift_amd_destroy(c.ctx); c.ctx = nullptr; }
        if (pin) { (void)hipHostFree(pin); pin = nullptr; }
        if (copy_stream) {
            int prev = -1;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(copy_device);
            (void)hipStreamDestroy(copy_stream);
            if (prev >= 0) (void)hipSetDevice(prev);
            copy_stream = nullptr;
        }
        if (d_grad) {
            int prev = -1;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(grad_device);
            (void)hipFree(d_grad);
            if (prev >= 0) (void)hipSetDevice(prev);
            d_grad = nullptr; grad_floats = 0;
        }
    }
    ~ThreadCache() { release(); }
};
static thread_local ThreadCache g_cache;

static CachedCtx* cache_slot()       // an empty slot, else the least recently used one (emptied)
{
    CachedCtx* pick = &g_cache.e[0];
    for (auto& c : g_cache.e) {
        if (!c.ctx) { pick = &c; break; }
        if (c.used < pick->used) pick = &c;
    }
    if (pick->ctx) { sift_amd_destroy(pick->ctx); pick->ctx = nullptr; }
    return pick;
}

// cap/fcap: list capacities (0 = defaults); a cached context with at least these is reused
static sift_amd_ctx* get_ctx(int w, int h, const sift_parameters* p, int flags, int cap = 0, int fcap = 0)
{
    const int dev = env_device();
    for (auto& c : g_cache.e)
        if (c.ctx && c.lowe && c.w == w && c.h == h && c.flags == flags && c.device == dev &&
            memcmp(&c.p, p, sizeof *p) == 0 && c.cap >= cap && c.fcap >= fcap) {
            c.used = ++g_cache.tick;
            return c.ctx;
        }
    // a context of the same key with smaller lists is replaced, not kept beside the new one
    for (auto& c : g_cache.e)
        if (c.ctx && c.lowe && c.w == w && c.h == h && c.flags == flags && c.device == dev &&
            memcmp(&c.p, p, sizeof *p) == 0) { sift_amd_destroy(c.ctx); c.ctx = nullptr; }
    CachedCtx* cc = cache_slot();
    SaLayout L;
    memset(&L, 0, sizeof L);
    const int n_oct = sa_number_of_octaves(w, h, p);
    if (p->n_spo < 1 || p->n_spo + 3 > SA_MAX_SCA || n_oct < 1 || n_oct > SA_MAX_OCT) {
        sa_set_error("image %dx%d with n_spo %d gives %d octaves (supported: 1..%d octaves, n_spo 1..%d)",
                     w, h, p->n_spo, n_oct, SA_MAX_OCT, SA_MAX_SCA - 3);
        die("cannot create context");
    }
    sa_layout_lowe(&L, n_oct, p->n_spo, w, h, p->delta_min, p->sigma_min);
    sift_amd_ctx* c = sa_create_from_layout(dev, w, h, 1, p, &L, p->sigma_in, flags, cap, fcap);
    if (!c) die("cannot create context");
    cc->ctx = c; cc->w = w; cc->h = h; cc->flags = flags; cc->device = dev; cc->p = *p; cc->lowe = true;
    int kcap;
    sa_ctx_limits(c, &cc->cap, &cc->fcap, &kcap);
    cc->used = ++g_cache.tick;
    return c;
}

// Context for a caller-supplied scale-space geometry (scalespace_compute).
static sift_amd_ctx* get_ctx_layout(int w, int h, const sift_parameters* p, const SaLayout* L, float sigma_in)
{
    const int dev = env_device();
    for (auto& c : g_cache.e)
        if (c.ctx && !c.lowe && c.w == w && c.h == h && c.device == dev && c.sigma_in == sigma_in &&
            memcmp(&c.L, L, sizeof *L) == 0) {
            c.used = ++g_cache.tick;
            return c.ctx;
        }
    CachedCtx* cc = cache_slot();
    sift_amd_ctx* c = sa_create_from_layout(dev, w, h, 1, p, L, sigma_in, 0, 0, 0);
    if (!c) die("scalespace_compute");
    cc->ctx = c; cc->w = w; cc->h = h; cc->flags = 0; cc->device = dev; cc->p = *p; cc->lowe = false;
    cc->L = *L; cc->sigma_in = sigma_in; cc->cap = cc->fcap = 0; cc->used = ++g_cache.tick;
    return c;
}

extern "C" void sift_amd_release_thread_cache(void) { g_cache.release(); }

// A list of the last run overflowed: the capacities that certainly hold it.  The scan counts
// every candidate it meets, also those beyond the list's end, so `cap` is exact; how many
// survive is only known for the stored ones, so the survivor lists get room for all of them.
static void grown_caps(sift_amd_ctx* c, int* cap, int* fcap)
{
    int cur_cap, cur_fcap, cur_kcap;
    sa_ctx_limits(c, &cur_cap, &cur_fcap, &cur_kcap);
    const int n_cand = sa_ctx_candidates_seen(c);
    *cap = n_cand > cur_cap ? n_cand : cur_cap;
    *fcap = *cap;
}

// -------------------------------------------------------------- containers --
// Same allocation contract as lib_keypoint.c:75-182 and lib_scalespace.c:91-151: every
// object is its own malloc block.

static void* must_malloc(size_t n)
{
    void* p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "sift_amd: out of host memory (%zu bytes)\n", n); abort(); }
    return p;
}

extern "C" struct keypoint* sift_malloc_keypoint(int n_ori, int n_hist, int n_bins)
{
    struct keypoint* k = (struct keypoint*)must_malloc(sizeof *k);
    memset(k, 0, sizeof *k);
    k->n_ori = n_ori; k->n_hist = n_hist; k->n_bins = n_bins;
    const int nd = n_ori * n_hist * n_hist;
    k->descr = (float*)must_malloc(nd * sizeof(float));
    k->orihist = (float*)must_malloc(n_bins * sizeof(float));
    for (int n = 0; n < nd; n++) k->descr[n] = 0.f;
    for (int n = 0; n < n_bins; n++) k->orihist[n] = 0.f;
    return k;
}

extern "C" struct keypoint* sift_malloc_keypoint_from_model_and_copy(const struct keypoint* src)
{
    struct keypoint* k = sift_malloc_keypoint(src->n_ori, src->n_hist, src->n_bins);
    k->x = src->x; k->y = src->y; k->sigma = src->sigma; k->theta = src->theta;
    k->o = src->o; k->s = src->s; k->i = src->i; k->j = src->j;
    k->val = src->val; k->edgeResp = src->edgeResp;
    memcpy(k->descr, src->descr, sizeof(float) * src->n_ori * src->n_hist * src->n_hist);
    memcpy(k->orihist, src->orihist, sizeof(float) * src->n_bins);
    return k;
}

extern "C" struct sift_keypoints* sift_malloc_keypoints(void)
{
    struct sift_keypoints* l = (struct sift_keypoints*)must_malloc(sizeof *l);
    l->capacity = 100;                       // lib_keypoint.c:142-144
    l->size = 0;
    l->list = (struct keypoint**)must_malloc(l->capacity * sizeof(struct keypoint*));
    return l;
}

extern "C" void sift_add_keypoint_to_list(struct keypoint* key, struct sift_keypoints* keys)
{
    if (keys->size > keys->capacity - 1) {   // doubling growth, lib_keypoint.c:149-164
        keys->capacity *= 2;
        keys->list = (struct keypoint**)realloc(keys->list, keys->capacity * sizeof(struct keypoint*));
        if (!keys->list) { fprintf(stderr, "sift_amd: realloc failed\n"); abort(); }
    }
    keys->list[keys->size++] = key;
}

extern "C" void sift_free_keypoint(struct keypoint* key)
{
    free(key->orihist);
    free(key->descr);
    free(key);
}

extern "C" void sift_free_keypoints(struct sift_keypoints* keys)
{
    for (int k = 0; k < keys->size; k++) sift_free_keypoint(keys->list[k]);
    free(keys->list);
    free(keys);
}

extern "C" struct sift_scalespace* sift_malloc_scalespace(int nOct, const float* deltas, const int* ws,
                                                           const int* hs, const int* nScas, float** sigmas)
{
    struct sift_scalespace* ss = (struct sift_scalespace*)must_malloc(sizeof *ss);
    ss->nOct = nOct;
    for (int o = 0; o < nOct; o++) {
        struct octa* oc = (struct octa*)must_malloc(sizeof *oc);
        oc->delta = deltas[o]; oc->w = ws[o]; oc->h = hs[o]; oc->nSca = nScas[o];
        oc->sigmas = (float*)must_malloc(nScas[o] * sizeof(float));
        oc->imStack = (float*)must_malloc((size_t)nScas[o] * ws[o] * hs[o] * sizeof(float));
        for (int s = 0; s < nScas[o]; s++) oc->sigmas[s] = sigmas[o][s];
        ss->octaves[o] = oc;
    }
    return ss;
}

extern "C" void sift_free_scalespace(struct sift_scalespace* ss)
{
    for (int o = 0; o < ss->nOct; o++) {
        free(ss->octaves[o]->sigmas);
        free(ss->octaves[o]->imStack);
        free(ss->octaves[o]);
    }
    free(ss);
}

// shape = model's octave sizes with `drop` fewer images per octave (lib_scalespace.c:164-198, 266-297)
static struct sift_scalespace* scalespace_like(const struct sift_scalespace* m, int drop)
{
    const int n = m->nOct;
    std::vector<float> deltas(n);
    std::vector<int> ws(n), hs(n), ns(n);
    std::vector<float*> sig(n);
    for (int o = 0; o < n; o++) {
        deltas[o] = m->octaves[o]->delta; ws[o] = m->octaves[o]->w; hs[o] = m->octaves[o]->h;
        ns[o] = m->octaves[o]->nSca - drop;
        sig[o] = m->octaves[o]->sigmas;
    }
    return sift_malloc_scalespace(n, deltas.data(), ws.data(), hs.data(), ns.data(), sig.data());
}

extern "C" struct sift_scalespace* sift_malloc_scalespace_from_model(struct sift_scalespace* m)
{
    return scalespace_like(m, 0);
}

extern "C" struct sift_scalespace* sift_malloc_scalespace_dog_from_scalespace(struct sift_scalespace* m)
{
    return scalespace_like(m, 1);
}

static struct sift_scalespace* scalespace_from_layout(const SaLayout* L, int drop)
{
    std::vector<int> ns(L->n_oct, L->n_sca - drop);
    std::vector<float*> sig(L->n_oct);
    for (int o = 0; o < L->n_oct; o++) sig[o] = const_cast<float*>(L->sigmas[o]);
    return sift_malloc_scalespace(L->n_oct, L->delta, L->w, L->h, ns.data(), sig.data());
}

extern "C" struct sift_scalespace* sift_malloc_scalespace_lowe(int nOct, int nSca, int im_w, int im_h,
                                                                float delta_min, float sigma_min)
{
    if (nOct > SA_MAX_OCT || nSca + 3 > SA_MAX_SCA || nOct < 0) {
        fprintf(stderr, "sift_amd: scale-space shape %d octaves x %d scales unsupported\n", nOct, nSca);
        abort();
    }
    SaLayout L;
    memset(&L, 0, sizeof L);
    sa_layout_lowe(&L, nOct, nSca, im_w, im_h, delta_min, sigma_min);
    return scalespace_from_layout(&L, 0);
}

// ------------------------------------------------------------- conversions --

static struct keypoint* key_from_record(const sift_amd_key& r, const sift_parameters* p)
{
    struct keypoint* k = sift_malloc_keypoint(p->n_ori, p->n_hist, p->n_bins);
    k->x = r.x; k->y = r.y; k->sigma = r.sigma; k->theta = r.theta;
    k->o = r.o; k->s = r.s; k->i = r.i; k->j = r.j;
    k->val = r.val; k->edgeResp = r.edge;
    return k;
}

// The calling thread's own non-blocking stream on the current device (the library stays off the
// legacy default stream: its implicit synchronisation fails while another host thread captures a
// hipGraph).
static hipStream_t thread_stream()
{
    ThreadCache& tc = g_cache;
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (tc.copy_stream && tc.copy_device != dev) { (void)hipStreamDestroy(tc.copy_stream); tc.copy_stream = nullptr; }
    if (!tc.copy_stream) {
        if (hipStreamCreateWithFlags(&tc.copy_stream, hipStreamNonBlocking) != hipSuccess) die("thread stream");
        tc.copy_device = dev;
    }
    return tc.copy_stream;
}
static bool copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    hipStream_t st = thread_stream();
    return hipMemcpyAsync(dst, src, bytes, kind, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
}

// Device memory into a freshly malloc'd (pageable, not yet touched) host block.  A plain hipMemcpy
// spends most of its time in the page faults of the destination; here the bytes arrive in pinned
// staging memory, 32 MB at a time, and a few host threads copy each piece to its place (and take
// the page faults side by side) while the next piece is on its way.
static void download_to_pageable(void* dst, const void* d_src, size_t bytes)
{
    constexpr size_t CH = 32u << 20;
    if (bytes < (4u << 20)) {
        if (!copy_sync(dst, d_src, bytes, hipMemcpyDeviceToHost)) die("download");
        return;
    }
    ThreadCache& tc = g_cache;
    (void)thread_stream();
    if (!tc.pin && hipHostMalloc((void**)&tc.pin, 2 * CH, hipHostMallocDefault) != hipSuccess) die("pinned staging");
    static const unsigned n_thr = [] {
        const char* e = getenv("SIFT_AMD_COPY_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return n < 1 ? 1u : (n > 8 ? 8u : n);
    }();
    auto issue = [&](size_t off, int half) {
        const size_t n = std::min(CH, bytes - off);
        if (hipMemcpyAsync(tc.pin + half * CH, (const char*)d_src + off, n, hipMemcpyDeviceToHost, tc.copy_stream) != hipSuccess)
            die("download");
    };
    issue(0, 0);
    int half = 0;
    for (size_t off = 0; off < bytes; off += CH, half ^= 1) {
        if (hipStreamSynchronize(tc.copy_stream) != hipSuccess) die("download");
        if (off + CH < bytes) issue(off + CH, half ^ 1);
        const size_t n = std::min(CH, bytes - off);
        const size_t per = ((n + n_thr - 1) / n_thr + 4095) & ~(size_t)4095;
        std::vector<std::thread> workers;
        char* to = (char*)dst + off;
        const char* from = tc.pin + half * CH;
        for (unsigned t = 1; t < n_thr && t * per < n; t++) {
            const size_t at = t * per, len = std::min(per, n - at);
            workers.emplace_back([to, from, at, len] { memcpy(to + at, from + at, len); });
        }
        memcpy(to, from, std::min(per, n));
        for (auto& w : workers) w.join();
    }
}

struct DeviceScope {     // the reference-named calls run on SIFT_AMD_DEVICE and leave the caller's device as it was
    int prev = -1;
    explicit DeviceScope(int dev) { (void)hipGetDevice(&prev); if (hipSetDevice(dev) != hipSuccess) { sa_set_error("hipSetDevice failed"); die("device"); } }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static void download_stack(sift_amd_ctx* c, int which, struct sift_scalespace* ss)
{
    DeviceScope on(env_device());
    for (int o = 0; o < ss->nOct; o++) {
        const struct octa* oc = ss->octaves[o];
        const float* d = sa_ctx_stack(c, which, o);
        if (!d) die("copy scale-space");
        download_to_pageable(oc->imStack, d, (size_t)oc->nSca * oc->w * oc->h * sizeof(float));
    }
}

// ss[2], ss[3] of sift_anatomy (lib_sift_anatomy.c:202-218, 842-844) from the Gaussian stack that
// is still on the device.
static void gradient_stacks_from_ctx(sift_amd_ctx* c, struct sift_scalespace* sx, struct sift_scalespace* sy);

// d/di and d/dj of every image of a host scale-space (lib_sift_anatomy.c:202-218).  The device
// scratch (one stack in, two out, sized for the largest octave) is kept per host thread.
static void gradient_stacks(const struct sift_scalespace* s, struct sift_scalespace* sx, struct sift_scalespace* sy)
{
    const int dev = env_device();
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(dev) != hipSuccess) { sa_set_error("hipSetDevice failed"); die("gradient"); }
    size_t biggest = 0;
    for (int o = 0; o < s->nOct; o++) {
        const struct octa* oc = s->octaves[o];
        biggest = std::max(biggest, (size_t)oc->nSca * oc->w * oc->h);
    }
    ThreadCache& tc = g_cache;
    if (tc.grad_floats < biggest || tc.grad_device != dev) {
        if (tc.d_grad) { (void)hipSetDevice(tc.grad_device); (void)hipFree(tc.d_grad); (void)hipSetDevice(dev); }
        tc.d_grad = nullptr; tc.grad_floats = 0;
        if (hipMalloc((void**)&tc.d_grad, 3 * biggest * sizeof(float)) != hipSuccess) { sa_set_error("hipMalloc failed"); die("gradient"); }
        tc.grad_floats = biggest; tc.grad_device = dev;
    }
    float *d_im = tc.d_grad, *d_gi = tc.d_grad + tc.grad_floats, *d_gj = tc.d_grad + 2 * tc.grad_floats;
    for (int o = 0; o < s->nOct; o++) {
        const struct octa* oc = s->octaves[o];
        const size_t bytes = (size_t)oc->nSca * oc->w * oc->h * 4;
        bool ok = copy_sync(d_im, oc->imStack, bytes, hipMemcpyHostToDevice);
        sa::launch_gradient(thread_stream(), d_im, d_gi, d_gj, oc->w, oc->h, oc->nSca);
        ok = ok && copy_sync(sx->octaves[o]->imStack, d_gi, bytes, hipMemcpyDeviceToHost);
        ok = ok && copy_sync(sy->octaves[o]->imStack, d_gj, bytes, hipMemcpyDeviceToHost);
        if (!ok) { sa_set_error("gradient transfer failed: %s", hipGetErrorString(hipGetLastError())); die("gradient"); }
    }
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
}

static void gradient_stacks_from_ctx(sift_amd_ctx* c, struct sift_scalespace* sx, struct sift_scalespace* sy)
{
    const int dev = env_device();
    DeviceScope on(dev);
    size_t biggest = 0;
    for (int o = 0; o < sx->nOct; o++) {
        const struct octa* oc = sx->octaves[o];
        biggest = std::max(biggest, (size_t)oc->nSca * oc->w * oc->h);
    }
    ThreadCache& tc = g_cache;
    if (tc.grad_floats < biggest || tc.grad_device != dev) {
        if (tc.d_grad) { (void)hipSetDevice(tc.grad_device); (void)hipFree(tc.d_grad); (void)hipSetDevice(dev); }
        tc.d_grad = nullptr; tc.grad_floats = 0;
        if (hipMalloc((void**)&tc.d_grad, 3 * biggest * sizeof(float)) != hipSuccess) { sa_set_error("hipMalloc failed"); die("gradient"); }
        tc.grad_floats = biggest; tc.grad_device = dev;
    }
    float *d_gi = tc.d_grad + tc.grad_floats, *d_gj = tc.d_grad + 2 * tc.grad_floats;
    for (int o = 0; o < sx->nOct; o++) {
        const struct octa* oc = sx->octaves[o];
        const float* d = sa_ctx_stack(c, 0, o);
        if (!d) die("gradient");
        sa::launch_gradient(thread_stream(), d, d_gi, d_gj, oc->w, oc->h, oc->nSca);
        if (hipStreamSynchronize(thread_stream()) != hipSuccess) die("gradient");
        const size_t bytes = (size_t)oc->nSca * oc->w * oc->h * sizeof(float);
        download_to_pageable(sx->octaves[o]->imStack, d_gi, bytes);
        download_to_pageable(sy->octaves[o]->imStack, d_gj, bytes);
    }
}

// ---------------------------------------------------------- reference API --

extern "C" struct sift_parameters* sift_assign_default_parameters(void)
{
    struct sift_parameters* p = (struct sift_parameters*)must_malloc(sizeof *p);
    sa_default_parameters(p);
    return p;
}

// Runs detection (+ description) for one host image and fills the intermediate lists.
// n_lists = 6 for sift_anatomy (A..F), 5 for the variant without description (A..E).
static struct sift_keypoints* run_full(const float* x, int w, int h, const sift_parameters* p,
                                       struct sift_scalespace** ss, int n_ss, struct sift_keypoints** kk,
                                       int n_lists, bool describe)
{
    // The reference's lists grow as needed; here a run whose lists overflow is repeated once in
    // a context whose lists hold what the first run counted.
    sift_amd_ctx* c = nullptr;
    int cap = 0, fcap = 0;
    for (int attempt = 0;; attempt++) {
        c = get_ctx(w, h, p, SIFT_AMD_KEEP_ALL_EXTREMA | SIFT_AMD_KEEP_SCALESPACE, cap, fcap);
        hipStream_t st = sa_ctx_stream(c);
        if (hipMemcpyAsync(sa_ctx_input_buffer(c), x, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) {
            sa_set_error("input upload failed");
            die("sift_anatomy");
        }
        int rc;
        if (describe) {
            rc = sift_amd_detect_device(c, sa_ctx_input_buffer(c), 1, nullptr);
        } else {
            rc = sift_amd_scalespace_device(c, sa_ctx_input_buffer(c), 1, nullptr);
            if (!rc) rc = sa_run_detect(c, 1, st);
        }
        if (rc) die("pipeline");
        rc = sa_sync_counters(c);
        if (rc == -2 && attempt == 0) { grown_caps(c, &cap, &fcap); continue; }
        if (rc) die("pipeline");
        break;
    }

    const SaLayout* L = sa_ctx_layout(c);
    // scale-spaces
    ss[0] = scalespace_from_layout(L, 0);
    ss[1] = scalespace_from_layout(L, 1);
    download_stack(c, 0, ss[0]);
    download_stack(c, 1, ss[1]);
    if (n_ss == 4) {
        ss[2] = scalespace_from_layout(L, 0);
        ss[3] = scalespace_from_layout(L, 0);
        gradient_stacks_from_ctx(c, ss[2], ss[3]);
    }

    // stage lists, in canonical order
    int sizes[6], ext_oct[SIFT_AMD_MAX_OCT];
    if (sift_amd_stage_counts(c, 0, sizes, ext_oct)) die("stage counts");
    const int n = sizes[0];
    std::vector<sift_amd_key> seed(n), ref(n);
    std::vector<int> stage(n);
    const int got = sift_amd_fetch_candidates(c, 0, seed.data(), ref.data(), stage.data(), n);
    if (got != n) { sa_set_error("candidate list truncated (%d of %d)", got, n); die("sift_anatomy"); }

    const int nb = p->n_bins, nd = p->n_hist * p->n_hist * p->n_ori;
    int n_keys = 0;
    std::vector<sift_amd_key> keys;
    std::vector<float> descr, fhist;
    std::vector<int> src;
    if (describe) {
        if (sift_amd_keypoint_counts(c, &n_keys)) die("counts");
        keys.resize(n_keys); descr.resize((size_t)n_keys * nd); src.resize(n_keys);
        if (n_keys) {
            if (sift_amd_fetch_keypoints(c, 0, nullptr, keys.data(), n_keys) != n_keys) die("fetch keys");
            if (sift_amd_fetch_extras(c, 0, nullptr, descr.data(), src.data(), n_keys) != n_keys) die("fetch extras");
        }
        fhist.resize((size_t)sizes[5] * nb);
        if (sizes[5] && sift_amd_fetch_stage_f_histograms(c, 0, fhist.data(), sizes[5]) != sizes[5]) die("fetch histograms");
    }

    struct sift_keypoints* result = sift_malloc_keypoints();
    int f_rank = 0, key_pos = 0;
    for (int r = 0; r < n; r++) {
        sift_add_keypoint_to_list(key_from_record(seed[r], p), kk[0]);
        if (stage[r] >= 1) sift_add_keypoint_to_list(key_from_record(seed[r], p), kk[1]);
        sift_amd_key q = ref[r];
        const float edge = q.edge;
        q.edge = 0.f;
        if (stage[r] >= 2) sift_add_keypoint_to_list(key_from_record(q, p), kk[2]);
        q.edge = edge;
        if (stage[r] >= 3) sift_add_keypoint_to_list(key_from_record(q, p), kk[3]);
        if (stage[r] >= 4) sift_add_keypoint_to_list(key_from_record(q, p), kk[4]);
        if (stage[r] >= 5) {
            struct keypoint* kf = key_from_record(q, p);
            if (describe) {
                // the reference accumulates the histogram inside the list-F entry and copies it
                // into every oriented keypoint derived from it (lib_sift_anatomy.c:628-640)
                const float* hist = &fhist[(size_t)f_rank * nb];
                memcpy(kf->orihist, hist, nb * sizeof(float));
                while (key_pos < n_keys && src[key_pos] == f_rank) {
                    struct keypoint* kfin = key_from_record(keys[key_pos], p);
                    memcpy(kfin->orihist, hist, nb * sizeof(float));
                    memcpy(kfin->descr, &descr[(size_t)key_pos * nd], nd * sizeof(float));
                    sift_add_keypoint_to_list(kfin, result);
                    key_pos++;
                }
            }
            if (n_lists == 6) sift_add_keypoint_to_list(kf, kk[5]);
            else sift_add_keypoint_to_list(kf, result);
            f_rank++;
        }
    }
    return result;
}

extern "C" struct sift_keypoints* sift_anatomy(const float* x, int w, int h, const struct sift_parameters* p,
                                               struct sift_scalespace* ss[4], struct sift_keypoints* kk[6])
{
    return run_full(x, w, h, p, ss, 4, kk, 6, true);
}

extern "C" struct sift_keypoints* sift_anatomy_without_description(const float* x, int w, int h,
                                                                   const struct sift_parameters* p,
                                                                   struct sift_scalespace* ss[2],
                                                                   struct sift_keypoints* kk[5])
{
    return run_full(x, w, h, p, ss, 2, kk, 5, false);
}

static void describe_list(const float* x, int w, int h, const sift_parameters* p, struct sift_keypoints* k,
                          bool find_orientation)
{
    sift_amd_ctx* c = get_ctx(w, h, p, 0, 0, k->size);      // survivor lists with room for the caller's keypoints
    hipStream_t st = sa_ctx_stream(c);
    if (hipMemcpyAsync(sa_ctx_input_buffer(c), x, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) {
        sa_set_error("input upload failed");
        die("describe");
    }
    if (sift_amd_scalespace_device(c, sa_ctx_input_buffer(c), 1, nullptr)) die("scale-space");
    const SaLayout* L = sa_ctx_layout(c);
    const int n = k->size;
    std::vector<SaKey> in(n);
    for (int q = 0; q < n; q++) {
        const struct keypoint* kp = k->list[q];
        if (kp->o < 0 || kp->o >= L->n_oct || kp->s < 0 || kp->s >= L->n_sca) {
            sa_set_error("keypoint %d refers to octave %d scale %d outside the scale-space", q, kp->o, kp->s);
            die("describe");
        }
        SaKey& r = in[q];
        r.x = kp->x; r.y = kp->y; r.sigma = kp->sigma; r.theta = kp->theta;
        r.o = kp->o; r.s = kp->s; r.i = kp->i; r.j = kp->j; r.val = kp->val; r.edge = kp->edgeResp;
    }
    if (sa_describe_given(c, in.data(), n, find_orientation)) die("describe");
    const int nb = p->n_bins, nd = p->n_hist * p->n_hist * p->n_ori;
    std::vector<sift_amd_key> out(n);
    std::vector<float> orihist((size_t)n * nb), descr((size_t)n * nd);
    if (n) {
        if (sift_amd_fetch_keypoints(c, 0, nullptr, out.data(), n) != n) die("fetch keys");
        if (sift_amd_fetch_extras(c, 0, find_orientation ? orihist.data() : nullptr, descr.data(), nullptr, n) != n) die("fetch extras");
    }
    for (int q = 0; q < n; q++) {
        struct keypoint* kp = k->list[q];
        if (find_orientation) {
            kp->theta = out[q].theta;
            memcpy(kp->orihist, &orihist[(size_t)q * nb], nb * sizeof(float));
        }
        memcpy(kp->descr, &descr[(size_t)q * nd], nd * sizeof(float));
    }
}

extern "C" void sift_anatomy_only_description(const float* x, int w, int h, const struct sift_parameters* p,
                                              struct sift_keypoints* k)
{
    describe_list(x, w, h, p, k, false);
}

extern "C" void sift_anatomy_orientation_and_description(const float* x, int w, int h,
                                                         const struct sift_parameters* p,
                                                         struct sift_keypoints* k)
{
    describe_list(x, w, h, p, k, true);
}

extern "C" void scalespace_compute(struct sift_scalespace* ss, const float* image, int im_w, int im_h,
                                   float sigma_in)
{
    SaLayout L;
    memset(&L, 0, sizeof L);
    if (ss->nOct < 1 || ss->nOct > SA_MAX_OCT) { sa_set_error("%d octaves unsupported", ss->nOct); die("scalespace_compute"); }
    L.n_oct = ss->nOct;
    L.n_sca = ss->octaves[0]->nSca;
    if (L.n_sca > SA_MAX_SCA) { sa_set_error("%d scales unsupported", L.n_sca); die("scalespace_compute"); }
    for (int o = 0; o < ss->nOct; o++) {
        const struct octa* oc = ss->octaves[o];
        if (oc->nSca != L.n_sca) { sa_set_error("octaves with different numbers of scales"); die("scalespace_compute"); }
        L.w[o] = oc->w; L.h[o] = oc->h; L.delta[o] = oc->delta;
        for (int s = 0; s < oc->nSca; s++) L.sigmas[o][s] = oc->sigmas[s];
    }
    // the checks of lib_sift_anatomy.c:108-109
    if (L.w[0] != (int)((float)im_w / L.delta[0]) || L.h[0] != (int)((float)im_h / L.delta[0])) {
        sa_set_error("scale-space does not match the image size");
        die("scalespace_compute");
    }
    sift_parameters p;
    sa_default_parameters(&p);
    p.n_spo = L.n_sca - 3; p.delta_min = L.delta[0]; p.sigma_min = L.sigmas[0][0]; p.sigma_in = sigma_in;
    // only the Gaussian stack is handed back: the context stores no DoG images, and it stays
    // cached for the next call with this geometry
    sift_amd_ctx* c = get_ctx_layout(im_w, im_h, &p, &L, sigma_in);
    if (hipMemcpyAsync(sa_ctx_input_buffer(c), image, (size_t)im_w * im_h * sizeof(float), hipMemcpyHostToDevice,
                       sa_ctx_stream(c)) != hipSuccess) {
        sa_set_error("input upload failed");
        die("scalespace_compute");
    }
    if (sift_amd_scalespace_device(c, sa_ctx_input_buffer(c), 1, nullptr)) die("scalespace_compute");
    download_stack(c, 0, ss);
}

extern "C" void scalespace_compute_gradient(const struct sift_scalespace* s, struct sift_scalespace* sx,
                                            struct sift_scalespace* sy)
{
    gradient_stacks(s, sx, sy);
}

// ------------------------------------------------ flat API (lib_sift.c) ----

extern "C" struct sift_keypoint_std* sift_compute_features(const float* x, int w, int h, int* n)
{
    sift_parameters p;
    sa_default_parameters(&p);
    sift_amd_ctx* c = nullptr;
    int cnt = 0, cap = 0, fcap = 0;
    for (int attempt = 0;; attempt++) {
        c = get_ctx(w, h, &p, 0, cap, fcap);
        if (sift_amd_detect_host(c, x, 1)) die("sift_compute_features");
        const int rc = sift_amd_keypoint_counts(c, &cnt);
        if (rc == -2 && attempt == 0) { grown_caps(c, &cap, &fcap); continue; }
        if (rc) die("sift_compute_features");
        break;
    }
    struct sift_keypoint_std* k = (struct sift_keypoint_std*)must_malloc((size_t)cnt * sizeof *k);
    if (cnt && sift_amd_fetch_keypoints(c, 0, k, nullptr, cnt) != cnt) die("sift_compute_features");
    *n = cnt;
    return k;
}

// Keypoints after the border filter, orientation 0 and empty descriptors: what
// lib_sift.c:192-236 returns (it calls the variant without description).
extern "C" struct sift_keypoint_std* sift_compute_points(const float* x, int w, int h, int* n)
{
    sift_parameters p;
    sa_default_parameters(&p);
    sift_amd_ctx* c = nullptr;
    int sizes[6], ext[SIFT_AMD_MAX_OCT], cap = 0, fcap = 0;
    for (int attempt = 0;; attempt++) {
        c = get_ctx(w, h, &p, 0, cap, fcap);
        hipStream_t st = sa_ctx_stream(c);
        if (hipMemcpyAsync(sa_ctx_input_buffer(c), x, (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) {
            sa_set_error("input upload failed");
            die("sift_compute_points");
        }
        if (sift_amd_scalespace_device(c, sa_ctx_input_buffer(c), 1, nullptr) || sa_run_detect(c, 1, st)) die("sift_compute_points");
        const int rc = sift_amd_stage_counts(c, 0, sizes, ext);
        if (rc == -2 && attempt == 0) { grown_caps(c, &cap, &fcap); continue; }
        if (rc) die("sift_compute_points");
        break;
    }
    const int nc = sift_amd_fetch_candidates(c, 0, nullptr, nullptr, nullptr, 0);   // list B
    if (nc < 0) die("sift_compute_points");
    std::vector<sift_amd_key> ref(nc);
    std::vector<int> stage(nc);
    if (sift_amd_fetch_candidates(c, 0, nullptr, ref.data(), stage.data(), nc) != nc) die("sift_compute_points");
    struct sift_keypoint_std* k = (struct sift_keypoint_std*)must_malloc((size_t)sizes[5] * sizeof *k);
    int m = 0;
    for (int r = 0; r < nc; r++)
        if (stage[r] >= 5) {
            memset(&k[m], 0, sizeof k[m]);
            k[m].x = ref[r].x; k[m].y = ref[r].y; k[m].scale = ref[r].sigma; k[m].orientation = 0.f;
            m++;
        }
    *n = m;
    return k;
}

// Infers (o,s,i,j) from the scale as lib_sift.c:73-113 does.
static struct sift_keypoints* std_to_list(const struct sift_keypoint_std* k, int n, const sift_parameters* p)
{
    struct sift_keypoints* keys = sift_malloc_keypoints();
    for (int q = 0; q < n; q++) {
        struct keypoint* key = sift_malloc_keypoint(p->n_ori, p->n_hist, p->n_bins);
        key->x = k[q].x; key->y = k[q].y; key->sigma = k[q].scale; key->theta = k[q].orientation;
        int o, s;
        const int a = (int)(round(p->n_spo * log((double)(k[q].scale / p->sigma_min)) / 0.693147180559945309417));
        o = (a - 1) / p->n_spo;
        if (o > -1) { s = (a - 1) % p->n_spo + 1; } else { o = 0; s = 0; }
        key->o = o; key->s = s;
        const double d = (double)p->delta_min * exp(o * 0.693147180559945309417);
        key->i = (int)((double)key->x / d + 0.5);
        key->j = (int)((double)key->y / d + 0.5);
        sift_add_keypoint_to_list(key, keys);
    }
    return keys;
}

static void list_to_std(const struct sift_keypoints* keys, struct sift_keypoint_std* k)
{
    for (int q = 0; q < keys->size; q++) {
        const struct keypoint* key = keys->list[q];
        k[q].x = key->x; k[q].y = key->y; k[q].scale = key->sigma; k[q].orientation = key->theta;
        for (int j = 0; j < 128; j++) k[q].descriptor[j] = (unsigned char)(key->descr[j]);
    }
}

extern "C" void sift_fill_descriptors(const float* x, int w, int h, struct sift_keypoint_std* k, int n)
{
    sift_parameters p;
    sa_default_parameters(&p);
    struct sift_keypoints* keys = std_to_list(k, n, &p);
    sift_anatomy_only_description(x, w, h, &p, keys);
    list_to_std(keys, k);
    sift_free_keypoints(keys);
}

extern "C" void sift_find_ori_and_fill_descriptors(const float* x, int w, int h, struct sift_keypoint_std* k, int n)
{
    sift_parameters p;
    sa_default_parameters(&p);
    struct sift_keypoints* keys = std_to_list(k, n, &p);
    sift_anatomy_orientation_and_description(x, w, h, &p, keys);
    list_to_std(keys, k);
    sift_free_keypoints(keys);
}

// ------------------------------------------- keypoint text format (I/O) ----
// lib_keypoint.c:184-297, lib_sift.c:287-367.

extern "C" void fprintf_one_keypoint(FILE* f, const struct keypoint* k, int n_descr, int n_bins, int flag)
{
    fprintf(f, "%f %f %f %f ", k->x, k->y, k->sigma, k->theta);
    if (flag > 0)
        for (int n = 0; n < n_descr; n++) fprintf(f, "%i ", (int)k->descr[n]);
    if (flag > 1)
        for (int n = 0; n < n_bins; n++) fprintf(f, "%f ", k->orihist[n]);
}

extern "C" void fprintf_keypoints(FILE* f, const struct sift_keypoints* keys, int flag)
{
    if (keys->size <= 0) return;
    const struct keypoint* k0 = keys->list[0];
    const int n_descr = k0->n_hist * k0->n_hist * k0->n_ori, n_bins = k0->n_bins;
    for (int k = 0; k < keys->size; k++) {
        fprintf_one_keypoint(f, keys->list[k], n_descr, n_bins, flag);
        fprintf(f, "\n");
    }
}

extern "C" void sift_save_keypoints(const struct sift_keypoints* keys, const char* name, int flag)
{
    FILE* f = fopen(name, "w");
    if (!f) { fprintf(stderr, "Fatal error: Failed to open %s for writing\n\n", name); exit(1); }
    fprintf_keypoints(f, keys, flag);
    fclose(f);
}

extern "C" void sift_print_keypoints(const struct sift_keypoints* keys, int flag)
{
    fprintf_keypoints(stdout, keys, flag);
}

extern "C" void sift_read_keypoints(struct sift_keypoints* keys, const char* name, int n_hist, int n_ori,
                                    int n_bins, int flag)
{
    const size_t cap = 1024 * 1024;
    char* line = (char*)must_malloc(cap);
    FILE* f = fopen(name, "r");
    if (!f) { fprintf(stderr, "Fatal error: File \"%s\" not found.\n", name); exit(1); }
    while (fgets(line, (int)cap, f)) {
        int pos = 0, used = 0;
        struct keypoint* key = sift_malloc_keypoint(n_ori, n_hist, n_bins);
        sscanf(line + pos, "%f  %f  %f  %f %n", &key->x, &key->y, &key->sigma, &key->theta, &used);
        pos += used;
        if (flag > 0)
            for (int i = 0; i < n_hist * n_hist * n_ori; i++) {
                sscanf(line + pos, "%f %n", &key->descr[i], &used);
                pos += used;
            }
        if (flag > 1)
            for (int i = 0; i < n_bins; i++) {
                sscanf(line + pos, "%f %n", &key->orihist[i], &used);
                pos += used;
            }
        sift_add_keypoint_to_list(key, keys);
    }
    fclose(f);
    free(line);
}

extern "C" void fprintf_keypoint_std(FILE* f, const struct sift_keypoint_std* k, int n)
{
    for (int i = 0; i < n; i++) {
        fprintf(f, "%f %f %f %f ", k[i].x, k[i].y, k[i].scale, k[i].orientation);
        for (int j = 0; j < 128; j++) fprintf(f, "%u ", k[i].descriptor[j]);
        fprintf(f, "\n");
    }
}

extern "C" void sift_write_to_file(const char* filename, const struct sift_keypoint_std* k, int n)
{
    FILE* f = fopen(filename, "w");
    if (!f) { fprintf(stderr, "Fatal error: Failed to open %s for writing\n\n", filename); exit(1); }
    fprintf_keypoint_std(f, k, n);
    fclose(f);
}

static struct sift_keypoint_std* read_std(const char* filename, int* n, int flag)
{
    sift_parameters p;
    sa_default_parameters(&p);
    struct sift_keypoints* keys = sift_malloc_keypoints();
    sift_read_keypoints(keys, filename, p.n_hist, p.n_ori, p.n_bins, flag);
    *n = keys->size;
    struct sift_keypoint_std* k = (struct sift_keypoint_std*)must_malloc((size_t)keys->size * sizeof *k);
    list_to_std(keys, k);
    sift_free_keypoints(keys);
    return k;
}

extern "C" struct sift_keypoint_std* sift_read_from_file(const char* filename, int* n)
{
    return read_std(filename, n, 1);
}

extern "C" struct sift_keypoint_std* sift_read_keyslocation_from_file(char* filename, int* n)
{
    return read_std(filename, n, 0);
}
