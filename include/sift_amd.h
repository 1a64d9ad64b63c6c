/*
 * sift_amd.h -- C ABI of libsift_amd.so, the MI355X (gfx950) implementation of the
 * `sift_anatomy` hot path.
 *
 * Part 1 declares the structs and entry points of the reference with identical names,
 * layouts, argument meaning and ownership rules, so existing C host code (sift_cli.c,
 * lib_sift.c callers, example.c) links against this library instead of
 * lib_sift_anatomy.o / lib_discrete.o / lib_description.o.  Each declaration cites the
 * reference interface it replaces (paths relative to sift_anatomy/src/).
 *
 * Part 2 is this library's own batched interface: a context that owns a pyramid pool in
 * HBM and processes N same-sized images per call, returning only the final keypoints.
 *
 * All pointers are plain host pointers unless a parameter is named d_* (device memory).
 * Every keypoint, descriptor, histogram, list, octave and image stack handed to the
 * caller through Part 1 is an individual libc malloc block, because callers release
 * them with sift_free_keypoints / sift_free_scalespace (lib_keypoint.c:167-182,
 * lib_scalespace.c:139-151).
 */
#ifndef SIFT_AMD_H
#define SIFT_AMD_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== Part 1 == */

/* lib_sift_anatomy.h:63-67 */
struct sift_parameters {
    int n_oct, n_spo, n_hist, n_bins, n_ori, itermax;
    float sigma_min, delta_min, sigma_in, C_DoG, C_edge, lambda_ori, t, lambda_descr;
};

/* lib_keypoint.h:76-96 */
struct keypoint {
    float x;      /* row coordinate, input-image pixels */
    float y;      /* column coordinate */
    float sigma;
    float theta;
    int o, s, i, j;
    float val;
    float edgeResp;
    int n_hist;
    int n_ori;
    float* descr;    /* n_hist*n_hist*n_ori floats holding 0..255 after quantisation */
    int n_bins;
    float* orihist;  /* n_bins floats */
};

/* lib_keypoint.h:106-110 */
struct sift_keypoints {
    int size;
    int capacity;
    struct keypoint** list;
};

/* lib_scalespace.h:62-74 */
struct octa {
    float delta;
    int w;
    int h;
    int nSca;
    float* sigmas;
    float* imStack;  /* nSca images of w*h samples, scale-major */
};
struct sift_scalespace {
    int nOct;
    struct octa* octaves[100];
};

/* lib_sift.h:73-80 */
struct sift_keypoint_std {
    float x, y, scale, orientation;
    unsigned char descriptor[128];
};

/* lib_sift_anatomy.h:69  (lib_sift_anatomy.c:771-789) */
struct sift_parameters* sift_assign_default_parameters(void);

/* lib_sift_anatomy.h:71-73  (lib_sift_anatomy.c:818-875).
 * ss[0..3] receive gss, dog, d/di, d/dj scale-spaces; kk[0..5] (pre-allocated by the caller
 * with sift_malloc_keypoints) receive the six intermediate lists. */
struct sift_keypoints* sift_anatomy(const float* x, int w, int h, const struct sift_parameters* p,
                                    struct sift_scalespace* ss[4], struct sift_keypoints* kk[6]);

/* lib_sift_anatomy.h:75-77  (lib_sift_anatomy.c:879-926) */
struct sift_keypoints* sift_anatomy_without_description(const float* x, int w, int h,
                                                        const struct sift_parameters* p,
                                                        struct sift_scalespace* ss[2],
                                                        struct sift_keypoints* kk[5]);

/* lib_sift_anatomy.h:79  (lib_sift_anatomy.c:931-949): descriptors for given oriented keypoints */
void sift_anatomy_only_description(const float* x, int w, int h, const struct sift_parameters* p,
                                   struct sift_keypoints* k);

/* lib_sift_anatomy.h:81  (lib_sift_anatomy.c:953-975): one orientation + descriptor per keypoint */
void sift_anatomy_orientation_and_description(const float* x, int w, int h,
                                              const struct sift_parameters* p,
                                              struct sift_keypoints* k);

/* lib_sift_anatomy.h:84-88  (lib_sift_anatomy.c:96-161): fills the image stacks of a
 * pre-allocated Gaussian scale-space */
void scalespace_compute(struct sift_scalespace* ss, const float* image, int im_w, int im_h,
                        float sigma_in);

/* lib_sift_anatomy.h:90-92  (lib_sift_anatomy.c:202-218) */
void scalespace_compute_gradient(const struct sift_scalespace* scalespace,
                                 struct sift_scalespace* sx, struct sift_scalespace* sy);

/* lib_sift.h:91-112  (lib_sift.c:140-285): flat API with default parameters */
struct sift_keypoint_std* sift_compute_points(const float* x, int w, int h, int* n);
struct sift_keypoint_std* sift_compute_features(const float* x, int w, int h, int* n);
void sift_fill_descriptors(const float* x, int w, int h, struct sift_keypoint_std* k, int n);
void sift_find_ori_and_fill_descriptors(const float* x, int w, int h, struct sift_keypoint_std* k, int n);

/* lib_sift.h:115-123  (lib_sift.c:287-367): keypoint text format */
void fprintf_keypoint_std(FILE* f, const struct sift_keypoint_std* k, int n);
void sift_write_to_file(const char* filename, const struct sift_keypoint_std* k, int n);
struct sift_keypoint_std* sift_read_from_file(const char* filename, int* n);
struct sift_keypoint_std* sift_read_keyslocation_from_file(char* filename, int* n);

/* Containers: lib_keypoint.h:98-147 (lib_keypoint.c:75-297), lib_scalespace.h:79-95
 * (lib_scalespace.c:122-297).  Exported so hosts need not compile the reference's
 * container files; a host that does keeps using its own (same malloc/free contract). */
struct keypoint* sift_malloc_keypoint(int n_ori, int n_hist, int n_bins);
struct keypoint* sift_malloc_keypoint_from_model_and_copy(const struct keypoint* key);
struct sift_keypoints* sift_malloc_keypoints(void);
void sift_add_keypoint_to_list(struct keypoint* key, struct sift_keypoints* keys);
void sift_free_keypoint(struct keypoint* key);
void sift_free_keypoints(struct sift_keypoints* keys);
void fprintf_one_keypoint(FILE* f, const struct keypoint* k, int n_descr, int n_bins, int flag);
void fprintf_keypoints(FILE* f, const struct sift_keypoints* keys, int flag);
void sift_save_keypoints(const struct sift_keypoints* keys, const char* name, int flag);
void sift_print_keypoints(const struct sift_keypoints* keys, int flag);
void sift_read_keypoints(struct sift_keypoints* keys, const char* name, int n_hist, int n_ori,
                         int n_bins, int flag);
struct sift_scalespace* sift_malloc_scalespace(int nOct, const float* deltas, const int* ws,
                                               const int* hs, const int* nScas, float** sigmas);
void sift_free_scalespace(struct sift_scalespace* sift_scalespace);
struct sift_scalespace* sift_malloc_scalespace_from_model(struct sift_scalespace* model);
struct sift_scalespace* sift_malloc_scalespace_dog_from_scalespace(struct sift_scalespace* scalespace);
struct sift_scalespace* sift_malloc_scalespace_lowe(int nOct, int nSca, int im_w, int im_h,
                                                    float delta_min, float sigma_min);

/* lib_matching.h:49-65 (lib_matching.c:92-172): for every keypoint of k1 the two nearest
 * descriptors of k2 (Euclidean); the pair is kept when the distance (flag 0) or the ratio of
 * the two distances (flag 1) is below thresh.  Appends copies to the three output lists.
 * k2 must hold at least two keypoints. */
void matching(struct sift_keypoints* k1, struct sift_keypoints* k2, struct sift_keypoints* out_k1,
              struct sift_keypoints* out_k2A, struct sift_keypoints* out_k2B, float thresh, int flag);
void print_pairs(const struct sift_keypoints* k1, const struct sift_keypoints* k2);
void save_pairs_extra(const char* name, const struct sift_keypoints* k1,
                      const struct sift_keypoints* k2A, const struct sift_keypoints* k2B);

/* ===================================================================== Part 2 == */
/* No counterpart in the reference: batched, context-based interface.  Functions return 0
 * on success and a negative value on failure; sift_amd_last_error() describes the failure
 * of the calling thread's last call.
 * Devices and streams: every entry point runs on the context's device and leaves the calling
 * thread's current HIP device as it found it.  A `stream` argument must belong to the context's
 * device.  The runs of one context share its pool, so they execute one after the other: a run
 * queued on another stream than the previous run first waits, on the device, for the previous
 * run.  The library does not touch a caller's stream after the call that received it has
 * returned (later runs and readers wait for an event of the context), so the caller may destroy
 * the stream at any time.  When a list of a run overflows its capacity, every call that reads
 * that run's results returns -2 until the next run. */

typedef struct sift_amd_ctx sift_amd_ctx;

#define SIFT_AMD_MAX_OCT 16
#define SIFT_AMD_MAX_SCA 12

enum {
    SIFT_AMD_KEEP_ALL_EXTREMA = 1, /* list every discrete extremum (needed to reproduce kk[0]);
                                      default keeps only those passing the first contrast test */
    SIFT_AMD_KEEP_SCALESPACE = 2   /* also store the DoG images; by default they never exist in
                                      memory (the scan and the refinement form DoG values from
                                      the Gaussian stack); required by
                                      sift_amd_copy_scalespace(which = 1) */
};

/* Kernel classes for the built-in event timers. */
enum {
    SIFT_AMD_K_OVERSAMPLE = 0, SIFT_AMD_K_SUBSAMPLE, SIFT_AMD_K_BLUR_V, SIFT_AMD_K_BLUR_H,
    SIFT_AMD_K_EXTREMA, SIFT_AMD_K_REFINE, SIFT_AMD_K_RANK, SIFT_AMD_K_ORIENTATION,
    SIFT_AMD_K_EXPAND, SIFT_AMD_K_DESCRIPTOR, SIFT_AMD_K_BLUR_FUSED, SIFT_AMD_K_COUNT
};

struct sift_amd_geometry {
    int n_oct, n_sca;                       /* n_sca Gaussian, n_sca-1 DoG images per octave */
    int w[SIFT_AMD_MAX_OCT], h[SIFT_AMD_MAX_OCT];
    float delta[SIFT_AMD_MAX_OCT];
    float sigmas[SIFT_AMD_MAX_OCT][SIFT_AMD_MAX_SCA];
    int blur_radius[SIFT_AMD_MAX_OCT][SIFT_AMD_MAX_SCA];  /* [o][0] is the seed blur (octave 0 only) */
    float thresh_pre, thresh, thresh_edge;
};

/* 40-byte keypoint record used for stage lists and final keys. */
struct sift_amd_key {
    float x, y, sigma, theta;
    int o, s, i, j;
    float val, edge;
};

const char* sift_amd_last_error(void);
int sift_amd_device_count(void);

/* Allocates the pyramid pool for up to max_batch images of w x h on `device`.
 * p == NULL selects the default parameters.  Returns NULL on failure. */
sift_amd_ctx* sift_amd_create(int device, int w, int h, int max_batch,
                              const struct sift_parameters* p, int flags);
void sift_amd_destroy(sift_amd_ctx* ctx);
int sift_amd_get_geometry(const sift_amd_ctx* ctx, struct sift_amd_geometry* g);

/* Runs the whole pipeline on n <= max_batch images already resident in device memory
 * (float32, row-major, [n][h][w]).  Asynchronous: work is queued on `stream` (a hipStream_t
 * passed as void*; NULL = the context's own stream).  d_images must stay valid until the
 * next synchronising call. */
int sift_amd_detect_device(sift_amd_ctx* ctx, const float* d_images, int n, void* stream);

/* Same for host images: copies them to the device first. */
int sift_amd_detect_host(sift_amd_ctx* ctx, const float* images, int n);

/* 8-bit input: the conversion the reference's loader applies before sift_anatomy
 * (io_png.c:480-520 gray = 0.2126 R + 0.7152 G + 0.0722 B evaluated in double and rounded to
 * float, alpha dropped; sift_cli.c:280-281 division by 256) runs on the device, then the
 * pipeline as above.  channels: 1 gray, 3 RGB, 4 RGBA, interleaved [n][h][w][channels]. */
int sift_amd_detect_device_u8(sift_amd_ctx* ctx, const unsigned char* d_images, int n, int channels,
                              void* stream);
int sift_amd_detect_host_u8(sift_amd_ctx* ctx, const unsigned char* images, int n, int channels);
/* The converted float images of the last *_u8 call ([n][h][w]), for inspection. */
int sift_amd_copy_ingested(sift_amd_ctx* ctx, int n, float* dst);

/* Scale-space only (oversample, blurs, DoG) on device-resident images. */
int sift_amd_scalespace_device(sift_amd_ctx* ctx, const float* d_images, int n, void* stream);

/* Waits for the last run, then reports per image the number of final keypoints. */
int sift_amd_keypoint_counts(sift_amd_ctx* ctx, int* counts /* [n] */);

/* Copies the final keypoints of image `img` (call sift_amd_keypoint_counts first).
 * Either output may be NULL.  Returns the number of keypoints copied. */
int sift_amd_fetch_keypoints(sift_amd_ctx* ctx, int img, struct sift_keypoint_std* out_std,
                             struct sift_amd_key* out_keys, int max_keys);

/* Host gather: all keypoints of the last run, image after image, into one malloc'd array
 * (caller frees with free()); counts[i] receives the per-image counts. */
int sift_amd_gather_keypoints(sift_amd_ctx* ctx, struct sift_keypoint_std** out, int* counts);

/* Same into a caller-owned buffer of max_records records (pinned host memory makes the copy
 * asynchronous-capable and fastest); nothing is allocated.  Returns the number of records. */
int sift_amd_gather_keypoints_into(sift_amd_ctx* ctx, struct sift_keypoint_std* out, int max_records,
                                   int* counts);

/* Stage-level access for parity tooling.  stage_sizes[0..5] are the sizes of the lists
 * A..F of lib_sift_anatomy.c:843-848, extrema_per_octave the discrete-extrema counts.
 * Without SIFT_AMD_KEEP_ALL_EXTREMA the scan skips regions in which no sample passes the
 * first contrast test, so extrema below that threshold are not counted: stage_sizes[0] and
 * the per-octave counts are then reported as -1 (lists B..F are unaffected). */
int sift_amd_stage_counts(sift_amd_ctx* ctx, int img, int stage_sizes[6],
                          int extrema_per_octave[SIFT_AMD_MAX_OCT]);
/* Candidate records in canonical (o,s,i,j)-of-seed order with the last stage each reached
 * (0=A .. 5=F).  Returns the number written; with all three outputs NULL, the number
 * available.  Without SIFT_AMD_KEEP_ALL_EXTREMA the candidates are list B. */
int sift_amd_fetch_candidates(sift_amd_ctx* ctx, int img, struct sift_amd_key* seed,
                              struct sift_amd_key* refined, int* stage, int max_n);
/* which: 0 = Gaussian stack (n_sca images), 1 = DoG stack (n_sca-1 images) of one octave. */
int sift_amd_copy_scalespace(sift_amd_ctx* ctx, int img, int which, int octave, float* dst);
/* Per final keypoint: smoothed orientation histogram (n_bins floats), float descriptor
 * (n_hist*n_hist*n_ori floats holding 0..255) and the rank, within list F, of the keypoint
 * it was derived from.  Any output may be NULL.  Returns the number of keypoints. */
int sift_amd_fetch_extras(sift_amd_ctx* ctx, int img, float* orihist, float* descr, int* src_index,
                          int max_keys);
/* Smoothed orientation histograms of the list-F keypoints, in list order. */
int sift_amd_fetch_stage_f_histograms(sift_amd_ctx* ctx, int img, float* orihist, int max_n);
/* Matching as lib_matching.c:92-135 on row-major host descriptors (n x dim, dim <= 256).
 * For every query i of desc1: iA/iB = indices of its nearest and second nearest row of desc2,
 * dA/dB their distances, pass[i] = 1 when the pair satisfies the threshold (flag as in
 * `matching`).  All outputs have n1 entries; returns the number of passing queries, or a
 * negative value.  The u8 form computes on the matrix cores and equals the float form on
 * the same values. */
int sift_amd_match_u8(int device, const unsigned char* desc1, int n1, const unsigned char* desc2,
                      int n2, int dim, float thresh, int flag, int* iA, int* iB, float* dA,
                      float* dB, unsigned char* pass);
int sift_amd_match_f32(int device, const float* desc1, int n1, const float* desc2, int n2, int dim,
                       float thresh, int flag, int* iA, int* iB, float* dA, float* dB,
                       unsigned char* pass);
/* Same between the keypoints of two images of the context's last run, without leaving the
 * device; outputs need room for the keypoint count of img_a (cap entries given). */
int sift_amd_match_images(sift_amd_ctx* ctx, int img_a, int img_b, float thresh, int flag, int* iA,
                          int* iB, float* dA, float* dB, unsigned char* pass, int cap);
/* No counterpart in the reference.  sift_amd_match_u8 for callers that cannot allocate: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The call
 * allocates nothing (no hipMalloc/hipFree, no host allocation): descriptors (n x dim bytes,
 * row-major) and the five result arrays (n1 entries each, meaning as in sift_amd_match_u8) are
 * device memory owned by the caller, and the kernels' scratch (packed rows, norms, partial
 * minima) is a fixed 5.3 MB array inside the library's code object, one copy per GPU.
 * Limits: n1, n2 <= 8192, dim <= 256, n2 >= 2; anything larger is refused with an error.
 * Runs on the calling thread's current device; work is queued on `stream` (hipStream_t as
 * void*) and the results are valid once the stream has caught up.  NULL means the legacy null
 * stream, the only place where this library touches it (creating a stream would be an
 * allocation): the null stream synchronises with every blocking stream of the process and a
 * launch on it fails while another host thread captures a hipGraph, so callers with more than
 * one thread should pass a stream created with hipStreamNonBlocking.
 * Returns 0 when queued, a negative value on error.  Every call on one GPU uses that GPU's one
 * scratch array: calls that overlap on a GPU (different streams without ordering) race; calls
 * on one stream are ordered.  The first call on a device looks up the array's device address
 * (hipGetSymbolAddress); make that call outside a graph capture. */
int sift_amd_match_u8_noalloc(const unsigned char* d_desc1, int n1, const unsigned char* d_desc2,
                              int n2, int dim, float thresh, int flag, int* d_iA, int* d_iB,
                              float* d_dA, float* d_dB, unsigned char* d_pass, void* stream);
/* Frees the context that the reference-named entry points keep per host thread. */
void sift_amd_release_thread_cache(void);

/* Measurement aid: the rate (read + written bytes per second, in GB/s) that a plain copy kernel
 * reaches on `device` between two freshly allocated buffers of `bytes` bytes, averaged over
 * `reps` launches.  A practical ceiling to put beside the nominal HBM peak. */
int sift_amd_copy_ceiling(int device, size_t bytes, int reps, double* gb_per_s);
/* Likewise for arithmetic: the rate, in 1e9 lane-operations per second, at which `device`
 * executes float32 multiplies and adds that are not fused (each rounded on its own, as the
 * blur of lib_discrete.c:286-291 prescribes), from registers, with nothing else in the way. */
int sift_amd_valu_ceiling(int device, int reps, double* lane_gops);

/* Event timers: enable for a mask of (1 << SIFT_AMD_K_*) classes; read accumulates since the
 * last reset.  ms = summed kernel durations, launches = number of launches timed. */
int sift_amd_profile_enable(sift_amd_ctx* ctx, unsigned mask);
int sift_amd_profile_read(sift_amd_ctx* ctx, int kernel_class, int octave /* -1 = all */,
                          double* ms, long long* launches, int reset);
/* The blur launches (classes SIFT_AMD_K_BLUR_FUSED / SIFT_AMD_K_OVERSAMPLE) one by one: the launch
 * that writes image `scale` of the Gaussian stack of `octave` (scale 0 = the seed launch). */
int sift_amd_profile_read_blur(sift_amd_ctx* ctx, int octave, int scale, double* ms, long long* launches,
                               int reset);

/* Test aid: how many blur launches of this process took which route since the last reset
 * (a hipGraph replay launches nothing and counts nothing).  out has SIFT_AMD_LS_COUNT entries. */
enum {
    SIFT_AMD_LS_PAIR_SEED = 0,          /* two-blur launches: oversampled seed + first blur */
    SIFT_AMD_LS_PAIR,                   /* two-blur launches of two stack images */
    SIFT_AMD_LS_PAIR_STRIP_MAX,         /* tallest strip (rows) of any two-blur launch */
    SIFT_AMD_LS_WIDE_AHEAD2,            /* single-blur launches, 480-column tiles, rows fetched two iterations ahead */
    SIFT_AMD_LS_WIDE_AHEAD2_STRIP_MAX,  /* their tallest strip */
    SIFT_AMD_LS_WIDE_OTHER,             /* other 480-column-tile launches (seed blur, short strips) */
    SIFT_AMD_LS_NARROW,                 /* 224-column-tile launches */
    SIFT_AMD_LS_SPARSE,                 /* launches of the last stack image that skipped unread regions */
    SIFT_AMD_LS_COUNT
};
int sift_amd_launch_stats(long long* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* SIFT_AMD_H */
