// Host-callable launch wrappers of the kernels in sift_kernels.hip (internal).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sift_amd.h"
#include "sift_dev.h"

struct sift_amd_ctx;

namespace sa {

struct BlurArgs {
    const float* in;     // source images            [batch] stride s_in
    float* tmp;          // column-pass result       [batch] stride s_tmp
    float* out;          // blurred images           [batch] stride s_out
    float* dog;          // optional: out - in       [batch] stride s_dog (nullptr = skip)
    float* sub;          // optional: even samples of out (next octave), fused kernel only
    long long s_in, s_tmp, s_out, s_dog, s_sub;
    int w, h, batch;
    int radius;
    SaTaps taps;             // radius <= SA_MAX_RADIUS_FAST
    const float* taps_dev;   // radius+1 taps in device memory (any radius)
    // fused kernel only: if ovs_w > 0, `in` is the ovs_w x ovs_h input image and the source of
    // the blur is its bilinear oversampling at step ovs_delta (w x h = oversampled size)
    int ovs_w = 0, ovs_h = 0;
    float ovs_delta = 0.f;
    int* flags = nullptr;    // fused kernel only: marks of 8x64 cells holding a |DoG| sample > flag_thr
    float flag_thr = 0.f;
    // fused kernel, plain launches only: marks of the cells whose neighbourhood is the only part of
    // `out` that will be read; the rest of `out` is then left as it is (nullptr = whole image)
    const int* need_cells = nullptr;
};

// two successive blurs of one octave in one launch (in -> out_a -> out_b)
struct BlurPairArgs {
    const float* in;
    float *out_a, *out_b;
    long long s_in, s_a, s_b;
    int w, h, batch;
    int radius_a, radius_b;
    SaTaps taps_a, taps_b;
    int* flags = nullptr;        // contrast marks from the DoG of out_a (flag_on_a) and/or of out_b (flag_on_b)
    bool flag_on_a = false, flag_on_b = false;
    float flag_thr = 0.f;
    float* sub = nullptr;        // optional: even samples of out_b (the next octave's first image)
    long long s_sub = 0;
    // if ovs_w > 0, `in` is the ovs_w x ovs_h input image and the first blur's source is its bilinear
    // oversampling at step ovs_delta (w x h = oversampled size)
    int ovs_w = 0, ovs_h = 0;
    float ovs_delta = 0.f;
};
bool launch_blur_pair(hipStream_t st, const BlurPairArgs& a, bool probe = false);
void launch_blur_pass(hipStream_t st, const BlurArgs& a, int pass);
bool launch_blur_fused(hipStream_t st, const BlurArgs& a);
void launch_clear_words(hipStream_t st, void* p, long long n_words);   // n_words rounded up to a multiple of 4
void launch_oversample(hipStream_t st, const float* in, int wi, int hi, long long s_in,
                       float* out, int wo, int ho, long long s_out, float delta, int batch);
void launch_ingest_u8(hipStream_t st, const unsigned char* in, int channels, float* out, long long n_px);
void launch_subsample(hipStream_t st, const float* in, int wi, long long s_in,
                      float* out, int wo, int ho, long long s_out, int batch);
void launch_sub(hipStream_t st, const float* a, const float* b, float* out, long long n,
                long long s_a, long long s_b, long long s_out, int batch);
void launch_gradient(hipStream_t st, const float* im, float* gi, float* gj, int w, int h, int n_img);
int launch_extrema(hipStream_t st, const float* gss, long long s_gss, int w, int h, int n_dog, int o,
                   float thr_pre, int emit_all, unsigned long long* ckey, float* cval,
                   int* counters, int cap, int batch, const int* flags);
void launch_refine(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const unsigned long long* ckey,
                   const float* cval, int* counters, int cap, SaKey* ref, int* stage,
                   unsigned long long* skey, int* sidx, int fcap, int batch);
void launch_rank(hipStream_t st, const unsigned long long* skey, const int* sidx, int* counters,
                 int* flist, int fcap, int batch);
void launch_orientation(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const SaKey* keys,
                        const int* index, const int* counters, int cnt_slot, int kcap, int icap, int mode,
                        int* ori_n, float* ori_theta, float* ori_hist, int batch);
void launch_expand(hipStream_t st, const SaKey* keys, const int* index, int* counters, int cnt_slot,
                   int kcap, int icap, const int* ori_n, const float* ori_theta, SaKey* out,
                   int* out_src, int ocap, int batch);
void launch_descriptor(hipStream_t st, const SaPyramid& P, const SaScalars& sc, const SaKey* keys,
                       const int* counters, int cnt_slot, int kcap, SaStd* out_std, float* out_descr,
                       int batch);

void launch_stats(long long* out, int reset);   // SIFT_AMD_LS_* counters of this process

void launch_pack_std(hipStream_t st, const SaStd* std_recs, const int* counters, int cnt_slot, int kcap,
                     SaStd* packed, int batch);

}  // namespace sa
