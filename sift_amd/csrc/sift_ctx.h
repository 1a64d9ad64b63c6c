// Internal interface between the pipeline (sift_ctx.hip) and the reference-named C API
// (sift_compat.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sift_amd.h"
#include "sift_dev.h"

struct SaLayout {
    int n_oct, n_sca;
    int w[SA_MAX_OCT], h[SA_MAX_OCT];
    float delta[SA_MAX_OCT];
    float sigmas[SA_MAX_OCT][SA_MAX_SCA];
};

void sa_set_error(const char* fmt, ...);
void sa_default_parameters(sift_parameters* p);
int sa_number_of_octaves(int w, int h, const sift_parameters* p);
float sa_convert_threshold(const sift_parameters* p);
void sa_layout_lowe(SaLayout* L, int n_oct, int n_spo, int im_w, int im_h, float delta_min, float sigma_min);

sift_amd_ctx* sa_create_from_layout(int device, int im_w, int im_h, int max_batch, const sift_parameters* p,
                                    const SaLayout* L, float sigma_in, int flags, int cap_req, int fcap_req);
// whole_stack: every image of every stack is computed everywhere.  Without it a context that keeps
// neither the DoG stacks nor every extremum computes the last image of an octave only where the
// extrema scan and the refinement of the same run can read it.
int sa_run_scalespace(sift_amd_ctx* c, const float* d_images, int n, hipStream_t st, bool whole_stack = true);
int sa_run_detect(sift_amd_ctx* c, int n, hipStream_t st);
int sa_run_orient(sift_amd_ctx* c, int n, hipStream_t st, bool given, int mode);
int sa_run_describe(sift_amd_ctx* c, int n, hipStream_t st);
int sa_sync_counters(sift_amd_ctx* c);
int sa_describe_given(sift_amd_ctx* c, const SaKey* keys, int n, bool find_orientation);
int sa_ctx_limits(const sift_amd_ctx* c, int* cap, int* fcap, int* kcap);
int sa_ctx_candidates_seen(const sift_amd_ctx* c);
const SaLayout* sa_ctx_layout(const sift_amd_ctx* c);
const sift_parameters* sa_ctx_params(const sift_amd_ctx* c);
float* sa_ctx_input_buffer(sift_amd_ctx* c);
// device records of the final keypoints of one image of the last run (waits for it)
int sa_ctx_image_keys(sift_amd_ctx* c, int img, const void** d_std, const float** d_descr, int* n, int* n_descr);
hipStream_t sa_ctx_stream(sift_amd_ctx* c);
// device address of image 0's Gaussian (which = 0) or DoG (1) stack of octave o; waits for the last run
const float* sa_ctx_stack(sift_amd_ctx* c, int which, int o);
