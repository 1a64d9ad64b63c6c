// Internal definitions shared by the HIP kernels and the host-side pipeline.
// Nothing in this header is part of the public C ABI (see include/sift_amd.h).
#pragma once
#include <stdint.h>

#define SA_MAX_OCT 16
#define SA_MAX_SCA 12          // Gaussian images per octave (n_spo + 3)  => n_spo <= 9
#define SA_MAX_RADIUS_FAST 16  // blur radii up to this use the register-window kernels
#define SA_MAX_TAPS 512        // generic blur path: radius < SA_MAX_TAPS
#define SA_MAX_BINS 64         // orientation histogram bins (one lane per bin)
#define SA_MAX_DESCR 256       // n_hist*n_hist*n_ori (one thread per bin)
#define SA_MAX_PEAKS 32        // principal orientations kept per keypoint (>= n_bins/2)

// Per-image counters, one block of SA_CNT_STRIDE ints per image.
enum {
    SA_CNT_CAND = 0,      // candidates appended by the extrema scan
    SA_CNT_OVERFLOW = 1,  // non-zero if any list ran out of room
    SA_CNT_NF = 2,        // keypoints that passed every filter (stage F)
    SA_CNT_KEYS = 3,      // oriented keypoints (final list length)
    SA_CNT_STAGE = 4,     // [4..9]   stage sizes A..F (A = all discrete extrema)
    SA_CNT_EXT_OCT = 10,  // [10..25] discrete extrema per octave
    SA_CNT_SURV = 26,     // survivors of all filters appended (unordered) by the refinement kernel
    SA_CNT_STRIDE = 32
};

// Stage reached by a candidate (cumulative): the reference's kk[0..5] lists.
enum { SA_ST_A = 0, SA_ST_B = 1, SA_ST_C = 2, SA_ST_D = 3, SA_ST_E = 4, SA_ST_F = 5 };

// 40-byte keypoint record used on the device and for transfers.
struct SaKey {
    float x, y, sigma, theta;  // input-image units; x = row axis, y = column axis
    int o, s, i, j;            // octave, scale index and sample coordinates
    float val, edge;           // DoG value and edge response
};

// 144-byte output record, identical to struct sift_keypoint_std of lib_sift.h.
struct SaStd {
    float x, y, scale, orientation;
    unsigned char descriptor[128];
};

struct SaOctave {
    float* gss;              // [batch][n_sca][h][w]
    float* dog;              // [batch][n_sca-1][h][w]
    long long gss_stride;    // floats between consecutive images of the batch
    long long dog_stride;
    int w, h;
    float delta;
    int pad_;
    float sigmas[SA_MAX_SCA];
};

struct SaPyramid {
    int n_oct, n_sca;        // n_sca Gaussian images, n_sca-1 DoG images per octave
    int im_w, im_h;          // input image size
    SaOctave oct[SA_MAX_OCT];
};

struct SaTaps { float g[SA_MAX_RADIUS_FAST + 1]; };

// Scalar parameters of the detection / description stages (all already converted the
// way the reference converts them on the host).
struct SaScalars {
    float thr_pre;      // 0.8 * thresh, rounded to float
    float thr;          // contrast threshold
    float thr_edge;     // (C_edge+1)^2 / C_edge
    float sigma_ratio;  // sigmas[0][1] / sigmas[0][0]
    float lambda_ori, t_peak, lambda_descr;
    int itermax, n_bins, n_hist, n_ori;
    int debug;               // diagnostic switches (SIFT_AMD_DEBUG): 1 = generic descriptor list walk
};

#if defined(__HIPCC__)
#define SA_HD __host__ __device__
#else
#define SA_HD
#endif
SA_HD static inline unsigned long long sa_pack_key(int o, int s, int i, int j)
{
    return ((unsigned long long)o << 44) | ((unsigned long long)s << 40) |
           ((unsigned long long)i << 20) | (unsigned long long)j;
}
