/* Wall time of the reference-named calls as a C host sees them: sift_compute_features()
 * (lib_sift.c:140-184: float image in host memory -> malloc'd sift_keypoint_std[]) and the full
 * sift_anatomy() (lib_sift_anatomy.c:818-875, called as sift_cli.c:285-364 does: four scale-spaces
 * and six lists handed back as malloc'd structs and freed by the caller).  Prints one JSON object.
 *   timing_example <w> <h> <calls of sift_compute_features> <calls of sift_anatomy> */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "sift_amd.h"

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

/* smooth image: bilinear interpolation of a coarse lattice of LCG values plus a finer one */
static void fill(float* x, int w, int h)
{
    unsigned st = 12345u;
    int lw = w / 64 + 2, lh = h / 64 + 2, fw = w / 16 + 2, fh = h / 16 + 2;
    float* a = malloc(sizeof(float) * lw * lh);
    float* b = malloc(sizeof(float) * fw * fh);
    for (int i = 0; i < lw * lh; i++) { st = st * 1664525u + 1013904223u; a[i] = (st >> 8) / 16777216.0f; }
    for (int i = 0; i < fw * fh; i++) { st = st * 1664525u + 1013904223u; b[i] = (st >> 8) / 16777216.0f; }
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float fi = i / 64.0f, fj = j / 64.0f, gi = i / 16.0f, gj = j / 16.0f;
            int i0 = (int)fi, j0 = (int)fj, k0 = (int)gi, l0 = (int)gj;
            float ti = fi - i0, tj = fj - j0, si = gi - k0, sj = gj - l0;
            float c = (1 - ti) * ((1 - tj) * a[i0 * lw + j0] + tj * a[i0 * lw + j0 + 1])
                    + ti * ((1 - tj) * a[(i0 + 1) * lw + j0] + tj * a[(i0 + 1) * lw + j0 + 1]);
            float f = (1 - si) * ((1 - sj) * b[k0 * fw + l0] + sj * b[k0 * fw + l0 + 1])
                    + si * ((1 - sj) * b[(k0 + 1) * fw + l0] + sj * b[(k0 + 1) * fw + l0 + 1]);
            x[(size_t)i * w + j] = 0.7f * c + 0.25f * f;
        }
    free(a); free(b);
}

int main(int argc, char** argv)
{
    int w = argc > 1 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1080;
    int n_flat = argc > 3 ? atoi(argv[3]) : 12, n_full = argc > 4 ? atoi(argv[4]) : 4;
    if (n_flat > 64) n_flat = 64;
    if (n_full > 64) n_full = 64;
    float* x = malloc((size_t)w * h * sizeof(float));
    fill(x, w, h);
    double t_flat[64], t_full[64];
    int n = 0, n_keys_full = 0;
    for (int r = 0; r < n_flat; r++) {
        double t0 = now_ms();
        struct sift_keypoint_std* k = sift_compute_features(x, w, h, &n);
        t_flat[r] = now_ms() - t0;
        free(k);
    }
    struct sift_parameters* p = sift_assign_default_parameters();
    for (int r = 0; r < n_full; r++) {
        struct sift_keypoints* kk[6];
        struct sift_scalespace* ss[4];
        double t0 = now_ms();
        for (int i = 0; i < 6; i++) kk[i] = sift_malloc_keypoints();
        struct sift_keypoints* keys = sift_anatomy(x, w, h, p, ss, kk);
        n_keys_full = keys->size;
        sift_free_keypoints(keys);
        for (int i = 0; i < 6; i++) sift_free_keypoints(kk[i]);
        for (int i = 0; i < 4; i++) sift_free_scalespace(ss[i]);
        t_full[r] = now_ms() - t0;
    }
    double first_flat = n_flat ? t_flat[0] : 0, first_full = n_full ? t_full[0] : 0;
    /* the first call of each kind builds the thread's context (HBM pool): reported apart */
    if (n_flat > 1) qsort(t_flat + 1, n_flat - 1, sizeof(double), cmp);
    if (n_full > 1) qsort(t_full + 1, n_full - 1, sizeof(double), cmp);
    printf("{\"image\": \"%dx%d\", \"keypoints\": %d, \"sift_compute_features_ms\": {\"first_call\": %.3f, \"median_later\": %.3f, \"calls\": %d}, "
           "\"sift_anatomy_ms\": {\"first_call\": %.3f, \"median_later\": %.3f, \"calls\": %d, \"keypoints\": %d}}\n",
           w, h, n, first_flat, n_flat > 1 ? t_flat[1 + (n_flat - 1) / 2] : first_flat, n_flat,
           first_full, n_full > 1 ? t_full[1 + (n_full - 1) / 2] : first_full, n_full, n_keys_full);
    free(p); free(x);
    return 0;
}
