/*
 * sift_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the `sift_anatomy()` hot path of the
 * reference (IPOL "Anatomy of the SIFT Method", as vendored in NitishNaineni/sift under
 * sift_anatomy/src).  It exists only so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check / time the HIP path on machines where the
 * reference sources are not present.  The shipped library never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py compares every stage of this
 * file bit-for-bit with the reference sources compiled in place (oracle/_ref, built by
 * oracle/Makefile from /root/reference), and tests/golden/ holds digests produced by that
 * compiled reference, so the same check runs where the reference is absent.
 *
 * Design differences from the reference (results are bit-identical):
 *   - flat arrays instead of per-keypoint heap objects; one result struct;
 *   - the blur precomputes its mirrored index table once per line length and runs the
 *     tap loop outermost, so each output still sees   sum = c*g0; sum += g[k]*(l+r), k=1..r
 *     in that order (float, no contraction) but the inner loop is a contiguous stream;
 *   - gradients are evaluated on the fly from the Gaussian images with the border rule of
 *     the reference's gradient routine instead of being materialised first.
 *
 * Every function cites the reference lines it restates as  [file:lines].
 * Build with the reference's flags (C99 => -ffp-contract=off, -O3, baseline x86-64).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* strict C99 hides the M_* macros; same literals as [lib_util.h:7-17] */
#define OS_PI    3.14159265358979323846
#define OS_SQRT2 1.41421356237309504880
#define OS_LN2   0.693147180559945309417

#define OS_MAX_OCT 32
#define OS_MAX_SCA 16

typedef struct {                    /* same layout as struct sift_parameters */
    int n_oct, n_spo, n_hist, n_bins, n_ori, itermax;     /* [lib_sift_anatomy.h:63-67] */
    float sigma_min, delta_min, sigma_in, C_DoG, C_edge, lambda_ori, t, lambda_descr;
} os_params;

typedef struct {
    int o, s, i, j;
    float x, y, sigma, theta, val, edge;
} os_point;

typedef struct {
    int n_oct, n_sca;                         /* n_sca = n_spo + 3 Gaussian images per octave */
    int w[OS_MAX_OCT], h[OS_MAX_OCT];
    float delta[OS_MAX_OCT];
    float sigmas[OS_MAX_OCT][OS_MAX_SCA];
    float thresh, thresh_pre, thresh_edge;
    float *gss[OS_MAX_OCT];                   /* n_sca   * h * w */
    float *dog[OS_MAX_OCT];                   /* n_sca-1 * h * w */
    int n_stage[6];
    os_point *stage[6];                       /* A..F */
    int n_keys, n_descr, n_bins;
    os_point *keys;
    float *descr;                             /* n_keys * n_descr, values 0..255 */
    float *orihist;                           /* n_keys * n_bins (smoothed) */
} os_result;

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a < b ? b : a; }

/* ---------------------------------------------------------------- scalars -- */

/* [lib_sift_anatomy.c:793-802] octave count; note the integer division h0/12. */
int os_number_of_octaves(int w, int h, const os_params *p)
{
    int hmin = 12;
    int h0 = (int)((float)imin(w, h) / p->delta_min);
    int n = (int)(log((double)(h0 / hmin)) / OS_LN2) + 1;
    return imin(p->n_oct, n);
}

/* [lib_sift_anatomy.c:807-814] */
float os_convert_threshold(const os_params *p)
{
    float k_nspo = (float)exp(OS_LN2 / (double)((float)p->n_spo));
    float k_3 = (float)exp(OS_LN2 / (double)((float)3));
    return (k_nspo - 1) / (k_3 - 1) * p->C_DoG;
}

/* Geometry of Lowe's layout. [lib_scalespace.c:216-252] */
static void os_layout(os_result *r, int n_oct, int n_spo, int im_w, int im_h,
                      float delta_min, float sigma_min)
{
    r->n_oct = n_oct;
    r->n_sca = n_spo + 3;
    r->delta[0] = delta_min;
    r->h[0] = (int)((float)im_h / delta_min);
    r->w[0] = (int)((float)im_w / delta_min);
    for (int o = 1; o < n_oct; o++) {
        r->w[o] = r->w[o - 1] / 2;
        r->h[o] = r->h[o - 1] / 2;
        r->delta[o] = (float)(r->delta[o - 1] * 2.0);
    }
    for (int o = 0; o < n_oct; o++)
        for (int s = 0; s < n_spo + 3; s++)
            r->sigmas[o][s] = (float)((double)(r->delta[o] / r->delta[0] * sigma_min)
                                      * pow(2.0, (double)((float)s / (float)n_spo)));
}

/* ---------------------------------------------------------- image kernels -- */

/* Taps of the half kernel g[0..rad]. [lib_discrete.c:145-164, 172-174]
 * Returns the radius (int)ceil(4*sigma); taps must hold rad+1 floats. */
int os_gaussian_taps(float sigma, float *taps, int cap)
{
    int rad = (int)ceil((double)(4 * sigma));
    if (rad + 1 > cap) return -1;
    taps[0] = 1.f;
    if (sigma > 0) {
        float sum = taps[0];
        for (int i = 1; i <= rad; i++) {
            taps[i] = (float)exp(-0.5 * (double)(float)i * (double)(float)i
                                 / (double)sigma / (double)sigma);
            sum += 2 * taps[i];
        }
        for (int i = 0; i <= rad; i++) taps[i] /= sum;
    } else {
        for (int i = 1; i <= rad; i++) taps[i] = 0.f;
    }
    return rad;
}

/* Half-sample symmetric, period 2l. [lib_discrete.c:243-249]  (valid for i >= -2l) */
static int os_mirror(int i, int l)
{
    int ll = 2 * l;
    i = (i + ll) % ll;
    if (i > l - 1) i = ll - 1 - i;
    return i;
}

/* Separable symmetric blur: pass 1 along i (rows index), pass 2 along j.
 * [lib_discrete.c:172-177, 278-308] */
void os_blur(const float *in, float *out, int w, int h, float sigma)
{
    float taps[256];
    int r = os_gaussian_taps(sigma, taps, 256);
    float *tmp = malloc((size_t)w * h * sizeof(float));
    int len = imax(w, h);
    int *idx = malloc((size_t)(len + 2 * r) * sizeof(int));

    /* pass 1: tmp[i][j] = in[i][j]*g0 + sum_k g[k]*(in[m(i-k)][j] + in[m(i+k)][j]) */
    for (int t = 0; t < h + 2 * r; t++) idx[t] = os_mirror(t - r, h);
    for (int i = 0; i < h; i++) {
        float *acc = tmp + (size_t)i * w;
        const float *c = in + (size_t)i * w;
        for (int j = 0; j < w; j++) acc[j] = c[j] * taps[0];
        for (int k = 1; k <= r; k++) {
            const float *a = in + (size_t)idx[i - k + r] * w;
            const float *b = in + (size_t)idx[i + k + r] * w;
            float g = taps[k];
            for (int j = 0; j < w; j++) acc[j] += g * (a[j] + b[j]);
        }
    }
    /* pass 2 along j, on tmp */
    for (int t = 0; t < w + 2 * r; t++) idx[t] = os_mirror(t - r, w);
    for (int i = 0; i < h; i++) {
        const float *row = tmp + (size_t)i * w;
        float *o = out + (size_t)i * w;
        for (int j = 0; j < w; j++) {
            float sum = row[j] * taps[0];
            const int *ix = idx + j + r;
            for (int k = 1; k <= r; k++)
                sum += taps[k] * (row[ix[-k]] + row[ix[k]]);
            o[j] = sum;
        }
    }
    free(idx);
    free(tmp);
}

/* Bilinear resampling at step delta_out <= 1. [lib_discrete.c:207-237] */
void os_oversample(const float *in, int wi, int hi, float *out, int wo, int ho, float delta_out)
{
    for (int i = 0; i < ho; i++) {
        float x = i * delta_out;
        int im = (int)x, ip = im + 1;
        if (ip >= hi) ip = 2 * hi - 1 - ip;
        if (im >= hi) im = 2 * hi - 1 - im;
        const float fx = (float)((double)x - floor((double)x));
        for (int j = 0; j < wo; j++) {
            float y = j * delta_out;
            int jm = (int)y, jp = jm + 1;
            if (jp >= wi) jp = 2 * wi - 1 - jp;
            if (jm >= wi) jm = 2 * wi - 1 - jm;
            const float fy = (float)((double)y - floor((double)y));
            out[(size_t)i * wo + j] =
                fx * (fy * in[(size_t)ip * wi + jp] + (1 - fy) * in[(size_t)ip * wi + jm])
              + (1 - fx) * (fy * in[(size_t)im * wi + jp] + (1 - fy) * in[(size_t)im * wi + jm]);
        }
    }
}

/* Keep even samples. [lib_discrete.c:180-191] */
void os_subsample2(const float *in, float *out, int wi, int hi)
{
    int wo = wi / 2, ho = hi / 2;
    for (int i = 0; i < ho; i++)
        for (int j = 0; j < wo; j++)
            out[(size_t)i * wo + j] = in[(size_t)(2 * i) * wi + 2 * j];
}

/* Gradient along i ("x", down) and j ("y", right) at one pixel: centred half difference
 * inside, plain one-sided difference on the first/last row or column.
 * [lib_discrete.c:87-130] */
static inline float os_grad_i(const float *im, int w, int h, int i, int j)
{
    if (i == 0) return im[(size_t)w + j] - im[j];
    if (i == h - 1) return im[(size_t)(h - 1) * w + j] - im[(size_t)(h - 2) * w + j];
    return (float)((double)(im[(size_t)(i + 1) * w + j] - im[(size_t)(i - 1) * w + j]) * 0.5);
}
static inline float os_grad_j(const float *im, int w, int h, int i, int j)
{
    (void)h;
    if (j == 0) return im[(size_t)i * w + 1] - im[(size_t)i * w];
    if (j == w - 1) return im[(size_t)i * w + w - 1] - im[(size_t)i * w + w - 2];
    return (float)((double)(im[(size_t)i * w + j + 1] - im[(size_t)i * w + j - 1]) * 0.5);
}

/* Materialised gradient images (only for comparing with the reference's ss[2], ss[3]). */
void os_gradient(const float *im, float *gi, float *gj, int w, int h)
{
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            gi[(size_t)i * w + j] = os_grad_i(im, w, h, i, j);
            gj[(size_t)i * w + j] = os_grad_j(im, w, h, i, j);
        }
}

/* ------------------------------------------------------------ scale-space -- */

/* [lib_sift_anatomy.c:96-161] + DoG [lib_sift_anatomy.c:169-186] */
static void os_build_scalespace(os_result *r, const float *image, int im_w, int im_h,
                                float sigma_in)
{
    float delta_min = r->delta[0];
    float sigma_min = r->sigmas[0][0];
    int n_sca = r->n_sca;
    for (int o = 0; o < r->n_oct; o++) {
        int w = r->w[o], h = r->h[o];
        size_t n = (size_t)w * h;
        float delta = r->delta[o];
        float *g = r->gss[o] = malloc(n * n_sca * sizeof(float));
        if (o == 0) {
            float se = (float)(sqrt((double)(sigma_min * sigma_min - sigma_in * sigma_in))
                               / (double)delta_min);
            if (delta_min < 1) {
                float *up = malloc(n * sizeof(float));
                os_oversample(image, im_w, im_h, up, w, h, delta_min);
                os_blur(up, g, w, h, se);
                free(up);
            } else {
                os_blur(image, g, w, h, se);
            }
        } else {
            int nspo = n_sca - 3;
            os_subsample2(r->gss[o - 1] + (size_t)nspo * r->w[o - 1] * r->h[o - 1], g,
                          r->w[o - 1], r->h[o - 1]);
        }
        for (int s = 1; s < n_sca; s++) {
            float sp = r->sigmas[o][s - 1], sn = r->sigmas[o][s];
            float se = (float)(sqrt((double)(sn * sn - sp * sp)) / (double)delta);
            os_blur(g + (size_t)(s - 1) * n, g + (size_t)s * n, w, h, se);
        }
        float *d = r->dog[o] = malloc(n * (n_sca - 1) * sizeof(float));
        for (int s = 0; s < n_sca - 1; s++)
            for (size_t p = 0; p < n; p++)
                d[s * n + p] = g[(s + 1) * n + p] - g[s * n + p];
    }
}

/* ---------------------------------------------------------------- lists ---- */

typedef struct { os_point *v; int n, cap; } os_list;
static void os_push(os_list *l, const os_point *p)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 256;
        l->v = realloc(l->v, (size_t)l->cap * sizeof(os_point));
    }
    l->v[l->n++] = *p;
}

/* Strict 26-neighbour extrema of the DoG stack, scanned in (o,s,i,j) order.
 * [lib_sift_anatomy.c:240-313] */
static void os_find_extrema(const os_result *r, os_list *out)
{
    for (int o = 0; o < r->n_oct; o++) {
        int ns = r->n_sca - 1, w = r->w[o], h = r->h[o];
        const float *d = r->dog[o];
        size_t n = (size_t)w * h;
        for (int s = 1; s < ns - 1; s++)
            for (int i = 1; i < h - 1; i++)
                for (int j = 1; j < w - 1; j++) {
                    const float *c = d + s * n + (size_t)i * w + j;
                    float v = *c;
                    int is_min = 1, is_max = 1;
                    for (int ds = -1; ds <= 1 && (is_min || is_max); ds++)
                        for (int di = -1; di <= 1; di++)
                            for (int dj = -1; dj <= 1; dj++) {
                                if (!ds && !di && !dj) continue;
                                float nb = c[(ptrdiff_t)ds * (ptrdiff_t)n + di * w + dj];
                                if (nb <= v) is_min = 0;
                                if (nb >= v) is_max = 0;
                            }
                    if (is_min || is_max) {
                        os_point p;
                        memset(&p, 0, sizeof p);
                        p.o = o; p.s = s; p.i = i; p.j = j;
                        p.x = r->delta[o] * i;
                        p.y = r->delta[o] * j;
                        p.sigma = r->sigmas[o][s];
                        p.val = v;
                        os_push(out, &p);
                    }
                }
    }
}

/* [lib_sift_anatomy.c:326-338] */
static void os_keep_contrast(const os_list *in, os_list *out, float thr)
{
    for (int k = 0; k < in->n; k++) {
        float v = in->v[k].val;
        if ((v < 0 ? -v : v) > thr) os_push(out, &in->v[k]);
    }
}

/* One quadratic fit at (s,i,j). [lib_sift_anatomy.c:358-405] */
static void os_taylor(const float *st, int w, int h, int i, int j, int s,
                      float *di, float *dj, float *ds, float *val)
{
    size_t n = (size_t)w * h;
#define D(ss, ii, jj) st[(size_t)(ss) * n + (size_t)(ii) * w + (jj)]
    float c = D(s, i, j);
    float hXX = D(s, i - 1, j) + D(s, i + 1, j) - 2 * c;
    float hYY = D(s, i, j + 1) + D(s, i, j - 1) - 2 * c;
    float hSS = D(s + 1, i, j) + D(s - 1, i, j) - 2 * c;
    float hXY = (float)(0.25 * (double)((D(s, i + 1, j + 1) - D(s, i + 1, j - 1))
                                       - (D(s, i - 1, j + 1) - D(s, i - 1, j - 1))));
    float hXS = (float)(0.25 * (double)((D(s + 1, i + 1, j) - D(s + 1, i - 1, j))
                                       - (D(s - 1, i + 1, j) - D(s - 1, i - 1, j))));
    float hYS = (float)(0.25 * (double)((D(s + 1, i, j + 1) - D(s + 1, i, j - 1))
                                       - (D(s - 1, i, j + 1) - D(s - 1, i, j - 1))));
    float gX = (float)(0.5 * (double)(D(s, i + 1, j) - D(s, i - 1, j)));
    float gY = (float)(0.5 * (double)(D(s, i, j + 1) - D(s, i, j - 1)));
    float gS = (float)(0.5 * (double)(D(s + 1, i, j) - D(s - 1, i, j)));
#undef D
    float det = hXX * hYY * hSS - hXX * hYS * hYS - hXY * hXY * hSS
              + 2 * hXY * hXS * hYS - hXS * hXS * hYY;
    float aa = (hYY * hSS - hYS * hYS) / det;
    float ab = (hXS * hYS - hXY * hSS) / det;
    float ac = (hXY * hYS - hXS * hYY) / det;
    float bb = (hXX * hSS - hXS * hXS) / det;
    float bc = (hXY * hXS - hXX * hYS) / det;
    float cc = (hXX * hYY - hXY * hXY) / det;
    float oX = -aa * gX - ab * gY - ac * gS;
    float oY = -ab * gX - bb * gY - bc * gS;
    float oS = -ac * gX - bc * gY - cc * gS;
    float oV = (float)(0.5 * (double)(gX * oX + gY * oY + gS * oS));
    *di = oX; *dj = oY; *ds = oS;
    *val = c + oV;
}

/* [lib_sift_anatomy.c:426-505] */
static void os_refine(const os_result *r, const os_list *in, os_list *out, int itermax)
{
    const float ofst_max = 0.6f;
    float ratio = r->sigmas[0][1] / r->sigmas[0][0];
    for (int k = 0; k < in->n; k++) {
        os_point key = in->v[k];
        int o = key.o, w = r->w[o], h = r->h[o], ns = r->n_sca - 1;
        const float *st = r->dog[o];
        int ic = key.i, jc = key.j, sc = key.s, it = 0, conv = 0;
        float oX = 0.f, oY = 0.f, oS = 0.f, val = key.val;
        while (it < itermax) {
            if (0 < ic && ic < h - 1 && 0 < jc && jc < w - 1)
                os_taylor(st, w, h, ic, jc, sc, &oX, &oY, &oS, &val);
            else
                oX = oY = oS = 5.0f;
            if (fabsf(oX) < ofst_max && fabsf(oY) < ofst_max && fabsf(oS) < ofst_max) {
                conv = 1;
                break;
            }
            if (oX > +ofst_max && ic + 1 < h - 1) ic += 1;
            if (oX < -ofst_max && ic - 1 > 0) ic -= 1;
            if (oY > +ofst_max && jc + 1 < w - 1) jc += 1;
            if (oY < -ofst_max && jc - 1 > 0) jc -= 1;
            if (oS > +ofst_max && sc + 1 < ns - 1) sc += 1;
            if (oS < -ofst_max && sc - 1 > 0) sc -= 1;
            it++;
        }
        if (conv) {
            key.x = (ic + oX) * r->delta[o];
            key.y = (jc + oY) * r->delta[o];
            key.i = ic; key.j = jc; key.s = sc;
            key.sigma = (float)((double)r->sigmas[o][sc] * pow((double)ratio, (double)oS));
            key.val = val;
            os_push(out, &key);
        }
    }
}

/* [lib_sift_anatomy.c:525-546] */
static void os_edge_response(const os_result *r, os_list *l)
{
    for (int k = 0; k < l->n; k++) {
        os_point *key = &l->v[k];
        int w = r->w[key->o], h = r->h[key->o], i = key->i, j = key->j;
        const float *im = r->dog[key->o] + (size_t)key->s * w * h;
        float c = im[(size_t)i * w + j];
        float hXX = im[(size_t)(i - 1) * w + j] + im[(size_t)(i + 1) * w + j] - 2 * c;
        float hYY = im[(size_t)i * w + j + 1] + im[(size_t)i * w + j - 1] - 2 * c;
        float hXY = (float)(1. / 4 * (double)((im[(size_t)(i + 1) * w + j + 1] - im[(size_t)(i + 1) * w + j - 1])
                                            - (im[(size_t)(i - 1) * w + j + 1] - im[(size_t)(i - 1) * w + j - 1])));
        key->edge = (hXX + hYY) * (hXX + hYY) / (hXX * hYY - hXY * hXY);
    }
}

/* [lib_sift_anatomy.c:563-575] */
static void os_keep_not_edge(const os_list *in, os_list *out, float thr)
{
    for (int k = 0; k < in->n; k++)
        if (fabsf(in->v[k].edge) <= thr) os_push(out, &in->v[k]);
}

/* [lib_sift_anatomy.c:685-703] */
static void os_keep_inside(const os_list *in, os_list *out, int w, int h, float lambda)
{
    for (int k = 0; k < in->n; k++) {
        float x = in->v[k].x, y = in->v[k].y, sg = in->v[k].sigma;
        int ok = ((double)(x - lambda * sg) > 0.0) && (x + lambda * sg < (float)h)
              && ((double)(y - lambda * sg) > 0.0) && (y + lambda * sg < (float)w);
        if (ok) os_push(out, &in->v[k]);
    }
}

/* ----------------------------------------------------------- description --- */

/* [lib_util.c:187-198] */
static float os_modulus(float x, float y)
{
    float z = x;
    int n;
    if (z < 0) {
        n = (int)((-z) / y) + 1;
        z += n * y;
    }
    n = (int)(z / y);
    z -= n * y;
    return z;
}

/* [lib_description.c:79-85] */
static int os_ori_to_bin(float ori, int nbins)
{
    if (ori < 0) ori = (float)((double)ori + 2 * OS_PI);
    return (int)((double)ori / (2 * OS_PI) * nbins + 0.5) % nbins;
}

/* [lib_description.c:88-94] */
static float os_bin_to_ori(float bin, int nbins)
{
    float ori = (float)(((double)bin + 0.5) * 2 * OS_PI / (double)(float)nbins);
    if ((double)ori > OS_PI) ori = (float)((double)ori - 2 * OS_PI);
    return ori;
}

/* Gaussian-weighted gradient-orientation histogram around (x,y,sigma) in octave pixels.
 * `im` is the Gaussian image of the keypoint's scale. [lib_description.c:122-169] */
void os_orientation_histogram(float xk, float yk, float sk, const float *im, int w, int h,
                              int nbins, float lambda_ori, float *hist)
{
    for (int b = 0; b < nbins; b++) hist[b] = 0.f;
    float R = 3 * lambda_ori * sk;
    int i0 = imax(0, (int)((double)(xk - R) + 0.5));
    int j0 = imax(0, (int)((double)(yk - R) + 0.5));
    int i1 = imin((int)((double)(xk + R) + 0.5), h - 1);
    int j1 = imin((int)((double)(yk + R) + 0.5), w - 1);
    for (int si = i0; si <= i1; si++)
        for (int sj = j0; sj <= j1; sj++) {
            float sX = (si - xk) / sk;
            float sY = (sj - yk) / sk;
            float dx = os_grad_i(im, w, h, si, sj);
            float dy = os_grad_j(im, w, h, si, sj);
            float ori = os_modulus((float)atan2((double)dy, (double)dx), (float)(2 * OS_PI));
            float r2 = sX * sX + sY * sY;
            float M = (float)(hypot((double)dx, (double)dy)
                              * exp((double)(-r2 / (2 * lambda_ori * lambda_ori))));
            hist[os_ori_to_bin(ori, nbins)] += M;
        }
}

/* Six passes of a circular 3-tap box. [lib_description.c:406-423] */
static void os_smooth_hist(float *hist, int nbins)
{
    float tmp[256];
    for (int it = 0; it < 6; it++) {
        for (int b = 0; b < nbins; b++) tmp[b] = hist[b];
        for (int b = 0; b < nbins; b++) {
            int p = (b - 1 + nbins) % nbins, n = (b + 1) % nbins;
            hist[b] = (float)((double)(tmp[p] + tmp[b] + tmp[n]) / 3.);
        }
    }
}

/* Peaks above t*max, refined by a parabola. Smooths `hist` in place.
 * [lib_description.c:201-227, 172-176; lib_util.c:64-96] */
int os_principal_orientations(float *hist, int nbins, float t, float *out)
{
    int n = 0;
    os_smooth_hist(hist, nbins);
    double mx = -3.402823466e+38;
    for (int b = 0; b < nbins; b++) if ((double)hist[b] > mx) mx = hist[b];
    float max_value = (float)mx;
    for (int b = 0; b < nbins; b++) {
        int p = (b - 1 + nbins) % nbins, q = (b + 1) % nbins;
        if (hist[b] > t * max_value && hist[b] > hist[p] && hist[b] > hist[q]) {
            float h1 = hist[p], h2 = hist[b], h3 = hist[q];
            float off = (h1 - h3) / (2 * (h1 + h3 - 2 * h2));
            out[n++] = os_bin_to_ori((float)b + off, nbins);
        }
    }
    return n;
}

/* Single dominant orientation. [lib_description.c:231-250] */
float os_one_orientation(float *hist, int nbins)
{
    os_smooth_hist(hist, nbins);
    int b = 0;
    double mx = -3.402823466e+38;
    for (int k = 0; k < nbins; k++) if ((double)hist[k] > mx) { b = k; mx = hist[k]; }
    int p = (b - 1 + nbins) % nbins, q = (b + 1) % nbins;
    float h1 = hist[p], h2 = hist[b], h3 = hist[q];
    float off = (h1 - h3) / (2 * (h1 + h3 - 2 * h2));
    return os_bin_to_ori((float)b + off, nbins);
}

/* 4x4x8-style descriptor accumulation. [lib_description.c:278-347; lib_util.c:201-209] */
void os_feature_vector(float xk, float yk, float sk, float theta, const float *im, int w, int h,
                       int Nhist, int Nbins, float lambda_descr, float *descr)
{
    for (int b = 0; b < Nhist * Nhist * Nbins; b++) descr[b] = 0.f;
    float R = (1 + 1 / (float)Nhist) * lambda_descr * sk;
    float Rp = (float)(OS_SQRT2 * (double)R);
    int i0 = imax(0, (int)((double)(xk - Rp) + 0.5));
    int j0 = imax(0, (int)((double)(yk - Rp) + 0.5));
    int i1 = imin((int)((double)(xk + Rp) + 0.5), h - 1);
    int j1 = imin((int)((double)(yk + Rp) + 0.5), w - 1);
    float ma = -theta;
    float c = (float)cos((double)ma), s = (float)sin((double)ma);
    for (int si = i0; si < i1; si++)
        for (int sj = j0; sj < j1; sj++) {
            float X0 = si - xk, Y0 = sj - yk;
            float X = c * X0 - s * Y0;
            float Y = s * X0 + c * Y0;
            float aX = X < 0 ? -X : X, aY = Y < 0 ? -Y : Y;
            if (!((aX < aY ? aY : aX) < R)) continue;
            double dx = os_grad_i(im, w, h, si, sj);
            double dy = os_grad_j(im, w, h, si, sj);
            float ori = (float)(atan2(dy, dx) - (double)theta);
            ori = os_modulus(ori, (float)(2 * OS_PI));
            double t = lambda_descr * sk;
            double M = hypot(dx, dy) * exp((double)(-(X * X + Y * Y)) / (2 * t * t));
            float alpha = (float)((double)(X / (2 * lambda_descr * sk / Nhist)) + (Nhist - 1.0) / 2.0);
            float beta = (float)((double)(Y / (2 * lambda_descr * sk / Nhist)) + (Nhist - 1.0) / 2.0);
            float gamma = (float)((double)ori / (2 * OS_PI) * Nbins);
            int a0 = (int)floor((double)alpha), b0 = (int)floor((double)beta);
            for (int i = imax(0, a0); i <= imin(a0 + 1, Nhist - 1); i++)
                for (int j = imax(0, b0); j <= imin(b0 + 1, Nhist - 1); j++) {
                    float da = (float)i - alpha, db = (float)j - beta;
                    if (da < 0) da = -da;
                    if (db < 0) db = -db;
                    int k = ((int)gamma + Nbins) % Nbins;
                    float *bin = &descr[i * Nhist * Nbins + j * Nbins + k];
                    *bin = (float)((double)*bin
                        + (1. - ((double)gamma - floor((double)gamma))) * (1.0 - (double)da) * (1.0 - (double)db) * M);
                    k = ((int)gamma + 1 + Nbins) % Nbins;
                    bin = &descr[i * Nhist * Nbins + j * Nbins + k];
                    *bin = (float)((double)*bin
                        + (1.0 - (floor((double)gamma) + 1 - (double)gamma)) * (1.0 - (double)da) * (1.0 - (double)db) * M);
                }
        }
}

/* Clip at thr*norm, renormalise, scale to 512, truncate, cap at 255.
 * [lib_description.c:374-389; lib_util.c:156-164] */
void os_quantize(float *descr, int n, float thr)
{
    float l2 = 0;
    for (int i = 0; i < n; i++) l2 += descr[i] * descr[i];
    l2 = (float)sqrt((double)l2);
    for (int i = 0; i < n; i++) {
        float cap = thr * l2;
        descr[i] = descr[i] < cap ? descr[i] : cap;
    }
    l2 = 0;
    for (int i = 0; i < n; i++) l2 += descr[i] * descr[i];
    l2 = (float)sqrt((double)l2);
    for (int i = 0; i < n; i++) {
        descr[i] = (float)(int)((double)descr[i] * 512.0 / (double)l2);
        descr[i] = descr[i] < 255 ? descr[i] : 255;
    }
}

/* ------------------------------------------------------------- the driver -- */

/* flags: bit0 = skip orientation+descriptor (detection only). */
os_result *os_sift(const float *x, int w, int h, const os_params *p, int flags)
{
    os_result *r = calloc(1, sizeof *r);
    int n_oct = os_number_of_octaves(w, h, p);                    /* [lib_sift_anatomy.c:831] */
    float thresh = os_convert_threshold(p);
    os_layout(r, n_oct, p->n_spo, w, h, p->delta_min, p->sigma_min);
    r->thresh = thresh;
    r->thresh_pre = (float)(0.8 * (double)thresh);                 /* [lib_sift_anatomy.c:855] */
    r->thresh_edge = (p->C_edge + 1) * (p->C_edge + 1) / p->C_edge; /* [lib_sift_anatomy.c:859] */
    os_build_scalespace(r, x, w, h, p->sigma_in);

    os_list L[6];
    memset(L, 0, sizeof L);
    os_find_extrema(r, &L[0]);
    os_keep_contrast(&L[0], &L[1], r->thresh_pre);
    os_refine(r, &L[1], &L[2], p->itermax);
    os_keep_contrast(&L[2], &L[3], thresh);
    os_edge_response(r, &L[3]);
    os_keep_not_edge(&L[3], &L[4], r->thresh_edge);
    os_keep_inside(&L[4], &L[5], w, h, 1.0f);
    for (int k = 0; k < 6; k++) { r->stage[k] = L[k].v; r->n_stage[k] = L[k].n; }

    r->n_descr = p->n_hist * p->n_hist * p->n_ori;
    r->n_bins = p->n_bins;
    if (flags & 1) return r;

    /* orientations: one output key per peak [lib_sift_anatomy.c:599-643] */
    os_list K = {0};
    float *oh = NULL;
    int oh_cap = 0;
    float hist[256], peaks[256];
    for (int k = 0; k < L[5].n; k++) {
        os_point key = L[5].v[k];
        int o = key.o, ow = r->w[o], ohh = r->h[o];
        float d = r->delta[o];
        const float *im = r->gss[o] + (size_t)key.s * ow * ohh;
        os_orientation_histogram(key.x / d, key.y / d, key.sigma / d, im, ow, ohh,
                                 p->n_bins, p->lambda_ori, hist);
        int np = os_principal_orientations(hist, p->n_bins, p->t, peaks);
        for (int q = 0; q < np; q++) {
            key.theta = peaks[q];
            os_push(&K, &key);
            if (K.n > oh_cap) {
                oh_cap = oh_cap ? 2 * oh_cap : 256;
                oh = realloc(oh, (size_t)oh_cap * p->n_bins * sizeof(float));
            }
            memcpy(oh + (size_t)(K.n - 1) * p->n_bins, hist, p->n_bins * sizeof(float));
        }
    }
    r->keys = K.v; r->n_keys = K.n; r->orihist = oh;

    /* descriptors [lib_sift_anatomy.c:724-765] */
    r->descr = malloc((size_t)(K.n ? K.n : 1) * r->n_descr * sizeof(float));
    for (int k = 0; k < K.n; k++) {
        os_point *key = &K.v[k];
        int o = key->o, ow = r->w[o], ohh = r->h[o];
        float d = r->delta[o];
        const float *im = r->gss[o] + (size_t)key->s * ow * ohh;
        float *out = r->descr + (size_t)k * r->n_descr;
        os_feature_vector(key->x / d, key->y / d, key->sigma / d, key->theta, im, ow, ohh,
                          p->n_hist, p->n_ori, p->lambda_descr, out);
        os_quantize(out, r->n_descr, 0.2f);
    }
    return r;
}

/* ------------------------------------------------------------------ matching --
 * [lib_matching.c:57-135] with [lib_util.c:108-153].  d1: n1 x dim, d2: n2 x dim (n2 >= 2).
 * For each row of d1: indices and distances of its two nearest rows of d2 and whether the
 * pair passes (flag 1: dA/dB < thresh, flag 0: dA < thresh).  Returns the number passing. */
int os_matching(const float *d1, int n1, const float *d2, int n2, int dim, float thresh, int flag,
                int *iA, int *iB, float *dA, float *dB, unsigned char *pass)
{
    int count = 0;
    float *dist = (float *)malloc((size_t)(n2 > 0 ? n2 : 1) * sizeof(float));
    for (int i = 0; i < n1; i++) {
        for (int j = 0; j < n2; j++) {                 /* [lib_util.c:137-153] */
            float d = 0.0f;
            for (int k = 0; k < dim; k++) {
                float t = d1[(size_t)i * dim + k] - d2[(size_t)j * dim + k];
                d += t * t;
            }
            dist[j] = (float)sqrt(d);
        }
        int a, b;                                      /* [lib_util.c:108-134] */
        float ma, mb;
        if (dist[0] < dist[1]) { a = 0; b = 1; ma = dist[0]; mb = dist[1]; }
        else                   { a = 1; b = 0; ma = dist[1]; mb = dist[0]; }
        for (int j = 2; j < n2; j++) {
            if (dist[j] < ma)      { mb = ma; ma = dist[j]; b = a; a = j; }
            else if (dist[j] < mb) { mb = dist[j]; b = j; }
        }
        iA[i] = a; iB[i] = b; dA[i] = ma; dB[i] = mb;
        float val = (flag == 1 ? ma / mb : ma);        /* [lib_matching.c:115] */
        pass[i] = (unsigned char)(val < thresh);
        count += pass[i];
    }
    free(dist);
    return count;
}

/* -------------------------------------------------------------------- ingest --
 * 8-bit interleaved pixels -> the float image handed to sift_anatomy by the command-line
 * tool: [io_png.c:491-520] (gray kept, RGB weighted in double, alpha dropped by
 * PNG_TRANSFORM_STRIP_ALPHA at :486-487) then [sift_cli.c:280-281]. */
void os_ingest_u8(const unsigned char *in, int channels, float *out, size_t n_px)
{
    for (size_t p = 0; p < n_px; p++) {
        float g;
        if (channels == 1) {
            g = (float)in[p];
        } else {
            float r = (float)in[p * channels], gg = (float)in[p * channels + 1], b = (float)in[p * channels + 2];
            g = (float)(0.212639005871510 * r + 0.715168678767756 * gg + 0.072192315360734 * b);
        }
        g /= 256.0;
        out[p] = g;
    }
}


void os_free(os_result *r)
{
    if (!r) return;
    for (int o = 0; o < r->n_oct; o++) { free(r->gss[o]); free(r->dog[o]); }
    for (int k = 0; k < 6; k++) free(r->stage[k]);
    free(r->keys); free(r->descr); free(r->orihist);
    free(r);
}

/* 64-bit FNV-1a over the raw bits of a float buffer: the digest stored in tests/golden. */
uint64_t os_hash_floats(const float *v, size_t n)
{
    const unsigned char *b = (const unsigned char *)v;
    uint64_t hsh = 1469598103934665603ULL;
    for (size_t k = 0; k < n * sizeof(float); k++) { hsh ^= b[k]; hsh *= 1099511628211ULL; }
    return hsh;
}
