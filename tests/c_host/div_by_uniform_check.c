/* CPU check of the three-instruction quotient the orientation and descriptor kernels use for
 * denominators that are the same for a whole keypoint (div_by_uniform in sift_kernels.hip):
 *   y = RN(1/b), q = RN(a*y), r = a - b*q (exact in an FMA), result RN(q + r*y)
 * against the correctly rounded a / b, for float and double: mantissas of all ones against
 * denominators just above a power of two and the reverse, every float mantissa of the numerator for a
 * set of denominators, and random operands.  Prints the number of differences; exit code 1 if any. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float dbu_f(float a, float b, float y) { float q = a * y; return fmaf(fmaf(-b, q, a), y, q); }
static double dbu_d(double a, double b, double y) { double q = a * y; return fma(fma(-b, q, a), y, q); }
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float mkf(uint32_t m, int e) { uint32_t u = ((uint32_t)(e + 127) << 23) | (m & 0x7fffff); float f; memcpy(&f, &u, 4); return f; }
static double mkd(uint64_t m, int e) { uint64_t u = ((uint64_t)(e + 1023) << 52) | (m & 0xfffffffffffffull); double f; memcpy(&f, &u, 8); return f; }

int main(void)
{
    long bad = 0, n = 0;
    for (uint32_t ma = 0x7fffff; ma > 0x7fffff - 2048; ma--)
        for (uint32_t mb = 0; mb < 2048; mb++) {
            float a = mkf(ma, 3), b = mkf(mb, 1), y = 1.0f / b;
            n++; if (dbu_f(a, b, y) != a / b) bad++;
            a = mkf(mb, 3); b = mkf(ma, 1); y = 1.0f / b;
            n++; if (dbu_f(a, b, y) != a / b) bad++;
        }
    for (long i = 0; i < 40000000; i++) {
        uint64_t r = rnd();
        float a = mkf((uint32_t)r, (int)((r >> 24) % 40) - 20), b = mkf((uint32_t)(r >> 32), (int)((r >> 56) % 20) - 10), y = 1.0f / b;
        n++; if (dbu_f(a, b, y) != a / b) bad++;
    }
    /* 2*pi (orientation), 2*lambda_ori^2 = 4.5, cell sizes and sigmas of the default parameters, edge mantissas */
    const float bs[] = {6.2831855f, 4.5f, 1.5f, 3.0f, 0.75f, 1.0000001f, 1.9999999f, 3.6000001f, 2.4000001f, 1.2262733f};
    for (unsigned k = 0; k < sizeof bs / sizeof bs[0]; k++) {
        const float b = bs[k], y = 1.0f / b;
        for (uint32_t m = 0; m < 0x800000; m++)
            for (int e = -3; e <= 6; e += 3) { const float a = mkf(m, e); n++; if (dbu_f(a, b, y) != a / b) bad++; }
    }
    printf("float: %ld of %ld differ\n", bad, n);
    long badd = 0, nd = 0;
    for (uint64_t ma = 0xfffffffffffffull; ma > 0xfffffffffffffull - 1500; ma--)
        for (uint64_t mb = 0; mb < 1500; mb++) {
            double a = mkd(ma, 3), b = mkd(mb, 1), y = 1.0 / b;
            nd++; if (dbu_d(a, b, y) != a / b) badd++;
            a = mkd(mb, 3); b = mkd(ma, 1); y = 1.0 / b;
            nd++; if (dbu_d(a, b, y) != a / b) badd++;
        }
    for (long i = 0; i < 40000000; i++) {
        uint64_t r = rnd(), r2 = rnd();
        double a = mkd(r, (int)(r >> 52) % 40 - 20), b = mkd(r2, (int)(r2 >> 52) % 20 - 10), y = 1.0 / b;
        nd++; if (dbu_d(a, b, y) != a / b) badd++;
    }
    /* the double denominators in use: 2*pi and 2*t*t of a few keypoint scales */
    const double bd[] = {6.283185307179586, 2 * 4.8 * 4.8, 2 * 9.6000003814697266 * 9.6000003814697266, 2 * 6.1 * 6.1};
    for (unsigned k = 0; k < sizeof bd / sizeof bd[0]; k++)
        for (long i = 0; i < 10000000; i++) {
            const double a = mkd(rnd(), (int)(rnd() % 20) - 10), b = bd[k], y = 1.0 / b;
            nd++; if (dbu_d(a, b, y) != a / b) badd++;
        }
    printf("double: %ld of %ld differ\n", badd, nd);
    return (bad || badd) ? 1 : 0;
}
