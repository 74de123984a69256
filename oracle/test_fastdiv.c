/* Test infrastructure (like everything under oracle/): CPU proof-by-enumeration of the two arithmetic shortcuts the HIP
 * kernels take in the bilinear sampler / flow_check, restated here in plain C:
 *   (1) psfm_div_r  (particle-sfm_amd/csrc/psfm_device.h): x / c through the refined reciprocal r of the launch-invariant
 *       divisor c = (size-1)/2 and two fma residual corrections -- must equal the correctly rounded quotient x / c that
 *       torch's grid_sample computes (trajectory.py:25-37), for every c = (W-1)/2 and r off by up to 1 ulp;
 *   (2) psfm_sq_threshold (psfm_chain.h): sqrtf(s) > thres  <=>  s > t2  (utils.py:87-88 without the square root).
 * Usage: test_fastdiv [divisions per W (default 20000)]  -> exit code 0 when no counter-example exists. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float div_r(float x, float c, float r)
{
    float q = x * r;
    float e = fmaf(-c, q, x);
    q = fmaf(e, r, q);
    e = fmaf(-c, q, x);
    return fmaf(e, r, q);
}
static inline float rcp_refined(float c, float r0)
{
    const float e = fmaf(-c, r0, 1.0f);
    return fmaf(e, r0, r0);
}
static float sq_threshold(float thres)
{
    if (thres != thres) return INFINITY;
    if (thres < 0.0f) return -1.0f;
    if (thres == INFINITY) return INFINITY;
    float t2 = thres * thres;
    if (t2 == INFINITY) t2 = 3.4028234663852886e38f;
    while (sqrtf(t2) > thres) t2 = nextafterf(t2, -INFINITY);
    while (t2 < 3.4028234663852886e38f && sqrtf(nextafterf(t2, INFINITY)) <= thres) t2 = nextafterf(t2, INFINITY);
    return t2;
}
static uint64_t st = 88172645463325252ull;
static inline uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

int main(int argc, char** argv)
{
    const int per_w = argc > 1 ? atoi(argv[1]) : 20000;
    long bad = 0, n = 0;
    for (int W = 2; W <= 8192; ++W) {
        const float c = (float)((double)(W - 1) / 2.0);
        const float r_exact = (float)(1.0 / (double)c);
        for (int pert = -1; pert <= 1; ++pert) {
            float r0 = r_exact;
            if (pert) { uint32_t u; memcpy(&u, &r0, 4); u += (uint32_t)pert; memcpy(&r0, &u, 4); }
            const float r = rcp_refined(c, r0);
            for (int i = 0; i < per_w; ++i) {
                float x;
                const uint64_t k = rnd();
                switch (k & 3) {
                    case 0: x = (float)((k >> 8) % (uint64_t)(W + 40)) - 20.0f + (float)((k >> 40) & 0xffff) / 65536.0f; break;
                    case 1: { uint32_t u = (uint32_t)(k >> 16); memcpy(&x, &u, 4); } break;
                    case 2: x = ldexpf((float)((k >> 8) & 0xffffff) / 16777216.0f + 1.0f, (int)((k >> 40) % 40) - 10); break;
                    default: x = (float)((k >> 8) % 4096) * 0.5f; break;
                }
                const float ax = fabsf(x);
                if (!(ax < 1e30f) || (ax < 1e-30f && ax != 0.0f)) continue;
                const float q1 = x / c, q2 = div_r(x, c, r);
                ++n;
                if (memcmp(&q1, &q2, 4) != 0 && !(q1 == 0.0f && q2 == 0.0f)) {
                    if (bad < 10) printf("division: W=%d x=%a: %a vs %a (r off by %d ulp)\n", W, x, q1, q2, pert);
                    ++bad;
                }
            }
        }
    }
    printf("%ld divisions checked, %ld counter-examples\n", n, bad);
    /* (2) thresholds: every float within 64 ulps of thres^2, plus the special values */
    const float ths[] = {1.0f, 3.0f, 0.5f, 0.1f, 2.5f, 1e-3f, 7.25f, 100.0f, 0.0f, 1e-20f, 1e19f, 3e19f, -1.0f, INFINITY, NAN};
    long m = 0, bad2 = 0;
    for (unsigned t = 0; t < sizeof(ths) / sizeof(ths[0]); ++t) {
        const float th = ths[t], t2 = sq_threshold(th);
        float cand[400];
        int nc = 0;
        float s = th * th;
        if (s == s && s != INFINITY) {
            float lo = s, hi = s;
            for (int i = 0; i < 64; ++i) { lo = nextafterf(lo, -INFINITY); hi = nextafterf(hi, INFINITY); cand[nc++] = lo; cand[nc++] = hi; }
            cand[nc++] = s;
        }
        const float extra[] = {0.0f, 1e-45f, 1.0f, 3.4028234663852886e38f, INFINITY, NAN, 1e-10f, 1e10f};
        for (unsigned i = 0; i < 8; ++i) cand[nc++] = extra[i];
        for (int i = 0; i < nc; ++i) {
            const float v = cand[i];
            if (v < 0.0f) continue;      /* s = ev*ev + eu*eu is never negative */
            ++m;
            if ((sqrtf(v) > th) != (v > t2)) { if (bad2 < 10) printf("threshold: thres=%a s=%a t2=%a\n", th, v, t2); ++bad2; }
        }
    }
    printf("%ld threshold comparisons checked, %ld counter-examples\n", m, bad2);
    return (bad || bad2) ? 1 : 0;
}
