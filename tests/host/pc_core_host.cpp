// Host build of particle-sfm_amd/csrc/psfm_pc_core.h for tests/test_pc_core_host.py: one trust-region iteration of a batch
// of tracks, sums added in track order.  Test infrastructure (compiled by the test with g++ -O2 -mfma).
#include "psfm_pc_core.h"

extern "C" void pc_host_iteration(long n, const double* x0, const double* x, const double* ref1, const double* ref2,
                                  const double* scale, const float* flow, int H, int W, double mu, double a, double b, int gn,
                                  double* sums, double* xp_out, double* cost_at_x)
{
    for (int k = 0; k < PC_NSUM; ++k) sums[k] = 0.0;
    const PcF2* F = (const PcF2*)flow;
    for (long i = 0; i < n; ++i) {
        double r[6], j[4], xp[4];
        const double s = scale[i];
        pc_core_eval(F, H, W, x0 + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        const PcConst c = pc_core_const(s, j);
        pc_core_eval(F, H, W, x + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        cost_at_x[i] = pc_core_cost(r);
        const double iA22 = pc_core_iA22(c, mu);
        if (gn) pc_core_iteration<true>(x + 4 * i, r, j, c, mu, iA22, 0.0, 1.0, sums, xp);
        else pc_core_iteration<false>(x + 4 * i, r, j, c, mu, iA22, a, b, sums, xp);
        pc_core_eval(F, H, W, xp, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        sums[SUM_COST] += pc_core_cost(r);
        sums[SUM_CNT] += 1.0;
        for (int k = 0; k < 4; ++k) xp_out[4 * i + k] = xp[k];
    }
}

// the sums at x with the two extra ones (slots 0 and 3) the launch chain's control step prices dogleg steps with
extern "C" void pc_host_system(long n, const double* x0, const double* x, const double* ref1, const double* ref2,
                               const double* scale, const float* flow, int H, int W, double mu, double* sums)
{
    for (int k = 0; k < PC_NSUM; ++k) sums[k] = 0.0;
    const PcF2* F = (const PcF2*)flow;
    for (long i = 0; i < n; ++i) {
        double r[6], j[4];
        const double s = scale[i];
        pc_core_eval(F, H, W, x0 + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        const PcConst c = pc_core_const(s, j);
        pc_core_eval(F, H, W, x + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        PcSys y;
        pc_core_system<true>(x + 4 * i, r, j, c, mu, pc_core_iA22(c, mu), sums, y, 0, 3);
    }
}

// the four taps of an evaluation, as 8-byte loads (PAIR = false) and as the halves of two 16-byte loads (PAIR = true)
extern "C" void pc_host_taps(long n, const double* x, const float* flow, int H, int W, int pair, float* out)
{
    const PcF2* F = (const PcF2*)flow;
    for (long i = 0; i < n; ++i) {
        const PcTaps t = pair ? pc_core_taps<true>(F, H, W, x + 4 * i) : pc_core_taps<false>(F, H, W, x + 4 * i);
        float* o = out + 8 * i;
        o[0] = t.p00.x; o[1] = t.p00.y; o[2] = t.p01.x; o[3] = t.p01.y;
        o[4] = t.p10.x; o[5] = t.p10.y; o[6] = t.p11.x; o[7] = t.p11.y;
    }
}

// residuals (n,6) and Jacobians (n,6,4) of the residual blocks at x, through pc_core_eval (PAIR taps when pair != 0): what
// psfm_pc_eval_kernel writes on the device (tests/test_pc_eval_autograd.py)
extern "C" void pc_host_eval(long n, const double* x, const double* ref1, const double* ref2, const double* scale, const float* flow,
                             int H, int W, int pair, double* res, double* jac)
{
    const PcF2* F = (const PcF2*)flow;
    for (long i = 0; i < n; ++i) {
        double r[6], j[4];
        const double s = scale[i];
        if (pair) pc_core_eval<true>(F, H, W, x + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        else pc_core_eval<false>(F, H, W, x + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        for (int k = 0; k < 6; ++k) res[6 * i + k] = r[k];
        double* J = jac + 24 * i;
        for (int k = 0; k < 24; ++k) J[k] = 0.0;
        J[0] = 1.0; J[5] = 1.0; J[10] = s; J[15] = s;
        J[16] = j[0]; J[17] = j[1]; J[18] = 1.0;
        J[20] = j[2]; J[21] = j[3]; J[23] = 1.0;
    }
}
