// Host build of the DEVICE's launch chain (particle-sfm_amd/csrc/psfm_pc_core.h + psfm_pc_control.h) for
// tests/test_pc_chain_host.py: one whole path-consistency solve of a batch of tracks the way psfm_pc_init_kernel /
// psfm_pc_iter_kernel / psfm_pc_persist_kernel run it -- iteration 0, then per round either the evaluate-ahead form
// (candidate of the step the control block names, its cost, the system there) or the refresh form (the system at x for a
// raised mu), the 13 sums of a round added in track order, pc_chain_control on them.  The loop bodies below restate
// pc_init_tracks / pc_iter_tracks of psfm_solver.hip line by line; everything they call is the device's own code.
// Test infrastructure (compiled by the test with g++ -O2 -mfma -ffp-contract=off).
#include <vector>

#include "psfm_pc_control.h"

// inject_fail_after: >= 1: the totals of the round in which the inject_fail_after-th step is accepted report a system that lost
// definiteness (SUM_FAIL) -- Ceres' FAILURE behind accepted steps, which no real batch of these tests produces: the solve must
// hand the START values back (Summary::IsSolutionUsable() is false), not the iterate it had reached.
extern "C" int pc_host_chain_solve_ex(long n, const double* x0, const double* ref1, const double* ref2, const double* scale,
                                      const float* flow, int H, int W, int pair, int inject_fail_after, double* x_out, int* stats,
                                      double* costs)
{
    const PcF2* F = (const PcF2*)flow;
    std::vector<double> b1(4 * n), b2(4 * n), js(2 * n);
    auto buf = [&](int m) -> double* { return m == 0 ? const_cast<double*>(x0) : (m == 1 ? b1.data() : b2.data()); };
    auto eval = [&](const double* x, long i, double s, double* r, double* j) {
        if (pair) pc_core_eval<true>(F, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
        else pc_core_eval<false>(F, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
    };
    double tot[PC_NSUM];
    for (int k = 0; k < PC_NSUM; ++k) tot[k] = 0.0;
    // ---- iteration 0 (pc_init_tracks, batch mode: references and scale are given) ----
    {
        const double mu = 1e-8;
        for (long i = 0; i < n; ++i) {
            const double s = scale[i];
            double r0[6], j0[4];
            eval(x0 + 4 * i, i, s, r0, j0);
            const PcConst c = pc_core_const(s, j0);
            js[2 * i] = c.S0q; js[2 * i + 1] = c.S1q;
            tot[SUM_CNT] += 1.0;
            tot[SUM_COST0] += pc_core_cost(r0);
            PcSys y;
            pc_core_system<true>(x0 + 4 * i, r0, j0, c, mu, pc_core_iA22(c, mu), tot, y, CH_QUD, CH_QDD);
        }
    }
    PsfmSolveCtrl C;
    pc_chain_control(C, tot, 0);
    C.launches += 1;
    int rounds = 0;
    while (!C.done && rounds < 2 * 200 + 64) {
        ++rounds;
        const double* xc = buf(C.cur);
        double* xn = buf(pc_other(C.cur));
        const double mu = C.mu, a = C.dl_a, b = C.dl_b;
        const bool refresh = C.kind_next != 0;
        const double mu_next = fmax(1e-8, 2.0 * mu / 10.0);
        for (int k = 0; k < PC_NSUM; ++k) tot[k] = 0.0;
        for (long i = 0; i < n; ++i) {
            const double s = scale[i];
            PcConst c;
            c.s = s; c.S0q = js[2 * i]; c.S1q = js[2 * i + 1]; c.H22 = fma(s, s, 1.0);      // pc_const_load
            const double* x = xc + 4 * i;
            double r[6], jac[4];
            PcSys y;
            eval(x, i, s, r, jac);
            if (refresh) {
                pc_core_system<true>(x, r, jac, c, mu, pc_core_iA22(c, mu), tot, y, CH_QUD, CH_QDD);
                continue;
            }
            double unused[PC_NSUM], xp[4];
            for (int k = 0; k < PC_NSUM; ++k) unused[k] = 0.0;
            pc_core_system<false>(x, r, jac, c, mu, pc_core_iA22(c, mu), unused, y, 0, 0);
            pc_core_step<false, false>(x, r, jac, c, y, a, b, tot, xp);
            for (int k = 0; k < 4; ++k) xn[4 * i + k] = xp[k];
            eval(xp, i, s, r, jac);
            tot[SUM_COST] += pc_core_cost(r);
            pc_core_system<true>(xp, r, jac, c, mu_next, pc_core_iA22(c, mu_next), tot, y, CH_QUD, CH_QDD);
        }
        if (inject_fail_after > 0 && !refresh && C.successful + 1 == inject_fail_after) tot[SUM_FAIL] += 1.0;   // (counts if this round accepts)
        pc_chain_control(C, tot, 1);
        C.launches += 1;
    }
    // ---- write-back (pc_writeback_tracks): the accepted iterate; a failed solve hands the START values back, whatever it had
    //      accepted on the way (buffer 0 is never written before this point) ----
    const double* xf = buf(C.failed ? 0 : C.cur);
    for (long i = 0; i < 4 * n; ++i) x_out[i] = xf[i];
    stats[0] = C.iteration; stats[1] = C.successful; stats[2] = C.n_tracks == 0 ? -1 : C.termination; stats[3] = C.nonGN;
    stats[4] = C.launches; stats[5] = C.done; stats[6] = C.failed;
    if (C.failed) stats[2] = PSFM_TERM_FAILURE;
    costs[0] = C.initial_cost; costs[1] = C.x_cost;
    return C.done ? 0 : 1;
}

extern "C" int pc_host_chain_solve(long n, const double* x0, const double* ref1, const double* ref2, const double* scale,
                                   const float* flow, int H, int W, int pair, double* x_out, int* stats, double* costs)
{
    return pc_host_chain_solve_ex(n, x0, ref1, ref2, scale, flow, H, W, pair, 0, x_out, stats, costs);
}

// ---- the FUSED solve (psfm_pc_fused_kernel / the frame kernels: pc_fused_body + pc_fused_replay of psfm_solver.hip) ----
// K trust-region iterations per track speculated as accepted Gauss-Newton steps at mu = min_mu, the K x 13 sums replayed by
// pc_control_step; when all K were accepted without ending the solve, continuation launches of up to two more iterations from
// the last iterate (pc_more_body); the first decision that is not "Gauss-Newton step accepted" ends the belief.
// Returns 0: solved as speculated (x_out = the accepted iterate; a failed solve hands the start values back);
//         1: not as speculated -- the product redoes the solve with the launch chain from the start values (x_out untouched).
#define PC_HOST_KMAX 8
extern "C" int pc_host_fused_solve(long n, const double* x0, const double* ref1, const double* ref2, const double* scale,
                                   const float* flow, int H, int W, int K, double* x_out, int* stats, double* costs)
{
    const PcF2* F = (const PcF2*)flow;
    const double mu = 1e-8;
    if (K < 1) K = 1;
    if (K > PC_HOST_KMAX) K = PC_HOST_KMAX;
    std::vector<std::vector<double>> it(PC_HOST_KMAX + 1);          // iterate m = the positions after m accepted steps
    it[0].assign(x0, x0 + 4 * n);
    std::vector<PcConst> cs(n);
    std::vector<double> c0(n);
    for (long i = 0; i < n; ++i) {                                    // pc_track_setup: Jacobi scaling at the start values
        double r[6], j[4];
        pc_core_eval<false>(F, H, W, x0 + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], scale[i], r, j);
        cs[i] = pc_core_const(scale[i], j);
        c0[i] = pc_core_cost(r);
    }
    auto rows_from = [&](int base, int n_it, bool first, double* tot) {   // n_it speculated iterations behind iterate `base`
        for (int q = 0; q < n_it * PC_NSUM; ++q) tot[q] = 0.0;
        for (int j = 0; j < n_it; ++j) it[base + j + 1].assign(4 * n, 0.0);
        for (long i = 0; i < n; ++i) {
            double x[4], r[6], jac[4];
            for (int k = 0; k < 4; ++k) x[k] = it[base][4 * i + k];
            pc_core_eval<false>(F, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], scale[i], r, jac);
            const double iA22 = pc_core_iA22(cs[i], mu);
            for (int j = 0; j < n_it; ++j) {
                double v[PC_NSUM], xp[4];
                for (int k = 0; k < PC_NSUM; ++k) v[k] = 0.0;
                pc_core_iteration<true>(x, r, jac, cs[i], mu, iA22, 0.0, 1.0, v, xp);           // pc_fused_iteration
                for (int k = 0; k < 4; ++k) { x[k] = xp[k]; it[base + j + 1][4 * i + k] = xp[k]; }
                pc_core_eval<false>(F, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], scale[i], r, jac);
                v[SUM_COST] += pc_core_cost(r);
                if (first && j == 0) { v[SUM_CNT] = 1.0; v[SUM_COST0] = c0[i]; }
                double* row = tot + j * PC_NSUM;
                for (int k = 0; k < PC_NSUM; ++k) row[k] = k == SUM_GMAX ? fmax(row[k], v[k]) : row[k] + v[k];
            }
        }
    };
    PsfmSolveCtrl C;
    memset(&C, 0, sizeof(C));
    double tot[PC_HOST_KMAX * PC_NSUM];
    int base = 0, n_it = K;
    bool first = true;
    for (;;) {
        rows_from(base, n_it, first, tot);
        for (int j = 1; j <= n_it; ++j) {                             // pc_fused_replay
            pc_control_step(C, tot + (j - 1) * PC_NSUM, first && j == 1, base + j);
            if (C.done) break;
            if (!(C.fresh_x && C.cur == base + j && C.dl_fixed == 0 && C.mu == 1e-8)) break;
        }
        C.launches += 1;
        if (C.done) break;
        const bool all_accepted = C.fresh_x && C.cur == base + n_it && C.dl_fixed == 0 && C.mu == 1e-8;
        if (!(all_accepted && C.cur + 1 <= PC_HOST_KMAX)) return 1;   // the launch chain takes over from the start values
        base = C.cur; n_it = PC_HOST_KMAX - base < 2 ? PC_HOST_KMAX - base : 2; first = false;      // pc_more_body
    }
    const std::vector<double>& xf = it[C.failed ? 0 : C.cur];
    for (long i = 0; i < 4 * n; ++i) x_out[i] = xf[i];
    stats[0] = C.iteration; stats[1] = C.successful; stats[2] = C.n_tracks == 0 ? -1 : C.termination; stats[3] = C.nonGN;
    stats[4] = C.launches; stats[5] = C.done; stats[6] = C.failed;
    if (C.failed) stats[2] = PSFM_TERM_FAILURE;
    costs[0] = C.initial_cost; costs[1] = C.x_cost;
    return 0;
}
