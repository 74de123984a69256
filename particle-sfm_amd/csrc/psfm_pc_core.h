// psfm_pc_core.h -- the per-track arithmetic of one trust-region iteration of the path-consistency solve
// (optimize/src/trajectory_optimize.cpp:30-96 as Ceres 2.0.0 runs it: TrustRegionMinimizer + TRADITIONAL_DOGLEG + Jacobi
// scaling + SPARSE_NORMAL_CHOLESKY), written for the structure of THIS problem instead of as a generic 4x4 solve.
//
// Per track x = (x1, y1, x2, y2), residuals (path_consistency_cost.h:50-57)
//     r = [ p1 - ref1 ; s (p2 - ref2) ; (p2 - p1) - F12(p1) ]
// and Jacobian  J = [ I2 0 ; 0 s I2 ; K I2 ]  with K = [j0 j1 ; j2 j3] = -(I + dF12/dp1).  Hence
//     H = J^T J = [ I + K^T K   K^T ; K   (s^2 + 1) I ],      g = J^T r.
// Ceres solves, in the space scaled by the Jacobi scaling S (computed once at x0) and the dogleg's diagonal
// D^2 = clamp(diag(Js^T Js), 1e-6, 1e32):   (Js^T Js + mu D^2) y = Js^T r,   gn = -D y,   step = gn / D,   x+ = x + S step.
// Substituting Js = J S and dividing the scaling back out gives the SAME step in the unscaled space:
//     (H + mu Hh) d = -g,     Hh = diag(D^2 / S^2)  (= diag(H) wherever the clamp is inactive, i.e. for every finite input
//                                                    whose columns are not absurdly scaled; the clamp is still applied)
// and every scalar Ceres reduces over the tracks follows without a square root:
//     |ghat|^2 = sum g_k u_k          with u = g / Hh        (ghat = Js^T r / D)
//     |gn|^2   = sum Hh_k d_k^2,      ghat . gn = sum g_k d_k
//     |Js (ghat / D)|^2 = |J u|^2     (the Cauchy-point denominator)
//     dogleg step a ghat + b gn  ->  unscaled step  dl = a u + b d,   |step|^2 = sum Hh_k dl_k^2
//     model decrease = -(J dl)^T (r + J dl / 2).
// The 4x4 system is solved through the 2x2 Schur complement of its lower-right block, which is (s^2 + 1 + mu Hh_22) I -- a
// per-track constant: one reciprocal per track and iteration (of the 2x2 determinant) instead of the four square roots,
// four reciprocals and fourteen divisions of a dense Cholesky factorisation + two triangular solves.  Multiply-adds are
// contracted (explicit fma).  ~190 f64 operations per track and iteration instead of ~560.
//
// Compared with the CPU oracle (the C restatement: a generic dense Cholesky in the scaled space, no contraction) the
// results differ by rounding only (~1e-13 px per solve on the parity runs); every DECISION of the trust-region loop --
// accept / reject, dogleg case, termination -- is tested to be the same (tests/test_gpu_solver.py,
// tests/test_gpu_whole_sequence.py).  north_star's bar is 1e-4 px.
//
// Host-compilable (tests/test_pc_core_host.py builds it with g++ and checks every sum of an iteration against a NumPy
// restatement of the scaled-space formulas); on the device the reciprocal is v_rcp_f64 + two Newton steps.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PC_HD __host__ __device__ __forceinline__
#else
#define PC_HD static inline
#endif

struct PcF2 { float x, y; };     // one flow vector (.flo-native interleaved layout == float2)

// sums slots (one row of 13 per trust-region iteration)
enum { SUM_MCC = 0, SUM_COST = 1, SUM_STEP2 = 2, SUM_DL2 = 3, SUM_XN2 = 4, SUM_GMAX = 5, SUM_G2 = 6, SUM_JG2 = 7,
       SUM_GN2 = 8, SUM_DOT = 9, SUM_FAIL = 10, SUM_CNT = 11, SUM_COST0 = 12 };
#define PC_NSUM 13

PC_HD double pc_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);          // ~2^-26 relative; two Newton steps -> within an ulp or two
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
#else
    return 1.0 / x;
#endif
}

// Per-track constants of a solve.
struct PcConst {
    double s;          // weight of the (p2 - ref2) residuals: (1 - occ02) * (|flow02| < 20), trajectory.py:179
    double S0q, S1q;   // squared Jacobi scaling of columns 0, 1: 1 / (1 + sqrt(colnorm^2 at x0))^2
    double H22;        // s^2 + 1 = H_22 = H_33.  It is also the dogleg diagonal of columns 2, 3: their scaling is
                       // S2 = 1 / (1 + sqrt(H22)) at every iterate (the columns do not depend on x), and
                       // S2^2 H22 = H22 / (1 + sqrt(H22))^2 lies in [1/4, 1) for every finite s -- the clamp never binds
                       // (a non-finite s fails the solve through iA22)
};

PC_HD double pc_clamped_diag(double Hkk, double Sq)
{
    // D_k^2 / S_k^2 with D_k^2 = clamp(S_k^2 H_kk, min_lm_diagonal, max_lm_diagonal) (dogleg_strategy.cc); the clamp only
    // binds for columns scaled by > 1e3 against x0's or non-finite ones (a cold path)
    const double n = Hkk * Sq;
    if (n >= 1e-6 && n <= 1e32) return Hkk;
    return fmin(fmax(n, 1e-6), 1e32) * pc_rcp(Sq);
}

// Jacobi scaling (trust_region_minimizer.cc, computed ONCE from the Jacobian at the start values).  j = the four Jacobian
// entries at x0.
PC_HD PcConst pc_core_const(double s, const double j[4])
{
    PcConst c;
    c.s = s;
    const double q0 = fma(j[2], j[2], fma(j[0], j[0], 1.0));
    const double q1 = fma(j[3], j[3], fma(j[1], j[1], 1.0));
    c.H22 = fma(s, s, 1.0);
    const double S0 = 1.0 / (1.0 + sqrt(q0)), S1 = 1.0 / (1.0 + sqrt(q1));
    c.S0q = S0 * S0; c.S1q = S1 * S1;
    return c;
}

// 1 / (H_22 + mu H_22): the inverse of the lower-right block of the damped system (a multiple of I2)
PC_HD double pc_core_iA22(const PcConst& c, double mu) { return pc_rcp(fma(mu, c.H22, c.H22)); }

// f64 clamp-to-edge bilinear interpolation of F12 at (row = y1, col = x1) (linear_interpolation.h:97-123 over
// ceres::Grid2D, path_consistency_cost.h:48), residuals and the four non-trivial Jacobian entries.
// f0 = a00 + tc (a01 - a00) etc.: the differences of two f32 values are exact in f64, so this is Ceres' (1-tc) a00 + tc a01
// with one rounding less.
// In two halves, so that a caller can have the four taps of one track in flight while it computes on another:
// pc_core_taps issues the loads, pc_core_eval_taps is everything behind them (it re-derives row / column from x: a
// dozen instructions against six registers held across the wait).
struct PcTaps { PcF2 p00, p01, p10, p11; };

PC_HD void pc_core_cell(const double x[4], int& row, int& col)
{
    double fr = floor(x[1]), fc = floor(x[0]);
    fr = fr > -1.0e9 ? fr : -1.0e9; fr = fr < 1.0e9 ? fr : 1.0e9;   // also maps NaN to -1e9
    fc = fc > -1.0e9 ? fc : -1.0e9; fc = fc < 1.0e9 ? fc : 1.0e9;
    row = (int)fr; col = (int)fc;
}

struct PcF4 { float x0, y0, x1, y1; };     // two flow vectors next to each other in a row
PC_HD PcF4 pc_load_pair(const char* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const PcF4*)p;               // one 16-byte load (8-byte aligned: fine for global memory)
#else
    PcF4 v;
    memcpy(&v, p, sizeof(v));
    return v;
#endif
}

// The two taps of a row are the halves of ONE 16-byte load of the columns (cb, cb + 1), cb = min(c0, W - 2): with the
// clamping at the borders c0 and c1 are each cb or cb + 1, whatever the position.  Half as many gather instructions per
// evaluation -- on scattered tracks every one of them is 64 separate cache-line look-ups.
template <bool PAIR>
PC_HD PcTaps pc_core_taps(const PcF2* flow, int H, int W, const double x[4])
{
    int row, col;
    pc_core_cell(x, row, col);
    const int r0 = row < 0 ? 0 : (row > H - 1 ? H - 1 : row), r1 = row + 1 < 0 ? 0 : (row + 1 > H - 1 ? H - 1 : row + 1);
    const int c0 = col < 0 ? 0 : (col > W - 1 ? W - 1 : col), c1 = col + 1 < 0 ? 0 : (col + 1 > W - 1 ? W - 1 : col + 1);
    const unsigned o0 = (unsigned)r0 * (unsigned)W, o1 = (unsigned)r1 * (unsigned)W;
    const char* base = (const char*)flow;
    PcTaps t;
    if (PAIR && W >= 2) {
        const int cb = c0 < W - 1 ? c0 : W - 2;
        const PcF4 a = pc_load_pair(base + (o0 + (unsigned)cb) * 8u), b = pc_load_pair(base + (o1 + (unsigned)cb) * 8u);
        const bool l0 = c0 == cb, l1 = c1 == cb;
        t.p00.x = l0 ? a.x0 : a.x1; t.p00.y = l0 ? a.y0 : a.y1; t.p01.x = l1 ? a.x0 : a.x1; t.p01.y = l1 ? a.y0 : a.y1;
        t.p10.x = l0 ? b.x0 : b.x1; t.p10.y = l0 ? b.y0 : b.y1; t.p11.x = l1 ? b.x0 : b.x1; t.p11.y = l1 ? b.y0 : b.y1;
        return t;
    }
    t.p00 = *(const PcF2*)(base + (o0 + (unsigned)c0) * 8u); t.p01 = *(const PcF2*)(base + (o0 + (unsigned)c1) * 8u);
    t.p10 = *(const PcF2*)(base + (o1 + (unsigned)c0) * 8u); t.p11 = *(const PcF2*)(base + (o1 + (unsigned)c1) * 8u);
    return t;
}

PC_HD void pc_core_eval_taps(const PcTaps& t, const double x[4], double r1x, double r1y, double r2x, double r2y,
                             double s, double r[6], double j[4])
{
    int row, col;
    pc_core_cell(x, row, col);
    const double tc = x[0] - (double)col, tr = x[1] - (double)row;
    double f[2], dr[2], dc[2];
    {
        const double a00 = t.p00.x, a10 = t.p10.x, d0 = (double)t.p01.x - a00, d1 = (double)t.p11.x - a10;
        const double f0 = fma(tc, d0, a00), f1 = fma(tc, d1, a10);
        dr[0] = f1 - f0;
        f[0] = fma(tr, dr[0], f0);
        dc[0] = fma(tr, d1 - d0, d0);
    }
    {
        const double a00 = t.p00.y, a10 = t.p10.y, d0 = (double)t.p01.y - a00, d1 = (double)t.p11.y - a10;
        const double f0 = fma(tc, d0, a00), f1 = fma(tc, d1, a10);
        dr[1] = f1 - f0;
        f[1] = fma(tr, dr[1], f0);
        dc[1] = fma(tr, d1 - d0, d0);
    }
    r[0] = x[0] - r1x;
    r[1] = x[1] - r1y;
    r[2] = (x[2] - r2x) * s;
    r[3] = (x[3] - r2y) * s;
    r[4] = (x[2] - x[0]) - f[0];
    r[5] = (x[3] - x[1]) - f[1];
    j[0] = -1.0 - dc[0];
    j[1] = -dr[0];
    j[2] = -dc[1];
    j[3] = -1.0 - dr[1];
}

// PAIR: the taps as two 16-byte loads (the launch chain, whose tracks are scattered); the fused solve keeps four 8-byte
// loads -- its lanes are neighbours in the image, and the frame kernel has no registers to spare for aligned quads.
template <bool PAIR = false>
PC_HD void pc_core_eval(const PcF2* flow, int H, int W, const double x[4], double r1x, double r1y, double r2x, double r2y,
                        double s, double r[6], double j[4])
{
    const PcTaps t = pc_core_taps<PAIR>(flow, H, W, x);
    pc_core_eval_taps(t, x, r1x, r1y, r2x, r2y, s, r, j);
}

PC_HD double pc_core_cost(const double r[6])
{
    return 0.5 * fma(r[5], r[5], fma(r[4], r[4], fma(r[3], r[3], fma(r[2], r[2], fma(r[1], r[1], r[0] * r[0])))));
}

// Everything Ceres evaluates AT an iterate x (already evaluated: r, j), for one track: the gradient terms, the dogleg
// diagonal Hh, the steepest-descent direction u = g / Hh and the damped Gauss-Newton step d.  Adds the track's terms to the
// sums of quantities that do not depend on the trust-region radius:
//     SUM_GMAX, SUM_XN2, SUM_G2 = g.u, SUM_JG2 = |J u|^2, SUM_GN2, SUM_DOT = g.d, SUM_FAIL
// and, with QQ, to the two more that price ANY dogleg step dl = a u + b d without another pass over the tracks:
//     v[slot_qud] += (J u).(J d),   v[slot_qdd] += |J d|^2
//     |step|^2 = a^2 G2 + 2ab DOT + b^2 GN2,    (J dl).(r + J dl / 2) = a G2 + b DOT + (a^2 JG2 + 2ab QUD + b^2 QDD) / 2
// ((J u).r = u.g = G2 and (J d).r = d.g = DOT).
struct PcSys { double u[4], d[4], Hh0, Hh1; };

template <bool QQ>
PC_HD void pc_core_system(const double x[4], const double r[6], const double j[4], const PcConst& c, double mu, double iA22,
                          double v[PC_NSUM], PcSys& y, int slot_qud, int slot_qdd)
{
    const double s = c.s;
    // gradient g = J^T r; |x - Plus(x, -g)|_inf (trust_region_minimizer.cc: the gradient tolerance test); |x|^2
    const double g0 = fma(j[2], r[5], fma(j[0], r[4], r[0])), g1 = fma(j[3], r[5], fma(j[1], r[4], r[1]));
    const double g2 = fma(s, r[2], r[4]), g3 = fma(s, r[3], r[5]);
    {
        const double m0 = fabs(x[0] - (x[0] - g0)), m1 = fabs(x[1] - (x[1] - g1));
        const double m2 = fabs(x[2] - (x[2] - g2)), m3 = fabs(x[3] - (x[3] - g3));
        v[SUM_GMAX] = fmax(v[SUM_GMAX], fmax(fmax(m0, m1), fmax(m2, m3)));
        v[SUM_XN2] += fma(x[3], x[3], fma(x[2], x[2], fma(x[1], x[1], x[0] * x[0])));
    }
    // H (upper-left block and the coupling K), the dogleg diagonal
    const double k00 = fma(j[2], j[2], j[0] * j[0]), k11 = fma(j[3], j[3], j[1] * j[1]), k01 = fma(j[2], j[3], j[0] * j[1]);
    const double H00 = 1.0 + k00, H11 = 1.0 + k11;
    const double Hh0 = pc_clamped_diag(H00, c.S0q), Hh1 = pc_clamped_diag(H11, c.S1q), Hh2 = c.H22;
    // steepest-descent direction in the dogleg's metric: u = g / Hh (one reciprocal for both varying columns)
    const double w01 = pc_rcp(Hh0 * Hh1);
    const double u0 = g0 * (Hh1 * w01), u1 = g1 * (Hh0 * w01);
    const double iH2 = pc_rcp(Hh2);
    const double u2 = g2 * iH2, u3 = g3 * iH2;
    v[SUM_G2] += fma(g3, u3, fma(g2, u2, fma(g1, u1, g0 * u0)));
    const double mu2 = s * u2, mu3 = s * u3;
    const double mu4 = fma(j[1], u1, fma(j[0], u0, u2)), mu5 = fma(j[3], u1, fma(j[2], u0, u3));      // J u (rows 0, 1: u0, u1)
    v[SUM_JG2] += fma(mu5, mu5, fma(mu4, mu4, fma(mu3, mu3, fma(mu2, mu2, fma(u1, u1, u0 * u0)))));
    // (H + mu Hh) d = -g through the Schur complement of the lower-right block A22 I (K^T K = [k00 k01 ; k01 k11]):
    //   [ A00 - k00/A22   k01 - k01/A22 ] [d0]   [ -g0 + (j0 g2 + j2 g3)/A22 ]          d2 = (-g2 - j0 d0 - j1 d1) / A22
    //   [ k01 - k01/A22   A11 - k11/A22 ] [d1] = [ -g1 + (j1 g2 + j3 g3)/A22 ],         d3 = (-g3 - j2 d0 - j3 d1) / A22
    const double A00 = fma(mu, Hh0, H00), A11 = fma(mu, Hh1, H11);
    const double S00 = fma(-k00, iA22, A00), S11 = fma(-k11, iA22, A11), S01 = fma(-k01, iA22, k01);
    const double b0 = fma(fma(j[2], g3, j[0] * g2), iA22, -g0), b1 = fma(fma(j[3], g3, j[1] * g2), iA22, -g1);
    const double det = fma(S00, S11, -(S01 * S01));
    const double idet = pc_rcp(det);
    const double d0 = fma(S11, b0, -(S01 * b1)) * idet, d1 = fma(S00, b1, -(S01 * b0)) * idet;
    const double d2 = -fma(j[1], d1, fma(j[0], d0, g2)) * iA22, d3 = -fma(j[3], d1, fma(j[2], d0, g3)) * iA22;
    // what a Cholesky factorisation would have refused: a non-positive pivot, or a step that is not finite
    const bool ok = iA22 > 0.0 && S00 > 0.0 && det > 0.0 && fabs(d0) <= 1.7e308 && fabs(d1) <= 1.7e308 && fabs(d2) <= 1.7e308 &&
                    fabs(d3) <= 1.7e308;
    if (!ok) v[SUM_FAIL] += 1.0;
    v[SUM_GN2] += fma(Hh2, fma(d3, d3, d2 * d2), fma(Hh1 * d1, d1, (Hh0 * d0) * d0));
    v[SUM_DOT] += fma(g3, d3, fma(g2, d2, fma(g1, d1, g0 * d0)));
    if (QQ) {
        const double md2 = s * d2, md3 = s * d3;
        const double md4 = fma(j[1], d1, fma(j[0], d0, d2)), md5 = fma(j[3], d1, fma(j[2], d0, d3));  // J d (rows 0, 1: d0, d1)
        v[slot_qud] += fma(mu5, md5, fma(mu4, md4, fma(mu3, md3, fma(mu2, md2, fma(u1, d1, u0 * d0)))));
        v[slot_qdd] += fma(md5, md5, fma(md4, md4, fma(md3, md3, fma(md2, md2, fma(d1, d1, d0 * d0)))));
    }
    y.u[0] = u0; y.u[1] = u1; y.u[2] = u2; y.u[3] = u3;
    y.d[0] = d0; y.d[1] = d1; y.d[2] = d2; y.d[3] = d3;
    y.Hh0 = Hh0; y.Hh1 = Hh1;
}

// The candidate xp = x + a u + b d of one track (a = 0, b = 1 at compile time with GN) and |x - xp|^2 (from the rounded
// candidate, as Ceres forms it).  PER_TRACK: also the step's own sums SUM_DL2 and SUM_MCC (the fused solve reduces them
// per iteration; the launch chain's control step derives them from the sums at x instead, see pc_core_system).
template <bool GN, bool PER_TRACK>
PC_HD void pc_core_step(const double x[4], const double r[6], const double j[4], const PcConst& c, const PcSys& y, double a, double b,
                        double v[PC_NSUM], double xp[4])
{
    double l0, l1, l2, l3;
    if (GN) { l0 = y.d[0]; l1 = y.d[1]; l2 = y.d[2]; l3 = y.d[3]; }
    else {
        l0 = fma(a, y.u[0], b * y.d[0]); l1 = fma(a, y.u[1], b * y.d[1]);
        l2 = fma(a, y.u[2], b * y.d[2]); l3 = fma(a, y.u[3], b * y.d[3]);
    }
    if (PER_TRACK) {
        const double s = c.s;
        v[SUM_DL2] += fma(c.H22, fma(l3, l3, l2 * l2), fma(y.Hh1 * l1, l1, (y.Hh0 * l0) * l0));
        // model_cost_change = -(J dl)^T (r + J dl / 2); the sum carries its negative (the control step flips the sign)
        const double m2 = s * l2, m3 = s * l3;
        const double m4 = fma(j[1], l1, fma(j[0], l0, l2)), m5 = fma(j[3], l1, fma(j[2], l0, l3));
        double t = l0 * fma(0.5, l0, r[0]);
        t = fma(l1, fma(0.5, l1, r[1]), t);
        t = fma(m2, fma(0.5, m2, r[2]), t);
        t = fma(m3, fma(0.5, m3, r[3]), t);
        t = fma(m4, fma(0.5, m4, r[4]), t);
        t = fma(m5, fma(0.5, m5, r[5]), t);
        v[SUM_MCC] += t;
    }
    xp[0] = x[0] + l0; xp[1] = x[1] + l1; xp[2] = x[2] + l2; xp[3] = x[3] + l3;
    const double e0 = x[0] - xp[0], e1 = x[1] - xp[1], e2 = x[2] - xp[2], e3 = x[3] - xp[3];
    v[SUM_STEP2] += fma(e3, e3, fma(e2, e2, fma(e1, e1, e0 * e0)));
}

// One whole trust-region iteration of one track at x with every per-iteration sum reduced per track (what the fused solve
// speculates, and the form tests/test_pc_core_host.py checks against the scaled-space formulas): all sums but SUM_COST
// (the candidate's cost needs the evaluation at xp), SUM_CNT and SUM_COST0.
template <bool GN>
PC_HD void pc_core_iteration(const double x[4], const double r[6], const double j[4], const PcConst& c, double mu, double iA22,
                             double a, double b, double v[PC_NSUM], double xp[4])
{
    PcSys y;
    pc_core_system<false>(x, r, j, c, mu, iA22, v, y, 0, 0);
    pc_core_step<GN, true>(x, r, j, c, y, a, b, v, xp);
}
