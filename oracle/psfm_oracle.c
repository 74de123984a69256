/*
 * psfm_oracle.c -- CPU restatement of ParticleSfM's point_trajectory hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (particle-sfm_amd/) never links, imports or falls back to it.
 *
 * Plain scalar C written from the reference's semantics (SURVEY.md Appendix
 * A/B).  Every function cites the reference file:line it restates (paths
 * relative to the reference root).  OpenMP spreads loops over INDEPENDENT
 * tracks / pixels / grid points over the host cores (OMP_NUM_THREADS; the
 * reference itself runs Ceres with 8 threads, trajectory_optimize.cpp:79) --
 * results do not depend on the thread count: sums over tracks are taken in
 * fixed chunks (ORC_CHUNK), see orc_optimize_location.
 *
 * Pinning status:
 *   - sampler / flow_check / track / track_optimize orchestration: PINNED against
 *     the reference's own Python (torch-CPU grid_sample, SciPy EDT, NumPy) run
 *     unmodified in the build container; vectors in tests/golden/ (see
 *     tests/golden/make_golden.py).
 *   - orc_optimize_location (the Ceres 2.0.0 trust-region/dogleg loop):
 *     PARITY UNPINNED.  Ceres is a third-party dependency that is absent from
 *     the reference tree and from this image (pinned 2.0.0 in
 *     misc/doc/ceres.md:5); the loop below restates its published algorithm
 *     (trust_region_minimizer.cc, dogleg_strategy.cc,
 *     trust_region_step_evaluator.cc, cubic_interpolation.h Grid2D) as
 *     configured at point_trajectory/optimize/src/trajectory_optimize.cpp:74-79.
 *     What IS checked without Ceres: the cost functor's residuals and
 *     Jacobians (orc_path_consistency_eval) against torch autograd through an
 *     independent float64 restatement of the Catmull-Rom interpolator
 *     (tests/test_pc_eval_autograd.py), and the step rule on linear problems
 *     whose every iterate is derived by hand (tests/test_ceres_hand_cases.py).
 *     What stays unpinned is Ceres' own stopping iterate on real inputs; the
 *     day oracle/_ref builds against a Ceres install, tests/test_ref_ceres.py
 *     runs the same cases through the reference's pybind module.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -- no fused ops except
 * the explicit fmaf() chains that restate ATen's vectorised kernel).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* cap the host threads of the parallel loops (no-op without OpenMP; results do not depend on it) */
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* host threads the parallel loops use (1 when built without OpenMP) */
ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* A-1  bilinear sampler == point_trajectory/trajectory.py:25-37              */
/*      (torch F.grid_sample, bilinear, zeros padding, align_corners=True)    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int x0, y0;           /* north-west tap */
    float nw, ne, sw, se; /* weights */
} orc_taps_t;

/* trajectory.py:29-35: positions are cast to fp32, divided by (W-1)/2 resp.
 * (H-1)/2 (true IEEE division by the fp32-rounded scalar), minus 1; ATen then
 * un-normalises (g+1)*((size-1)/2) and splits into floor + weights. */
static inline void orc_taps_f32(float x32, float y32, int H, int W, orc_taps_t* t)
{
    const float cw = (float)((double)(W - 1) / 2.0);
    const float ch = (float)((double)(H - 1) / 2.0);
    float gx = x32 / cw;
    float gy = y32 / ch;
    gx = gx - 1.0f;
    gy = gy - 1.0f;
    const float ix = (gx + 1.0f) * cw;
    const float iy = (gy + 1.0f) * ch;
    const float fx = floorf(ix), fy = floorf(iy);
    const float w = ix - fx, e = 1.0f - w;
    const float n = iy - fy, s = 1.0f - n;
    t->nw = s * e; t->ne = s * w; t->sw = n * e; t->se = n * w;
    /* clamp before the int conversion so absurd coordinates stay defined;
     * anything outside [-1, size] has all four taps out of bounds anyway */
    float cx = fx < -2.0f ? -2.0f : (fx > (float)W + 1.0f ? (float)W + 1.0f : fx);
    float cy = fy < -2.0f ? -2.0f : (fy > (float)H + 1.0f ? (float)H + 1.0f : fy);
    if (!(fx == fx)) cx = -2.0f; /* NaN -> all taps out of bounds */
    if (!(fy == fy)) cy = -2.0f;
    t->x0 = (int)cx; t->y0 = (int)cy;
}

static inline int orc_inb(int x, int y, int H, int W)
{
    return x >= 0 && x < W && y >= 0 && y < H;
}

/* ATen blend: ((nw_v*nw + ne_v*ne) + sw_v*sw) + se_v*se, contracted to an
 * fma chain by the build (probe-verified bit-exact). */
static inline float orc_blend(float vnw, float vne, float vsw, float vse, const orc_taps_t* t)
{
    return fmaf(vse, t->se, fmaf(vsw, t->sw, fmaf(vne, t->ne, vnw * t->nw)));
}

/* sample an interleaved HWC float map with C channels (C = 1 or 2) */
static inline void orc_sample_hwc(const float* map, int C, int H, int W, const orc_taps_t* t, float* out)
{
    const int x0 = t->x0, y0 = t->y0;
    const int inw = orc_inb(x0, y0, H, W), ine = orc_inb(x0 + 1, y0, H, W);
    const int isw = orc_inb(x0, y0 + 1, H, W), ise = orc_inb(x0 + 1, y0 + 1, H, W);
    for (int c = 0; c < C; ++c) {
        const float vnw = inw ? map[((int64_t)y0 * W + x0) * C + c] : 0.0f;
        const float vne = ine ? map[((int64_t)y0 * W + x0 + 1) * C + c] : 0.0f;
        const float vsw = isw ? map[((int64_t)(y0 + 1) * W + x0) * C + c] : 0.0f;
        const float vse = ise ? map[((int64_t)(y0 + 1) * W + x0 + 1) * C + c] : 0.0f;
        out[c] = orc_blend(vnw, vne, vsw, vse, t);
    }
}

static inline float orc_sample_u8(const uint8_t* map, int H, int W, const orc_taps_t* t)
{
    const int x0 = t->x0, y0 = t->y0;
    const float vnw = orc_inb(x0, y0, H, W) ? (float)(map[(int64_t)y0 * W + x0] != 0) : 0.0f;
    const float vne = orc_inb(x0 + 1, y0, H, W) ? (float)(map[(int64_t)y0 * W + x0 + 1] != 0) : 0.0f;
    const float vsw = orc_inb(x0, y0 + 1, H, W) ? (float)(map[(int64_t)(y0 + 1) * W + x0] != 0) : 0.0f;
    const float vse = orc_inb(x0 + 1, y0 + 1, H, W) ? (float)(map[(int64_t)(y0 + 1) * W + x0 + 1] != 0) : 0.0f;
    return orc_blend(vnw, vne, vsw, vse, t);
}

/* trajectory.py:25-37  grid_sample(data[C,H,W], xy[N,2]) -> [N,C]; here the map
 * is the file-native HWC interleaved layout (the permute at track.py:39 is a
 * view change only). */
ORC_API void orc_grid_sample(const float* map_hwc, int C, int H, int W,
                             const double* xy, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) {
        orc_taps_t t;
        orc_taps_f32((float)xy[2 * i], (float)xy[2 * i + 1], H, W, &t);
        orc_sample_hwc(map_hwc, C, H, W, &t, out + i * C);
    }
}

/* ------------------------------------------------------------------------- */
/* A-2  flow_check == point_trajectory/utils.py:58-105                        */
/* ------------------------------------------------------------------------- */
/* One frame pair.  F, B: (H,W,2) f32.  occ: (H,W) u8 0/1.  err: (H,W) f32 or NULL. */
ORC_API void orc_flow_check(const float* F, const float* B, int H, int W, float thres,
                            uint8_t* occ, float* err)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const int64_t p = (int64_t)y * W + x;
            const float fu = F[2 * p], fv = F[2 * p + 1];
            /* utils.py:73-78: coord + flow in fp32 */
            const float X = (float)x + fu, Y = (float)y + fv;
            orc_taps_t t;
            orc_taps_f32(X, Y, H, W, &t); /* utils.py:79-82 */
            float b[2];
            orc_sample_hwc(B, 2, H, W, &t, b);
            /* utils.py:87: torch.norm(warp + flow, dim=1) == sqrtf(fma(ev,ev,eu*eu)) */
            const float eu = b[0] + fu, ev = b[1] + fv;
            const float e = sqrtf(fmaf(ev, ev, eu * eu));
            /* utils.py:58-68 oob; :88-91 union */
            const int oob = (X < 0.0f) || (X > (float)(W - 1)) || (Y < 0.0f) || (Y > (float)(H - 1));
            occ[p] = (uint8_t)((e > thres) || oob);
            if (err) err[p] = e;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* A-7 / Appendix B   optimize_location                                       */
/*   == point_trajectory/optimize/src/trajectory_optimize.cpp:30-96           */
/*   cost functor  path_consistency_cost.h:42-59                              */
/*   interpolator  linear_interpolation.h:97-123 (+ Ceres Grid2D clamping)    */
/* ------------------------------------------------------------------------- */
typedef struct {
    const float* flow; /* (H,W,2) f32; the pybind force-cast to f64 is exact */
    int H, W;
} orc_grid_t;

/* ceres::Grid2D<double,2>::GetValue: indices clamped to the image */
static inline void orc_grid_get(const orc_grid_t* g, int r, int c, double* f)
{
    const int rr = r < 0 ? 0 : (r > g->H - 1 ? g->H - 1 : r);
    const int cc = c < 0 ? 0 : (c > g->W - 1 ? g->W - 1 : c);
    const float* p = g->flow + ((int64_t)rr * g->W + cc) * 2;
    f[0] = (double)p[0];
    f[1] = (double)p[1];
}

/* linear_interpolation.h:97-123 */
static inline void orc_bilerp(const orc_grid_t* g, double r, double c,
                              double* f, double* dfdr, double* dfdc)
{
    double fr = floor(r), fc = floor(c);
    /* keep the int conversion defined for absurd coordinates */
    if (!(fr > -1.0e9)) fr = -1.0e9; if (fr > 1.0e9) fr = 1.0e9;
    if (!(fc > -1.0e9)) fc = -1.0e9; if (fc > 1.0e9) fc = 1.0e9;
    const int row = (int)fr, col = (int)fc;
    double p00[2], p01[2], p10[2], p11[2];
    orc_grid_get(g, row, col, p00);
    orc_grid_get(g, row, col + 1, p01);
    orc_grid_get(g, row + 1, col, p10);
    orc_grid_get(g, row + 1, col + 1, p11);
    const double tc = c - (double)col, tr = r - (double)row;
    for (int k = 0; k < 2; ++k) {
        const double f0 = (1.0 - tc) * p00[k] + tc * p01[k];
        const double f1 = (1.0 - tc) * p10[k] + tc * p11[k];
        const double d0 = p01[k] - p00[k];
        const double d1 = p11[k] - p10[k];
        f[k] = (1.0 - tr) * f0 + tr * f1;
        if (dfdr) dfdr[k] = f1 - f0;
        if (dfdc) dfdc[k] = (1.0 - tr) * d0 + tr * d1;
    }
}

/* residuals (path_consistency_cost.h:50-57) and the 4 non-trivial Jacobian
 * entries the Jet evaluation produces:
 *   row4 = [-1-dfdc_u, -dfdr_u, 1, 0], row5 = [-dfdc_v, -1-dfdr_v, 0, 1]
 *   rows0..3 = diag(1,1,s,s).  x = (x1,y1,x2,y2). */
static inline void orc_pc_eval(const orc_grid_t* g, const double* x, const double* ref1,
                               const double* ref2, double s, double* r, double* jac /*[4]: j40,j41,j50,j51 or NULL*/)
{
    double f[2], dr[2], dc[2];
    orc_bilerp(g, x[1], x[0], f, jac ? dr : NULL, jac ? dc : NULL);
    r[0] = x[0] - ref1[0];
    r[1] = x[1] - ref1[1];
    r[2] = (x[2] - ref2[0]) * s;
    r[3] = (x[3] - ref2[1]) * s;
    r[4] = (x[2] - x[0]) - f[0];
    r[5] = (x[3] - x[1]) - f[1];
    if (jac) {
        jac[0] = -1.0 - dc[0]; /* d r4 / d x1 */
        jac[1] = 0.0 - dr[0];  /* d r4 / d y1 */
        jac[2] = 0.0 - dc[1];  /* d r5 / d x1 */
        jac[3] = -1.0 - dr[1]; /* d r5 / d y1 */
    }
}

/* 4x4 SPD solve by Cholesky (what the sparse normal Cholesky does on each
 * independent 4x4 diagonal block).  Returns 0 on success. */
static inline int orc_chol4(const double A[4][4], const double* b, double* y)
{
    double L[4][4];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j <= i; ++j) {
            double sum = A[i][j];
            for (int k = 0; k < j; ++k) sum -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(sum > 0.0)) return 1;
                L[i][i] = sqrt(sum);
            } else {
                L[i][j] = sum / L[j][j];
            }
        }
    }
    double z[4];
    for (int i = 0; i < 4; ++i) {
        double sum = b[i];
        for (int k = 0; k < i; ++k) sum -= L[i][k] * z[k];
        z[i] = sum / L[i][i];
    }
    for (int i = 3; i >= 0; --i) {
        double sum = z[i];
        for (int k = i + 1; k < 4; ++k) sum -= L[k][i] * y[k];
        y[i] = sum / L[i][i];
    }
    return 0;
}

/* scaled 6x4 Jacobian of one track: Js = J * diag(S) */
typedef struct { double a[6][4]; } orc_js_t;

static inline void orc_build_js(const double* jac, double s, const double* S, orc_js_t* J)
{
    memset(J, 0, sizeof(*J));
    J->a[0][0] = 1.0 * S[0];
    J->a[1][1] = 1.0 * S[1];
    J->a[2][2] = s * S[2];
    J->a[3][3] = s * S[3];
    J->a[4][0] = jac[0] * S[0]; J->a[4][1] = jac[1] * S[1]; J->a[4][2] = 1.0 * S[2];
    J->a[5][0] = jac[2] * S[0]; J->a[5][1] = jac[3] * S[1]; J->a[5][3] = 1.0 * S[3];
}

/* The residual blocks of trajectory_optimize.cpp:56-65 evaluated at uv12 the way AutoDiffCostFunction<PathConsistencyError, 6, 4>
 * hands them to Ceres: residuals (n,6), jacobians (n,6,4) row-major.  For tests that hold orc_pc_eval against an independent
 * differentiation of the functor (tests/test_pc_eval_autograd.py). */
ORC_API void orc_path_consistency_eval(const double* uv12, const double* ref1, const double* ref2, const double* scale,
                                       const float* flow12, int64_t n, int W, int H, double* res, double* jac)
{
    orc_grid_t g = { flow12, H, W };
    for (int64_t i = 0; i < n; ++i) {
        double r[6], j[4];
        orc_pc_eval(&g, uv12 + 4 * i, ref1 + 2 * i, ref2 + 2 * i, scale[i], r, j);
        for (int k = 0; k < 6; ++k) res[6 * i + k] = r[k];
        double* J = jac + 24 * i;
        for (int k = 0; k < 24; ++k) J[k] = 0.0;
        J[0] = 1.0; J[5] = 1.0; J[10] = scale[i]; J[15] = scale[i];
        J[16] = j[0]; J[17] = j[1]; J[18] = 1.0;
        J[20] = j[2]; J[21] = j[3]; J[23] = 1.0;
    }
}

#define ORC_CHUNK 2048   /* tracks per summation chunk (see orc_optimize_location) */

/* Track-sharded runs (tests of psfm_dist.connect_sharded: the tracks of ONE sequence split over several processes):
 * every scalar that Ceres forms over ALL residual blocks / parameters -- costs, norms, the model decrease, the
 * linear-solver failure flag -- passes through this hook right after the local loop: vals[0..n) hold this process's
 * part on entry and the value over all processes on return (is_max[i]: combine by max instead of sum).  NULL = the
 * single-process run. */
typedef void (*orc_reduce_fn)(double* vals, int n, const int* is_max, void* user);
static __thread orc_reduce_fn orc_reduce_hook = NULL;
static __thread void* orc_reduce_user = NULL;
#define ORC_REDUCE1(v) do { if (orc_reduce_hook) { double rv_[1] = {(v)}; const int rm_[1] = {0}; orc_reduce_hook(rv_, 1, rm_, orc_reduce_user); (v) = rv_[0]; } } while (0)
#define ORC_REDUCE2(a, b, amax, bmax) do { if (orc_reduce_hook) { double rv_[2] = {(a), (b)}; const int rm_[2] = {(amax), (bmax)}; orc_reduce_hook(rv_, 2, rm_, orc_reduce_user); (a) = rv_[0]; (b) = rv_[1]; } } while (0)

/* solver statistics (optional) */
typedef struct {
    int32_t iterations;       /* trust-region iterations executed (excluding iteration 0) */
    int32_t successful_steps;
    int32_t termination;      /* 0 function tol, 1 parameter tol, 2 gradient tol, 3 max iter, 4 min radius, 5 failure */
    int32_t dogleg_nonGN;     /* iterations whose step was not the pure Gauss-Newton step */
    double initial_cost, final_cost;
} orc_solve_stats_t;

/* ---- the places where this restatement of Ceres 2.0.0's trust-region loop rests on MEMORY of trust_region_minimizer.cc /
 * solver.cc rather than on anything in the reference tree (parity unpinned: DESIGN.md section 3), each behind a switch, so that
 * the day a real Ceres build exists tests/test_ref_ceres.py can say WHICH reading it agrees with.  Defaults = what the oracle
 * (and the device, and oracle/ceres_tr_numpy.py) implement:
 *   ORC_VAR_ITER0_SUCCESSFUL  0: IterationZero() leaves step_is_successful = true, so GradientToleranceReached() is tested
 *                                before the first iteration;                       1: the test is only due behind a real accepted step
 *   ORC_VAR_FTOL_BASE         0: FunctionToleranceReached(): |x_cost - candidate_cost| <= function_tolerance * x_cost;
 *                             1: ... * candidate_cost;                             2: ... * minimum_cost
 *   ORC_VAR_FAILURE_RETURNS   0: after FAILURE the parameters are handed back as they came in (Summary::IsSolutionUsable());
 *                             1: the best iterate seen is written back
 *   ORC_VAR_MIN_COST_TIES     0: `if (x_cost_ < minimum_cost_)` strictly;          1: <=  (a later iterate of equal cost wins) */
enum { ORC_VAR_ITER0_SUCCESSFUL = 0, ORC_VAR_FTOL_BASE = 1, ORC_VAR_FAILURE_RETURNS = 2, ORC_VAR_MIN_COST_TIES = 3, ORC_N_VARIANTS = 4 };
static int orc_variant[ORC_N_VARIANTS] = {0, 0, 0, 0};
ORC_API int orc_set_variant(int key, int value)     /* returns the previous value, -1 for an unknown key */
{
    if (key < 0 || key >= ORC_N_VARIANTS) return -1;
    const int old = orc_variant[key];
    orc_variant[key] = value;
    return old;
}

/* trajectory_optimize.cpp:30-96.  flow12 is the (H,W,2) f32 map (exactly what
 * the reference force-casts to double).  Restates Ceres 2.0.0:
 * TrustRegionMinimizer::Minimize with DoglegStrategy(TRADITIONAL_DOGLEG),
 * jacobi_scaling, SPARSE_NORMAL_CHOLESKY, max_num_iterations=200, defaults
 * otherwise.  All global scalars (cost, norms) run over ALL tracks. */
ORC_API int orc_optimize_location(const double* uv12, const double* ref1, const double* ref2,
                                  const double* scale, const float* flow12, int64_t n, int w, int h,
                                  double* out, orc_solve_stats_t* stats)
{
    orc_solve_stats_t st; memset(&st, 0, sizeof(st));
    if (n <= 0 && !orc_reduce_hook) { if (stats) *stats = st; return 0; }   /* (sharded: a process without tracks still takes part) */
    if (n < 0) n = 0;
    const orc_grid_t grid = { flow12, h, w };
    const int64_t P = 4 * n;

    double* x = (double*)malloc(sizeof(double) * P);     /* current iterate */
    double* xc = (double*)malloc(sizeof(double) * P);    /* candidate */
    double* res = (double*)malloc(sizeof(double) * 6 * n);
    double* jac = (double*)malloc(sizeof(double) * 4 * n);
    double* S = (double*)malloc(sizeof(double) * P);     /* jacobi scaling */
    double* diag = (double*)malloc(sizeof(double) * P);
    double* ghat = (double*)malloc(sizeof(double) * P);  /* scaled gradient */
    double* gn = (double*)malloc(sizeof(double) * P);    /* scaled Gauss-Newton step */
    double* step = (double*)malloc(sizeof(double) * P);
    double* best = (double*)malloc(sizeof(double) * P);  /* parameters_ (user-visible) */
    memcpy(x, uv12, sizeof(double) * P);

    /* solver constants (Ceres 2.0.0 defaults unless set at trajectory_optimize.cpp:74-79) */
    const int max_iter = 200;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_radius = 1e-32;
    const double min_diag = 1e-6, max_diag = 1e32;
    const double min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
    const int max_invalid = 5;
    double radius = 1e4, mu = min_mu, dogleg_step_norm = 0.0, alpha = 0.0;
    int reuse = 0;

    double x_cost = 0.0, gmax = 0.0, x_norm = 0.0;
    double cand_cost = 0.0, model_cost_change = 0.0;
    int n_invalid = 0;

    /* --- EvaluateGradientAndJacobian at iteration 0 (trust_region_minimizer.cc) --- */
    /* Sums over all tracks are taken over chunks of ORC_CHUNK tracks, in track order inside a chunk, and the chunk sums
     * are added in chunk order: the same bits whatever the number of threads, and exactly the plain sequential sum for
     * n <= ORC_CHUNK (every committed golden vector).  OpenMP only spreads the chunks over the host cores -- the
     * counterpart of the reference's solver_options.num_threads = 8 (trajectory_optimize.cpp:79). */
    const int64_t n_chunks = (n + ORC_CHUNK - 1) / ORC_CHUNK;
    double* part = (double*)malloc(sizeof(double) * 2 * (size_t)(n_chunks > 0 ? n_chunks : 1));
#define EVAL_AT_X(first)                                                              \
    do {                                                                              \
        _Pragma("omp parallel for schedule(static)")                                  \
        for (int64_t ch_ = 0; ch_ < n_chunks; ++ch_) {                                \
        double cs = 0.0, gm = 0.0;                                                    \
        const int64_t i1_ = (ch_ + 1) * ORC_CHUNK < n ? (ch_ + 1) * ORC_CHUNK : n;    \
        for (int64_t i = ch_ * ORC_CHUNK; i < i1_; ++i) {                             \
            const double s_ = scale[i];                                               \
            double* r_ = res + 6 * i; double* j_ = jac + 4 * i;                       \
            orc_pc_eval(&grid, x + 4 * i, ref1 + 2 * i, ref2 + 2 * i, s_, r_, j_);    \
            double ss = 0.0;                                                          \
            for (int k = 0; k < 6; ++k) ss += r_[k] * r_[k];                          \
            cs += 0.5 * ss;                                                           \
            /* unscaled gradient g = J^T r */                                         \
            double g[4];                                                              \
            g[0] = (r_[0] + j_[0] * r_[4]) + j_[2] * r_[5];                           \
            g[1] = (r_[1] + j_[1] * r_[4]) + j_[3] * r_[5];                           \
            g[2] = s_ * r_[2] + r_[4];                                                \
            g[3] = s_ * r_[3] + r_[5];                                                \
            if (first) {                                                              \
                /* jacobian_scaling = 1/(1+sqrt(colnorm^2)), computed ONCE */         \
                const double c0 = (1.0 + j_[0] * j_[0]) + j_[2] * j_[2];              \
                const double c1 = (1.0 + j_[1] * j_[1]) + j_[3] * j_[3];              \
                const double c2 = s_ * s_ + 1.0;                                      \
                S[4 * i + 0] = 1.0 / (1.0 + sqrt(c0));                                \
                S[4 * i + 1] = 1.0 / (1.0 + sqrt(c1));                                \
                S[4 * i + 2] = 1.0 / (1.0 + sqrt(c2));                                \
                S[4 * i + 3] = 1.0 / (1.0 + sqrt(c2));                                \
            }                                                                         \
            /* gradient_max_norm = |x - Plus(x, -g)|_inf */                           \
            for (int k = 0; k < 4; ++k) {                                             \
                const double xv = x[4 * i + k];                                       \
                const double d_ = fabs(xv - (xv + (-g[k])));                          \
                if (d_ > gm) gm = d_;                                                 \
            }                                                                         \
        }                                                                             \
        part[2 * ch_] = cs; part[2 * ch_ + 1] = gm;                                   \
        }                                                                             \
        double cs_ = 0.0, gm_ = 0.0;                                                  \
        for (int64_t ch_ = 0; ch_ < n_chunks; ++ch_) {                                \
            cs_ += part[2 * ch_]; if (part[2 * ch_ + 1] > gm_) gm_ = part[2 * ch_ + 1]; \
        }                                                                             \
        ORC_REDUCE2(cs_, gm_, 0, 1);                                                  \
        x_cost = cs_; gmax = gm_;                                                     \
    } while (0)

#define NORM_OF(v, outv)                                                              \
    do {                                                                              \
        _Pragma("omp parallel for schedule(static)")                                  \
        for (int64_t ch_ = 0; ch_ < n_chunks; ++ch_) {                                \
            double a_ = 0.0;                                                          \
            const int64_t q1_ = 4 * ((ch_ + 1) * ORC_CHUNK < n ? (ch_ + 1) * ORC_CHUNK : n); \
            for (int64_t q = 4 * ch_ * ORC_CHUNK; q < q1_; ++q) a_ += (v)[q] * (v)[q]; \
            part[ch_] = a_;                                                           \
        }                                                                             \
        double t_ = 0.0;                                                              \
        for (int64_t ch_ = 0; ch_ < n_chunks; ++ch_) t_ += part[ch_];                 \
        ORC_REDUCE1(t_);                                                              \
        (outv) = sqrt(t_);                                                            \
    } while (0)

    NORM_OF(x, x_norm);            /* Init(): x_norm_ = x_.norm() */
    EVAL_AT_X(1);                  /* IterationZero() */
    st.initial_cost = x_cost;
    memcpy(best, x, sizeof(double) * P);
    double minimum_cost = x_cost;
    int iteration = 0;
    int step_successful = orc_variant[ORC_VAR_ITER0_SUCCESSFUL] == 0;       /* iteration 0 counts as successful (default) */
    st.termination = 3;
    /* IterationZero(): EvaluateGradientAndJacobian fails when a residual (or Jacobian entry) of ANY block is not finite
     * (residual_block.cc ResidualBlock::Evaluate -> IsEvaluationValid / IsArrayValid) -- "Residual and Jacobian evaluation
     * failed.", termination FAILURE before the first iteration.  The cost is finite iff every residual is (a Jacobian entry
     * here is a difference of the taps the residual was blended from).  Evaluations of CANDIDATES that fail are steps of
     * infinite cost instead (ComputeCandidatePointAndEvaluateCost): rejected below by the NaN comparisons. */
    const int eval0_failed = !isfinite(x_cost);
    if (eval0_failed) st.termination = 5;

    while (!eval0_failed) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (step_successful && (x_cost < minimum_cost ||       /* trust_region_minimizer.cc: `if (x_cost_ < minimum_cost_)` -- strictly */
                                (orc_variant[ORC_VAR_MIN_COST_TIES] && x_cost == minimum_cost))) {
            minimum_cost = x_cost;
            memcpy(best, x, sizeof(double) * P);
        }
        if (iteration >= max_iter) { st.termination = 3; break; }
        if (step_successful && gmax <= gradient_tolerance) { st.termination = 2; break; }
        if (radius <= min_radius) { st.termination = 4; break; }

        ++iteration;
        st.iterations = iteration;
        step_successful = 0;

        /* ---- DoglegStrategy::ComputeStep ---- */
        int lin_fail = 0;
        if (!reuse) {
            reuse = 1;
            double g2 = 0.0, jg2 = 0.0;
            /* diagonal, gradient, Cauchy point (alpha) */
#pragma omp parallel for schedule(static)
            for (int64_t ch = 0; ch < n_chunks; ++ch) {
            double g2c = 0.0, jg2c = 0.0;
            const int64_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
            for (int64_t i = ch * ORC_CHUNK; i < i1; ++i) {
                orc_js_t J; orc_build_js(jac + 4 * i, scale[i], S + 4 * i, &J);
                const double* r_ = res + 6 * i;
                double d[4], gt[4], sg[4];
                for (int c = 0; c < 4; ++c) {
                    double cn = 0.0, gr = 0.0;
                    for (int q = 0; q < 6; ++q) { cn += J.a[q][c] * J.a[q][c]; gr += J.a[q][c] * r_[q]; }
                    cn = cn < min_diag ? min_diag : (cn > max_diag ? max_diag : cn);
                    d[c] = sqrt(cn);
                    gt[c] = gr / d[c];
                    sg[c] = gt[c] / d[c];
                    diag[4 * i + c] = d[c];
                    ghat[4 * i + c] = gt[c];
                    g2c += gt[c] * gt[c];
                }
                for (int q = 0; q < 6; ++q) {
                    double v = 0.0;
                    for (int c = 0; c < 4; ++c) v += J.a[q][c] * sg[c];
                    jg2c += v * v;
                }
            }
            part[2 * ch] = g2c; part[2 * ch + 1] = jg2c;
            }
            for (int64_t ch = 0; ch < n_chunks; ++ch) { g2 += part[2 * ch]; jg2 += part[2 * ch + 1]; }
            ORC_REDUCE2(g2, jg2, 0, 0);
            alpha = g2 / jg2;
            /* ComputeGaussNewtonStep: (Js^T Js + mu*diag^2) y = Js^T r, retry with mu*=10 */
            lin_fail = 1;
            while (mu < max_mu) {
                int fail = 0;
#pragma omp parallel for schedule(static) reduction(|:fail)
                for (int64_t ch = 0; ch < n_chunks; ++ch) {
                const int64_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
                for (int64_t i = ch * ORC_CHUNK; i < i1 && !fail; ++i) {
                    orc_js_t J; orc_build_js(jac + 4 * i, scale[i], S + 4 * i, &J);
                    const double* r_ = res + 6 * i;
                    double A[4][4], b[4], y[4];
                    const double smu = sqrt(mu);
                    for (int a = 0; a < 4; ++a) {
                        for (int c = 0; c <= a; ++c) {
                            double v = 0.0;
                            for (int q = 0; q < 6; ++q) v += J.a[q][a] * J.a[q][c];
                            A[a][c] = v; A[c][a] = v;
                        }
                        const double D = diag[4 * i + a] * smu; /* lm_diagonal */
                        A[a][a] += D * D;
                        double br = 0.0;
                        for (int q = 0; q < 6; ++q) br += J.a[q][a] * r_[q];
                        b[a] = br;
                    }
                    if (orc_chol4(A, b, y)) { fail = 1; break; }
                    for (int c = 0; c < 4; ++c) {
                        if (!(y[c] == y[c]) || isinf(y[c])) fail = 1;
                        gn[4 * i + c] = y[c] * (-diag[4 * i + c]); /* gauss_newton_step *= -diagonal */
                    }
                }
                }
                { double ff = (double)fail; if (orc_reduce_hook) { const int m_[1] = {1}; orc_reduce_hook(&ff, 1, m_, orc_reduce_user); } fail = ff != 0.0; }
                if (fail) { mu *= mu_increase; continue; }
                lin_fail = 0;
                break;
            }
        }
        int step_valid = 0;
        if (!lin_fail) {
            /* ComputeTraditionalDoglegStep */
            double gnorm, gnn;
            NORM_OF(ghat, gnorm);
            NORM_OF(gn, gnn);
            if (gnn <= radius) {
#pragma omp parallel for schedule(static)
                for (int64_t q = 0; q < P; ++q) step[q] = gn[q] / diag[q];
                dogleg_step_norm = gnn;
            } else if (gnorm * alpha >= radius) {
                for (int64_t q = 0; q < P; ++q) step[q] = (-(radius / gnorm) * ghat[q]) / diag[q];
                dogleg_step_norm = radius;
                st.dogleg_nonGN++;
            } else {
                double dot = 0.0;
                for (int64_t q = 0; q < P; ++q) dot += ghat[q] * gn[q];
                ORC_REDUCE1(dot);
                const double b_dot_a = -alpha * dot;
                const double a2 = pow(alpha * gnorm, 2.0);
                const double bma2 = a2 - 2 * b_dot_a + pow(gnn, 2);
                const double c = b_dot_a - a2;
                const double d = sqrt(c * c + bma2 * (pow(radius, 2.0) - a2));
                const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
                double sn = 0.0;
                for (int64_t q = 0; q < P; ++q) {
                    const double v = (-alpha * (1.0 - beta)) * ghat[q] + beta * gn[q];
                    sn += v * v;
                    step[q] = v / diag[q];
                }
                ORC_REDUCE1(sn);
                dogleg_step_norm = sqrt(sn);
                st.dogleg_nonGN++;
            }
            /* model_cost_change = -(J*step)'(f + J*step/2)   (ComputeTrustRegionStep) */
            double mcc = 0.0;
#pragma omp parallel for schedule(static)
            for (int64_t ch = 0; ch < n_chunks; ++ch) {
            double mc = 0.0;
            const int64_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
            for (int64_t i = ch * ORC_CHUNK; i < i1; ++i) {
                orc_js_t J; orc_build_js(jac + 4 * i, scale[i], S + 4 * i, &J);
                const double* r_ = res + 6 * i;
                for (int q = 0; q < 6; ++q) {
                    double m = 0.0;
                    for (int c = 0; c < 4; ++c) m += J.a[q][c] * step[4 * i + c];
                    mc += m * (r_[q] + m / 2.0);
                }
            }
            part[ch] = mc;
            }
            for (int64_t ch = 0; ch < n_chunks; ++ch) mcc += part[ch];
            ORC_REDUCE1(mcc);
            model_cost_change = -mcc;
            step_valid = model_cost_change > 0.0;
        }
        if (!step_valid) {
            /* HandleInvalidStep */
            if (++n_invalid >= max_invalid) { st.termination = 5; break; }
            mu *= mu_increase; reuse = 0; /* StepIsInvalid */
            continue;
        }
        n_invalid = 0;
        /* delta = step .* jacobian_scaling ; candidate = x + delta ; cost */
        {
            double cs = 0.0, sn2 = 0.0;
#pragma omp parallel for schedule(static)
            for (int64_t ch = 0; ch < n_chunks; ++ch) {
            double csc = 0.0, snc = 0.0;
            const int64_t i1 = (ch + 1) * ORC_CHUNK < n ? (ch + 1) * ORC_CHUNK : n;
            for (int64_t i = ch * ORC_CHUNK; i < i1; ++i) {
                for (int c = 0; c < 4; ++c) {
                    const double delta = step[4 * i + c] * S[4 * i + c];
                    xc[4 * i + c] = x[4 * i + c] + delta;
                    const double dd = x[4 * i + c] - xc[4 * i + c];
                    snc += dd * dd;
                }
                double r_[6];
                orc_pc_eval(&grid, xc + 4 * i, ref1 + 2 * i, ref2 + 2 * i, scale[i], r_, NULL);
                double ss = 0.0;
                for (int k = 0; k < 6; ++k) ss += r_[k] * r_[k];
                csc += 0.5 * ss;
            }
            part[2 * ch] = csc; part[2 * ch + 1] = snc;
            }
            for (int64_t ch = 0; ch < n_chunks; ++ch) { cs += part[2 * ch]; sn2 += part[2 * ch + 1]; }
            ORC_REDUCE2(cs, sn2, 0, 0);
            cand_cost = cs;
            /* ParameterToleranceReached */
            const double step_norm = sqrt(sn2);
            if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { st.termination = 1; break; }
        }
        /* FunctionToleranceReached */
        {
            const double ftol_base = orc_variant[ORC_VAR_FTOL_BASE] == 1 ? cand_cost : (orc_variant[ORC_VAR_FTOL_BASE] == 2 ? minimum_cost : x_cost);
            if (fabs(x_cost - cand_cost) <= function_tolerance * ftol_base) { st.termination = 0; break; }
        }
        /* IsStepSuccessful (monotonic step evaluator) */
        const double rho = (x_cost - cand_cost) / model_cost_change;
        if (rho > min_relative_decrease) {
            /* HandleSuccessfulStep */
            memcpy(x, xc, sizeof(double) * P);
            NORM_OF(x, x_norm);
            EVAL_AT_X(0);
            step_successful = 1;
            st.successful_steps++;
            /* DoglegStrategy::StepAccepted */
            if (rho < 0.25) radius *= 0.5;
            if (rho > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
            mu = fmax(min_mu, 2.0 * mu / mu_increase);
            reuse = 0;
        } else {
            /* HandleUnsuccessfulStep -> DoglegStrategy::StepRejected */
            radius *= 0.5;
            reuse = 1;
        }
    }
    st.final_cost = minimum_cost;
    /* solver.cc Minimize(): the user's parameters are only updated when Summary::IsSolutionUsable() -- after a FAILURE
     * Ceres restores the values it was called with; the reference ignores the failure (trajectory_optimize.cpp:81-82) */
    memcpy(out, (st.termination == 5 && orc_variant[ORC_VAR_FAILURE_RETURNS] == 0) ? uv12 : best, sizeof(double) * P);
    if (stats) *stats = st;
    free(x); free(xc); free(res); free(jac); free(S); free(diag); free(ghat); free(gn); free(step); free(best); free(part);
    return st.termination == 5 ? 1 : 0;
#undef EVAL_AT_X
#undef NORM_OF
}

/* ------------------------------------------------------------------------- */
/* track / track_optimize                                                     */
/*   == point_trajectory/track.py:24-50, track_optimize.py:24-53,             */
/*      trajectory.py:45-62 (step_forward), :98-194 (IncrementalTrajectorySet)*/
/*      optimize/src/trajectory_base.cpp:21-93 (Trajectory)                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t birth;     /* first time */
    int32_t len;       /* number of positions (xys + buffer) */
    int32_t cap;
    double* pts;       /* len x 2; the last min(len,buffer_size) entries are the buffer deque */
} orc_traj_t;

typedef struct {
    orc_traj_t** v; int64_t n, cap;
} orc_list_t;

static void orc_list_push(orc_list_t* l, orc_traj_t* t)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 1024; l->v = (orc_traj_t**)realloc(l->v, sizeof(void*) * l->cap); }
    l->v[l->n++] = t;
}

/* Trajectory(time, xy, buffer_size) + extend  (trajectory_base.cpp:21-24,55-67) */
static void orc_traj_extend(orc_traj_t* t, double x, double y)
{
    if (t->len == t->cap) { t->cap = t->cap ? t->cap * 2 : 8; t->pts = (double*)realloc(t->pts, sizeof(double) * 2 * t->cap); }
    t->pts[2 * t->len] = x; t->pts[2 * t->len + 1] = y; t->len++;
}

typedef struct {
    int64_t n_traj, n_points;
    int32_t* birth;   /* n_traj */
    int32_t* len;     /* n_traj */
    int64_t* off;     /* n_traj+1 */
    double* xy;       /* n_points x 2 */
    /* optional per-solve statistics (track_optimize only) */
    int32_t n_solves;
    orc_solve_stats_t* solves;
} orc_result_t;

ORC_API void orc_result_free(orc_result_t* r)
{
    if (!r) return;
    free(r->birth); free(r->len); free(r->off); free(r->xy); free(r->solves); free(r);
}

/* flows: n_flows pointers to (H,W,2) f32; occ: n_flows pointers to (H,W) u8.
 * flows_f2 / occ_s2: stride-2 stacks or NULL (-> track()).  The result lists
 * ALL trajectories in full_trajs order (the index is the saved id,
 * main_connect_point_trajectories.py:56-60); no min-length filter here. */
ORC_API orc_result_t* orc_track(const float* const* flows, const uint8_t* const* occ,
                                const float* const* flows_f2, const uint8_t* const* occ_s2,
                                int n_flows, int H, int W, int ratio)
{
    const int optimize = (flows_f2 != NULL);
    const int buffer_size = optimize ? 3 : 0;      /* track_optimize.py:30 / track.py:30 */
    const int GW = (W + ratio - 1) / ratio, GH = (H + ratio - 1) / ratio; /* trajectory.py:110-115 */
    const int64_t G = (int64_t)GW * GH;
    orc_list_t active = {0}, next_active = {0}, full = {0};
    uint8_t* cand = (uint8_t*)malloc(G);           /* sample_candidates as a grid mask */
    memset(cand, 1, G);                            /* trajectory.py:108: all candidates at start */
    uint8_t* occupied = (uint8_t*)malloc((size_t)H * W);
    orc_result_t* R = (orc_result_t*)calloc(1, sizeof(orc_result_t));
    if (optimize) R->solves = (orc_solve_stats_t*)calloc(n_flows > 0 ? n_flows : 1, sizeof(orc_solve_stats_t));
    double* nxt = NULL; uint8_t* flag = NULL; int64_t scratch_cap = 0;

    for (int f = 0; f < n_flows; ++f) {
        /* new_traj_all (trajectory.py:117-120): births in row-major grid order, time = f */
        for (int64_t g = 0; g < G; ++g) {
            if (!cand[g]) continue;
            orc_traj_t* t = (orc_traj_t*)calloc(1, sizeof(orc_traj_t));
            t->birth = f;
            orc_traj_extend(t, (double)((g % GW) * ratio), (double)((g / GW) * ratio));
            orc_list_push(&active, t);
        }
        const int64_t A = active.n;
        if (A > scratch_cap) { scratch_cap = A; nxt = (double*)realloc(nxt, sizeof(double) * 2 * A); flag = (uint8_t*)realloc(flag, A); }
        /* get_cur_pos + grid_sample(flow) + step_forward  (track.py:38-46, trajectory.py:45-62) */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < A; ++i) {
            const orc_traj_t* t = active.v[i];
            const double px = t->pts[2 * (t->len - 1)], py = t->pts[2 * (t->len - 1) + 1]; /* get_tail_location */
            orc_taps_t tp;
            orc_taps_f32((float)px, (float)py, H, W, &tp);
            float fl[2];
            orc_sample_hwc(flows[f], 2, H, W, &tp, fl);
            const float oc = orc_sample_u8(occ[f], H, W, &tp);
            const int occ_c = oc > 0.1f;                               /* trajectory.py:50 */
            const double nx = px + (double)fl[0], ny = py + (double)fl[1]; /* :55 */
            const int valid = (nx > 0) && (nx < (double)(W - 1)) && (ny > 0) && (ny < (double)(H - 1)); /* :56-57 */
            nxt[2 * i] = nx; nxt[2 * i + 1] = ny;
            flag[i] = (uint8_t)(valid && !occ_c);                      /* :61 */
        }
        /* extend_all (trajectory.py:129-152) */
        memset(occupied, 0, (size_t)H * W);
        int64_t n_occ = 0;
        next_active.n = 0;
        for (int64_t i = 0; i < A; ++i) {
            orc_traj_t* t = active.v[i];
            if (!flag[i]) {
                orc_list_push(&full, t);                               /* clear_buffer + append */
            } else {
                occupied[(int64_t)((int)nxt[2 * i + 1]) * W + (int)nxt[2 * i]] = 1;
                n_occ++;
                orc_traj_extend(t, nxt[2 * i], nxt[2 * i + 1]);
                orc_list_push(&next_active, t);
            }
        }
        { orc_list_t tmp = active; active = next_active; next_active = tmp; }
        /* respawn candidates: distance_transform_edt(1-occupied) > ratio on the stride grid
         * == no occupied pixel with dx^2+dy^2 <= ratio^2 (SURVEY A-5).  With no occupied
         * pixel at all SciPy measures to a phantom feature at (y=-1,x=0): everything
         * but grid point (0,0) respawns. */
#pragma omp parallel for schedule(static)
        for (int gy = 0; gy < GH; ++gy) {
            for (int gx = 0; gx < GW; ++gx) {
                const int cx = gx * ratio, cy = gy * ratio;
                int free_ = 1;
                if (n_occ == 0) {
                    free_ = !((cy + 1) * (cy + 1) + cx * cx <= ratio * ratio);
                } else {
                    for (int dy = -ratio; dy <= ratio && free_; ++dy) {
                        const int yy = cy + dy;
                        if (yy < 0 || yy >= H) continue;
                        for (int dx = -ratio; dx <= ratio; ++dx) {
                            const int xx = cx + dx;
                            if (xx < 0 || xx >= W) continue;
                            if (dx * dx + dy * dy > ratio * ratio) continue;
                            if (occupied[(int64_t)yy * W + xx]) { free_ = 0; break; }
                        }
                    }
                }
                cand[(int64_t)gy * GW + gx] = (uint8_t)free_;
            }
        }
        /* optimize_buffer (track_optimize.py:49-50, trajectory.py:161-194) */
        if (optimize && f + 1 >= 2) {
            int64_t N = 0;
            for (int64_t i = 0; i < active.n; ++i) if (active.v[i]->len >= buffer_size) N++;
            if (N > 0) { /* the reference raises on N == 0 (np.stack([])): we skip the solve */
                double* uv12 = (double*)malloc(sizeof(double) * 4 * N);
                double* r1 = (double*)malloc(sizeof(double) * 2 * N);
                double* r2 = (double*)malloc(sizeof(double) * 2 * N);
                double* sc = (double*)malloc(sizeof(double) * N);
                double* o = (double*)malloc(sizeof(double) * 4 * N);
                /* row k of the batch = the k-th active track with a full buffer (trajectory.py:165-170) */
                int64_t* row = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
                { int64_t k = 0; for (int64_t i = 0; i < active.n; ++i) if (active.v[i]->len >= buffer_size) row[k++] = i; }
#pragma omp parallel for schedule(static)
                for (int64_t k = 0; k < N; ++k) {
                    const orc_traj_t* t = active.v[row[k]];
                    const double* b0 = t->pts + 2 * (t->len - 3);
                    orc_taps_t tp;
                    orc_taps_f32((float)b0[0], (float)b0[1], H, W, &tp);
                    float f01[2], f02[2];
                    orc_sample_hwc(flows[f - 1], 2, H, W, &tp, f01);       /* trajectory.py:173-176 */
                    orc_sample_hwc(flows_f2[f - 1], 2, H, W, &tp, f02);
                    const float o02 = orc_sample_u8(occ_s2[f - 1], H, W, &tp); /* :177-178 */
                    /* :179  (1.0 - occ02) * (norm(flow02) < 20): all fp32, numpy norm = sqrt(u*u+v*v) */
                    const float nrm = sqrtf(f02[0] * f02[0] + f02[1] * f02[1]);
                    const float s = (1.0f - o02) * (nrm < 20.0f ? 1.0f : 0.0f);
                    r1[2 * k] = b0[0] + (double)f01[0]; r1[2 * k + 1] = b0[1] + (double)f01[1]; /* :182 */
                    r2[2 * k] = b0[0] + (double)f02[0]; r2[2 * k + 1] = b0[1] + (double)f02[1]; /* :183 */
                    sc[k] = (double)s;
                    uv12[4 * k] = b0[2]; uv12[4 * k + 1] = b0[3]; uv12[4 * k + 2] = b0[4]; uv12[4 * k + 3] = b0[5];
                }
                orc_optimize_location(uv12, r1, r2, sc, flows[f], N, W, H, o, &R->solves[R->n_solves]);
                R->n_solves++;
#pragma omp parallel for schedule(static)
                for (int64_t k = 0; k < N; ++k) {                           /* :190-194 set_buffer_xy(1|2) */
                    orc_traj_t* t = active.v[row[k]];
                    double* b0 = t->pts + 2 * (t->len - 3);
                    b0[2] = o[4 * k]; b0[3] = o[4 * k + 1]; b0[4] = o[4 * k + 2]; b0[5] = o[4 * k + 3];
                }
                free(row);
                free(uv12); free(r1); free(r2); free(sc); free(o);
            }
        }
    }
    /* clear_active (trajectory.py:154-158) */
    for (int64_t i = 0; i < active.n; ++i) orc_list_push(&full, active.v[i]);

    R->n_traj = full.n;
    R->birth = (int32_t*)malloc(sizeof(int32_t) * (full.n ? full.n : 1));
    R->len = (int32_t*)malloc(sizeof(int32_t) * (full.n ? full.n : 1));
    R->off = (int64_t*)malloc(sizeof(int64_t) * (full.n + 1));
    int64_t tot = 0;
    for (int64_t i = 0; i < full.n; ++i) { R->off[i] = tot; tot += full.v[i]->len; }
    R->off[full.n] = tot;
    R->n_points = tot;
    R->xy = (double*)malloc(sizeof(double) * 2 * (tot ? tot : 1));
    for (int64_t i = 0; i < full.n; ++i) {
        orc_traj_t* t = full.v[i];
        R->birth[i] = t->birth; R->len[i] = t->len;
        memcpy(R->xy + 2 * R->off[i], t->pts, sizeof(double) * 2 * t->len);
        free(t->pts); free(t);
    }
    free(active.v); free(next_active.v); free(full.v); free(cand); free(occupied); free(nxt); free(flag);
    return R;
}

/* ------------------------------------------------------------------------- */
/* Track-sharded stepping API (tests of psfm_dist.connect_sharded).           */
/* The tracks of ONE sequence are split over processes by the row band of the */
/* grid point they were born on; a process keeps the reference's active list  */
/* for ITS tracks and performs, frame by frame, exactly the steps of orc_track */
/* above -- what crosses processes is (a) which stride-r grid points lie       */
/* within distance r of a surviving track's pixel (the EDT respawn rule,       */
/* trajectory.py:150-152, at grid resolution) + whether any track survived,    */
/* and (b) the global scalars of the solve (orc_reduce_hook).                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int H, W, ratio, GW, GH, n_flows, optimize, buffer_size;
    int64_t G, g0, g1;               /* this process owns the births on grid points [g0, g1) (whole rows) */
    orc_list_t active, next_active, full;
    uint8_t* cand;                   /* G: respawn candidates for the next frame (only [g0,g1) is used) */
    orc_solve_stats_t* solves; int32_t n_solves;
    double* nxt; uint8_t* flag; int64_t scratch_cap;
} orc_shard_t;

ORC_API orc_shard_t* orc_shard_begin(int n_flows, int H, int W, int ratio, int64_t g0, int64_t g1, int optimize)
{
    orc_shard_t* s = (orc_shard_t*)calloc(1, sizeof(orc_shard_t));
    s->H = H; s->W = W; s->ratio = ratio; s->n_flows = n_flows; s->optimize = optimize;
    s->buffer_size = optimize ? 3 : 0;
    s->GW = (W + ratio - 1) / ratio; s->GH = (H + ratio - 1) / ratio;
    s->G = (int64_t)s->GW * s->GH; s->g0 = g0; s->g1 = g1;
    s->cand = (uint8_t*)malloc((size_t)s->G);
    memset(s->cand, 1, (size_t)s->G);                      /* trajectory.py:108 */
    if (optimize) s->solves = (orc_solve_stats_t*)calloc(n_flows > 0 ? n_flows : 1, sizeof(orc_solve_stats_t));
    return s;
}

/* births of frame f on the own band + chain step + extend_all for the own tracks (orc_track, same lines).
 * blocked_out[G]: 1 where a grid point has one of THIS process's survivors' pixels within distance ratio (all grid
 * points, not only the own band); *n_alive_out = this process's survivors. */
ORC_API void orc_shard_step(orc_shard_t* s, int f, const float* flow, const uint8_t* occ, uint8_t* blocked_out, int64_t* n_alive_out)
{
    const int H = s->H, W = s->W, ratio = s->ratio, GW = s->GW, GH = s->GH;
    for (int64_t g = s->g0; g < s->g1; ++g) {
        if (!s->cand[g]) continue;
        orc_traj_t* t = (orc_traj_t*)calloc(1, sizeof(orc_traj_t));
        t->birth = f;
        orc_traj_extend(t, (double)((g % GW) * ratio), (double)((g / GW) * ratio));
        orc_list_push(&s->active, t);
    }
    const int64_t A = s->active.n;
    if (A > s->scratch_cap) { s->scratch_cap = A; s->nxt = (double*)realloc(s->nxt, sizeof(double) * 2 * A); s->flag = (uint8_t*)realloc(s->flag, A); }
    double* nxt = s->nxt; uint8_t* flag = s->flag;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < A; ++i) {
        const orc_traj_t* t = s->active.v[i];
        const double px = t->pts[2 * (t->len - 1)], py = t->pts[2 * (t->len - 1) + 1];
        orc_taps_t tp;
        orc_taps_f32((float)px, (float)py, H, W, &tp);
        float fl[2];
        orc_sample_hwc(flow, 2, H, W, &tp, fl);
        const float oc = orc_sample_u8(occ, H, W, &tp);
        const int occ_c = oc > 0.1f;
        const double nx = px + (double)fl[0], ny = py + (double)fl[1];
        const int valid = (nx > 0) && (nx < (double)(W - 1)) && (ny > 0) && (ny < (double)(H - 1));
        nxt[2 * i] = nx; nxt[2 * i + 1] = ny;
        flag[i] = (uint8_t)(valid && !occ_c);
    }
    memset(blocked_out, 0, (size_t)s->G);
    int64_t n_occ = 0;
    s->next_active.n = 0;
    for (int64_t i = 0; i < A; ++i) {
        orc_traj_t* t = s->active.v[i];
        if (!flag[i]) {
            orc_list_push(&s->full, t);
        } else {
            /* occupied[int(y), int(x)] = 1 (trajectory.py:144), then every grid point whose disc of radius ratio holds it */
            const int px = (int)nxt[2 * i], py = (int)nxt[2 * i + 1];
            for (int gy = (py - ratio + ratio - 1) / ratio > 0 ? (py - ratio + ratio - 1) / ratio : 0; gy < GH && gy * ratio <= py + ratio; ++gy)
                for (int gx = (px - ratio + ratio - 1) / ratio > 0 ? (px - ratio + ratio - 1) / ratio : 0; gx < GW && gx * ratio <= px + ratio; ++gx) {
                    const int dx = gx * ratio - px, dy = gy * ratio - py;
                    if (dx * dx + dy * dy <= ratio * ratio) blocked_out[(int64_t)gy * GW + gx] = 1;
                }
            n_occ++;
            orc_traj_extend(t, nxt[2 * i], nxt[2 * i + 1]);
            orc_list_push(&s->next_active, t);
        }
    }
    { orc_list_t tmp = s->active; s->active = s->next_active; s->next_active = tmp; }
    *n_alive_out = n_occ;
}

/* the respawn candidates of the next frame from the maps of ALL processes (OR) and the global survivor count */
ORC_API void orc_shard_set_blocked(orc_shard_t* s, const uint8_t* blocked_global, int64_t n_alive_global)
{
    const int ratio = s->ratio, GW = s->GW;
    for (int64_t g = s->g0; g < s->g1; ++g) {
        const int cx = (int)(g % GW) * ratio, cy = (int)(g / GW) * ratio;
        if (n_alive_global == 0) s->cand[g] = (uint8_t)!((cy + 1) * (cy + 1) + cx * cx <= ratio * ratio);   /* SciPy's phantom feature */
        else s->cand[g] = (uint8_t)!blocked_global[g];
    }
}

/* optimize_buffer of loop index f for the own tracks (orc_track, same lines); the solve's global scalars go through
 * `reduce` (all processes call this together, also those without a track that has a full buffer) */
ORC_API void orc_shard_solve(orc_shard_t* s, int f, const float* flow_prev, const float* flow_cur, const float* flow2_prev,
                             const uint8_t* occ2_prev, orc_reduce_fn reduce, void* user)
{
    const int H = s->H, W = s->W, buffer_size = s->buffer_size;
    int64_t N = 0;
    for (int64_t i = 0; i < s->active.n; ++i) if (s->active.v[i]->len >= buffer_size) N++;
    double nn = (double)N;
    { const int m_[1] = {0}; if (reduce) reduce(&nn, 1, m_, user); }
    if (nn == 0.0) return;                                  /* no track anywhere has a full buffer: no solve (orc_track) */
    double* uv12 = (double*)malloc(sizeof(double) * 4 * (N ? N : 1));
    double* r1 = (double*)malloc(sizeof(double) * 2 * (N ? N : 1));
    double* r2 = (double*)malloc(sizeof(double) * 2 * (N ? N : 1));
    double* sc = (double*)malloc(sizeof(double) * (N ? N : 1));
    double* o = (double*)malloc(sizeof(double) * 4 * (N ? N : 1));
    int64_t* row = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
    { int64_t k = 0; for (int64_t i = 0; i < s->active.n; ++i) if (s->active.v[i]->len >= buffer_size) row[k++] = i; }
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < N; ++k) {
        const orc_traj_t* t = s->active.v[row[k]];
        const double* b0 = t->pts + 2 * (t->len - 3);
        orc_taps_t tp;
        orc_taps_f32((float)b0[0], (float)b0[1], H, W, &tp);
        float f01[2], f02[2];
        orc_sample_hwc(flow_prev, 2, H, W, &tp, f01);
        orc_sample_hwc(flow2_prev, 2, H, W, &tp, f02);
        const float o02 = orc_sample_u8(occ2_prev, H, W, &tp);
        const float nrm = sqrtf(f02[0] * f02[0] + f02[1] * f02[1]);
        const float sf = (1.0f - o02) * (nrm < 20.0f ? 1.0f : 0.0f);
        r1[2 * k] = b0[0] + (double)f01[0]; r1[2 * k + 1] = b0[1] + (double)f01[1];
        r2[2 * k] = b0[0] + (double)f02[0]; r2[2 * k + 1] = b0[1] + (double)f02[1];
        sc[k] = (double)sf;
        uv12[4 * k] = b0[2]; uv12[4 * k + 1] = b0[3]; uv12[4 * k + 2] = b0[4]; uv12[4 * k + 3] = b0[5];
    }
    orc_reduce_hook = reduce; orc_reduce_user = user;
    orc_optimize_location(uv12, r1, r2, sc, flow_cur, N, W, H, o, &s->solves[s->n_solves]);
    orc_reduce_hook = NULL; orc_reduce_user = NULL;
    s->n_solves++;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < N; ++k) {
        orc_traj_t* t = s->active.v[row[k]];
        double* b0 = t->pts + 2 * (t->len - 3);
        b0[2] = o[4 * k]; b0[3] = o[4 * k + 1]; b0[4] = o[4 * k + 2]; b0[5] = o[4 * k + 3];
    }
    free(uv12); free(r1); free(r2); free(sc); free(o); free(row);
}

/* clear_active + the own trajectories in the reference's order (dead ones by frame, then the still active ones) */
ORC_API orc_result_t* orc_shard_finish(orc_shard_t* s)
{
    for (int64_t i = 0; i < s->active.n; ++i) orc_list_push(&s->full, s->active.v[i]);
    orc_result_t* R = (orc_result_t*)calloc(1, sizeof(orc_result_t));
    const orc_list_t full = s->full;
    R->n_traj = full.n;
    R->birth = (int32_t*)malloc(sizeof(int32_t) * (full.n ? full.n : 1));
    R->len = (int32_t*)malloc(sizeof(int32_t) * (full.n ? full.n : 1));
    R->off = (int64_t*)malloc(sizeof(int64_t) * (full.n + 1));
    int64_t tot = 0;
    for (int64_t i = 0; i < full.n; ++i) { R->off[i] = tot; tot += full.v[i]->len; }
    R->off[full.n] = tot;
    R->n_points = tot;
    R->xy = (double*)malloc(sizeof(double) * 2 * (tot ? tot : 1));
    for (int64_t i = 0; i < full.n; ++i) {
        orc_traj_t* t = full.v[i];
        R->birth[i] = t->birth; R->len[i] = t->len;
        memcpy(R->xy + 2 * R->off[i], t->pts, sizeof(double) * 2 * t->len);
        free(t->pts); free(t);
    }
    R->n_solves = s->n_solves; R->solves = s->solves;
    free(s->active.v); free(s->next_active.v); free(s->full.v); free(s->cand); free(s->nxt); free(s->flag); free(s);
    return R;
}
