"""Trust-region solves whose whole course can be DERIVED BY HAND from Ceres 2.0.0's loop as configured at
optimize/src/trajectory_optimize.cpp:74-79 -- no restatement is run to produce the expectations below, they are
closed forms written out in the docstrings.  With a CONSTANT flow12 map the problem is linear: r(x) = J x - b with, per track,
    J = [ I2 0 ; 0 s I2 ; -I2 I2 ],   H = J^T J = [ 2 I2  -I2 ; -I2  (s^2 + 1) I2 ],   M = diag(H) = diag(2, 2, s^2 + 1, s^2 + 1)
(path_consistency_cost.h:50-57 with dF12 = 0), Jacobi scaling cancels out of every step, the quadratic model is exact (rho = 1 on
every step), the damped Gauss-Newton step is delta = -(H + mu M)^-1 g with mu = 1e-8 throughout (mu only rises on failures), and the
trust region |D y| <= radius reads sum_k M_k delta_k^2 <= radius^2.

Every case runs on the C oracle and the NumPy restatement here (CPU suite) and on psfm_optimize_location (`-m gpu`), and must give the
derived iterations / successful steps / termination / non-Gauss-Newton count and the derived positions."""
import numpy as np
import pytest

FUNCTION_TOL, PARAMETER_TOL, GRADIENT_TOL = 0, 1, 2
MU = 1e-8
H_IMG, W_IMG = 48, 64


def constant_flow(a, b):
    f = np.empty((H_IMG, W_IMG, 2), np.float32)
    f[..., 0], f[..., 1] = a, b
    return f


def run_c_oracle(uv12, ref1, ref2, scale, flow):
    from oracle import oracle as orc
    orc.build()
    return orc.optimize_location(uv12, ref1, ref2, scale, flow, return_stats=True)


def run_numpy(uv12, ref1, ref2, scale, flow):
    from oracle import ceres_tr_numpy as ctn
    return ctn.optimize_location(uv12, ref1, ref2, scale, flow, len(uv12), flow.shape[1], flow.shape[0])


def run_gpu(uv12, ref1, ref2, scale, flow):
    from point_trajectory.optimize.build import particlesfm
    out = particlesfm.optimize_location(uv12, ref1, ref2, scale, flow, len(uv12), flow.shape[1], flow.shape[0])
    return out, particlesfm.optimize_location.last_stats


ENGINES = [pytest.param(run_c_oracle, id="c-oracle"), pytest.param(run_numpy, id="numpy-restatement"),
           pytest.param(run_gpu, id="gpu", marks=pytest.mark.gpu)]


def linear_problem(x, ref1, ref2, s, f):
    """H (n,4,4), M (n,4), g(x) (n,4), cost(x) for the constant flow f = (a, b)."""
    n = len(x)
    J = np.zeros((n, 6, 4))
    J[:, 0, 0] = J[:, 1, 1] = 1.0
    J[:, 2, 2] = J[:, 3, 3] = s
    J[:, 4, 0] = J[:, 5, 1] = -1.0
    J[:, 4, 2] = J[:, 5, 3] = 1.0
    b = np.concatenate([ref1, s[:, None] * ref2, np.broadcast_to(np.asarray(f, np.float64), (n, 2))], 1)
    Hm = np.einsum("nqa,nqb->nab", J, J)
    M = np.einsum("nqa,nqa->na", J, J)
    res = lambda y: np.einsum("nqa,na->nq", J, y) - b
    grad = lambda y: np.einsum("nqa,nq->na", J, res(y))
    cost = lambda y: 0.5 * float((res(y) ** 2).sum())
    step = lambda y: -np.linalg.solve(Hm + MU * M[:, :, None] * np.eye(4), grad(y)[:, :, None])[:, :, 0]
    return M, grad, cost, step


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("near_origin", [False, True], ids=["parameter-tolerance", "function-tolerance"])
def test_linear_problem_one_gauss_newton_step_then_a_tolerance(engine, near_origin):
    """Iteration 1: |gn| = sqrt(sum M delta^2) is far below the initial radius 1e4 -> the pure Gauss-Newton step delta1 = -(H + mu M)^-1 g0,
    rho = 1 -> accepted.  It misses the minimiser by the damping: g1 = mu M delta1 (1 + O(mu)), ~1e-8 |delta1| > gradient_tolerance 1e-10.
    Iteration 2: delta2 = -(H + mu M)^-1 g1 ~ 1e-8 delta1; the tests of trust_region_minimizer.cc run in this order:
      ParameterToleranceReached:  |delta2| <= 1e-8 (|x1| + 1e-8)   -- true when the coordinates are tens of pixels and delta1 ~ 1 px
      FunctionToleranceReached:   |cost(x1) - cost(x1 + delta2)| <= 1e-6 cost(x1)   -- the one that fires when |x1| << |delta1| (tracks at
                                  the image origin, references ~30 px away) and the minimum cost is not zero
    Either way the solve ends IN iteration 2 with x1 returned (the candidate is not applied): iterations 2, successful_steps 1,
    dogleg_nonGN 0, positions = x0 + delta1."""
    rng = np.random.default_rng(3)
    n = 64
    f = (0.75, -1.25)
    s = rng.choice([0.0, 0.25, 0.5, 1.0], n)
    if near_origin:
        p = rng.uniform(-0.02, 0.02, (n, 2))
        ref1, ref2 = p + rng.normal(0, 30.0, (n, 2)), p + rng.normal(0, 30.0, (n, 2))
    else:
        p = rng.uniform([5, 5], [W_IMG - 6, H_IMG - 6], (n, 2))
        ref1, ref2 = p + rng.normal(0, 1.0, (n, 2)), p + f + rng.normal(0, 1.0, (n, 2))
    x0 = np.concatenate([p, p if near_origin else p + f], 1)
    M, grad, cost, step = linear_problem(x0, ref1, ref2, s, f)
    d1 = step(x0)
    assert np.sqrt((M * d1 ** 2).sum()) < 1e3                       # inside the radius by a factor of ten
    x1 = x0 + d1
    assert np.abs(grad(x1)).max() > 1e-9                             # the gradient test does not fire after iteration 1
    d2 = step(x1)
    ptol = np.sqrt((d2 ** 2).sum()) / (1e-8 * (np.sqrt((x1 ** 2).sum()) + 1e-8))
    ftol = abs(cost(x1) - cost(x1 + d2)) / (1e-6 * cost(x1))
    if near_origin:
        assert ptol > 3.0 and ftol < 0.01                            # margins: rounding cannot move either test across its threshold
        want = FUNCTION_TOL
    else:
        assert ptol < 0.3
        want = PARAMETER_TOL
    out, st = engine(x0, ref1, ref2, s.reshape(-1, 1), constant_flow(*f))
    assert (st["iterations"], st["successful_steps"], st["termination"], st["dogleg_nonGN"]) == (2, 1, want, 0), st
    assert float(np.abs(out - x1).max()) <= 1e-11 * max(1.0, float(np.abs(x1).max()))
    assert abs(st["initial_cost"] - cost(x0)) <= 1e-12 * cost(x0) and abs(st["final_cost"] - cost(x1)) <= 1e-9 * cost(x1)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("n", [1, 4])
def test_radius_binds_three_times_then_gauss_newton(engine, n):
    """s = 1, references consistent with the flow (minimiser x* = (p, p + f), zero cost), every track started L = 1e5 / sqrt(n) px from x*
    along e = (1, 0, 1, 0) / sqrt(2).  e is an eigenvector of H = [2 -1; -1 2] (x) I2 with eigenvalue 1 and M = 2 I, so the steepest
    descent direction, the Cauchy point and the Gauss-Newton point all lie on the line x* + t e and the trust region is the ball
    |delta| <= radius / sqrt(2) (summed over the n identical tracks: |gn| = sqrt(2 n) L_track = sqrt(2) 1e5 at the start).
        it 1  radius 1e4     |gn| = 141421 > radius  -> step of scaled norm radius along -e: L = 1e5 - 1e4/sqrt(2)   = 92928.9,  rho = 1 -> radius 3e4
        it 2  radius 3e4     |gn| = 131421 > radius  ->                                  L = 92928.9 - 3e4/sqrt(2) = 71715.7            -> radius 9e4
        it 3  radius 9e4     |gn| = 101421 > radius  ->                                  L = 71715.7 - 9e4/sqrt(2) =  8076.1            -> radius 2.7e5
        it 4  radius 2.7e5   |gn| =  11421 <= radius -> Gauss-Newton, damped: L = 8076.1 * 2 mu / (1 + 2 mu) = 1.6e-4; |g|_inf ~ 1e-4 > 1e-10
        it 5  Gauss-Newton: |delta| = 1.6e-4 > 1e-8 |x|, cost drops by all of itself (> 1e-6 cost) -> accepted, L = 3e-12, |g|_inf <= 1e-10
    -> GRADIENT_TOL after 5 iterations, all 5 successful, 3 of them not the Gauss-Newton step; positions = x* to rounding."""
    rng = np.random.default_rng(4)
    f = (0.5, -0.25)
    p = np.floor(rng.uniform([8, 8], [W_IMG - 9, H_IMG - 9], (n, 2)))
    xs = np.concatenate([p, p + f], 1)
    L = 1e5 / np.sqrt(n)
    x0 = xs + L * np.array([1.0, 0.0, 1.0, 0.0]) / np.sqrt(2.0)
    s = np.ones((n, 1))
    out, st = engine(x0, p.copy(), p + f, s, constant_flow(*f))
    assert (st["iterations"], st["successful_steps"], st["termination"], st["dogleg_nonGN"]) == (5, 5, GRADIENT_TOL, 3), st
    assert float(np.abs(out - xs).max()) <= 1e-9
    assert abs(st["initial_cost"] - 0.5 * 1e10) <= 1e-3 and st["final_cost"] <= 1e-20


@pytest.mark.parametrize("engine", ENGINES)
def test_zero_gradient_start_ends_at_iteration_zero(engine):
    """x0 = x* exactly (integer p, flow (0.5, -0.25): every residual is an exact 0) -> |g|_inf = 0 <= 1e-10 after IterationZero:
    CONVERGENCE by gradient tolerance with no iteration run, parameters untouched bit for bit."""
    rng = np.random.default_rng(6)
    n = 33
    f = (0.5, -0.25)
    p = np.floor(rng.uniform([2, 2], [W_IMG - 3, H_IMG - 3], (n, 2)))
    x0 = np.concatenate([p, p + f], 1)
    s = rng.choice([0.0, 0.5, 1.0], (n, 1))
    out, st = engine(x0, p.copy(), p + f, s, constant_flow(*f))
    assert (st["iterations"], st["successful_steps"], st["termination"], st["dogleg_nonGN"]) == (0, 0, GRADIENT_TOL, 0), st
    assert np.array_equal(out, x0) and st["initial_cost"] == 0.0
