"""Second, independent restatement of the reference's `optimize_location` -- TEST INFRASTRUCTURE ONLY.

    point_trajectory/optimize/src/trajectory_optimize.cpp:30-96   the problem and the solver options
    point_trajectory/optimize/src/path_consistency_cost.h:42-59   the six residuals
    point_trajectory/optimize/src/linear_interpolation.h:28-44,97-123   the f64 bilinear interpolator
    Ceres Solver 2.0.0 (pinned by misc/doc/ceres.md:5; NOT in the reference tree, not in this image)

Purpose ("next round" item 2): the C oracle's trust-region loop (oracle/psfm_oracle.c
orc_optimize_location) is a restatement of a third-party library that cannot be run here -- PARITY UNPINNED.  Until a
real Ceres build pins it (tests/test_ref_ceres.py, oracle/_ref/BUILD.md), two restatements that were written
separately and agree are the strongest evidence available.  This file is written from the published structure of Ceres
2.0.0 itself, component by component, NOT from the C file:

    class Program            ceres::Problem + ProgramEvaluator: residuals, cost, gradient, Jacobian, Plus
    class StepEvaluator      internal/ceres/trust_region_step_evaluator.cc (monotonic: max_consecutive_nonmonotonic_steps 0)
    class Dogleg             internal/ceres/dogleg_strategy.cc, TRADITIONAL_DOGLEG, + sparse_normal_cholesky_solver.cc
    class Minimizer          internal/ceres/trust_region_minimizer.cc (Minimize and every helper it calls, in its order)

and it deliberately computes differently from the C file wherever the mathematics allows: dense (n,6,4) Jacobians
through einsum instead of hand-expanded sparse rows, LAPACK batch Cholesky (numpy.linalg) instead of a scalar 4x4
routine, pairwise-summed NumPy reductions instead of sequential sums.  Agreement is therefore expected in every
DECISION (iteration count, accepted steps, termination type, dogleg case) and to ~1e-12 px in the positions, not bit for
bit.  tests/test_oracle_golden.py::test_second_restatement_agrees checks exactly that on the solver batches of the GPU
tests (well-behaved, kinked, image-border, scale 0, dogleg-forcing radius, non-finite).

This is an oracle: only tests/ may import it.
"""
import numpy as np

# ceres::Solver::Options as left by trajectory_optimize.cpp:74-79 (everything not set there is the 2.0.0 default)
MAX_NUM_ITERATIONS = 200                     # :76
FUNCTION_TOLERANCE = 1e-6
GRADIENT_TOLERANCE = 1e-10
PARAMETER_TOLERANCE = 1e-8
MIN_RELATIVE_DECREASE = 1e-3
INITIAL_TRUST_REGION_RADIUS = 1e4
MAX_TRUST_REGION_RADIUS = 1e16
MIN_TRUST_REGION_RADIUS = 1e-32
MIN_LM_DIAGONAL = 1e-6
MAX_LM_DIAGONAL = 1e32
MAX_NUM_CONSECUTIVE_INVALID_STEPS = 5
JACOBI_SCALING = True
# dogleg_strategy.cc
K_MIN_MU, K_MAX_MU, MU_INCREASE_FACTOR = 1e-8, 1.0, 10.0
INCREASE_THRESHOLD, DECREASE_THRESHOLD = 0.75, 0.25

# TerminationType details, numbered like include/psfm.h PSFM_TERM_*
# The readings of Ceres 2.0.0 this restatement (and oracle/psfm_oracle.c: orc_set_variant, same keys and values) rests on memory
# for; defaults = what oracle and device implement.  tests/test_ref_ceres.py reports which combination a real Ceres build matches.
VARIANTS = {"iter0_successful": 0,     # 0: GradientToleranceReached() is tested before the first iteration; 1: only behind an accepted step
            "ftol_base": 0,            # FunctionToleranceReached() against 0: x_cost, 1: candidate_cost, 2: minimum_cost
            "failure_returns": 0,      # after FAILURE 0: the input is handed back, 1: the best iterate
            "min_cost_ties": 0}        # 0: `x_cost < minimum_cost` strictly, 1: <=
VARIANT_KEYS = ("iter0_successful", "ftol_base", "failure_returns", "min_cost_ties")      # index = the C oracle's key
VARIANT_VALUES = {"iter0_successful": (0, 1), "ftol_base": (0, 1, 2), "failure_returns": (0, 1), "min_cost_ties": (0, 1)}

TERM_FUNCTION_TOL, TERM_PARAMETER_TOL, TERM_GRADIENT_TOL, TERM_MAX_ITER, TERM_MIN_RADIUS, TERM_FAILURE = 0, 1, 2, 3, 4, 5


class Grid2D:
    """ceres::Grid2D<double, 2>(data, 0, height, 0, width), row-major, interleaved (cubic_interpolation.h):
    GetValue clamps the row and column index into the grid."""

    def __init__(self, flow_hw2):
        self.data = np.asarray(flow_hw2, dtype=np.float64)     # the py::array_t<double> force-cast, trajectory_optimize.h:40
        self.rows, self.cols = self.data.shape[0], self.data.shape[1]

    def get(self, r, c):
        return self.data[np.clip(r, 0, self.rows - 1), np.clip(c, 0, self.cols - 1)]    # (n,2)


def bilinear(grid, r, c):
    """BiLinearInterpolator::Evaluate(r, c, f, dfdr, dfdc), linear_interpolation.h:97-123 over LinearInterpolate :28-44."""
    with np.errstate(invalid="ignore"):
        fr, fc = np.floor(r), np.floor(c)
        # `const int row = std::floor(r)`: out-of-range / NaN conversions are undefined in C++; any clamped index is as
        # good as another there (such solves fail on their non-finite cost anyway)
        row = np.where(np.isfinite(fr), np.clip(fr, -1e9, 1e9), -1e9).astype(np.int64)
        col = np.where(np.isfinite(fc), np.clip(fc, -1e9, 1e9), -1e9).astype(np.int64)
    tc = (c - col)[:, None]
    tr = (r - row)[:, None]
    p0, p1 = grid.get(row, col), grid.get(row, col + 1)
    f0, df0dc = (1 - tc) * p0 + tc * p1, p1 - p0
    p0, p1 = grid.get(row + 1, col), grid.get(row + 1, col + 1)
    f1, df1dc = (1 - tc) * p0 + tc * p1, p1 - p0
    f = (1 - tr) * f0 + tr * f1
    dfdr = f1 - f0
    dfdc = (1 - tr) * df0dc + tr * df1dc
    return f, dfdr, dfdc


class Program:
    """N residual blocks AutoDiffCostFunction<PathConsistencyError, 6, 4>, one 4-parameter block each, TrivialLoss,
    no bounds, no local parameterization (trajectory_optimize.cpp:51-70): Plus(x, d) = x + d."""

    def __init__(self, ref1, ref2, scale, flow12):
        self.ref1 = np.asarray(ref1, np.float64).reshape(-1, 2)
        self.ref2 = np.asarray(ref2, np.float64).reshape(-1, 2)
        self.s = np.asarray(scale, np.float64).reshape(-1)
        self.grid = Grid2D(flow12)
        self.n = self.ref1.shape[0]

    def residuals(self, x, want_jacobian):
        """path_consistency_cost.h:42-59 with the Jet chain rule of linear_interpolation.h:132-142: the interpolator is
        called as Evaluate(row = uv12[1], col = uv12[0])."""
        x = x.reshape(self.n, 4)
        f, dfdr, dfdc = bilinear(self.grid, x[:, 1], x[:, 0])
        r = np.empty((self.n, 6))
        r[:, 0] = x[:, 0] - self.ref1[:, 0]
        r[:, 1] = x[:, 1] - self.ref1[:, 1]
        r[:, 2] = (x[:, 2] - self.ref2[:, 0]) * self.s
        r[:, 3] = (x[:, 3] - self.ref2[:, 1]) * self.s
        r[:, 4] = (x[:, 2] - x[:, 0]) - f[:, 0]
        r[:, 5] = (x[:, 3] - x[:, 1]) - f[:, 1]
        if not want_jacobian:
            return r, None
        J = np.zeros((self.n, 6, 4))
        J[:, 0, 0] = 1.0
        J[:, 1, 1] = 1.0
        J[:, 2, 2] = self.s
        J[:, 3, 3] = self.s
        # d/d(x1) of -(f_u): x1 is the COLUMN argument; d/d(y1): the ROW argument
        J[:, 4, 0] = -1.0 - dfdc[:, 0]
        J[:, 4, 1] = -dfdr[:, 0]
        J[:, 4, 2] = 1.0
        J[:, 5, 0] = -dfdc[:, 1]
        J[:, 5, 1] = -1.0 - dfdr[:, 1]
        J[:, 5, 3] = 1.0
        return r, J

    def evaluate(self, x, want_jacobian):
        """ProgramEvaluator::Evaluate: cost = 1/2 |r|^2; gradient = J^T r (from the UNSCALED Jacobian); False when a
        residual / Jacobian entry is not finite (ceres::IsArrayValid in ResidualBlock::Evaluate)."""
        r, J = self.residuals(x, want_jacobian)
        ok = bool(np.isfinite(r).all()) and (J is None or bool(np.isfinite(J).all()))
        cost = 0.5 * float(np.sum(r * r))
        g = None if J is None else np.einsum("nqc,nq->nc", J, r)
        return ok, cost, r, g, J


class StepEvaluator:
    """trust_region_step_evaluator.cc with max_consecutive_nonmonotonic_steps = 0 (use_nonmonotonic_steps false)."""

    def __init__(self, initial_cost):
        self.minimum_cost = self.current_cost = self.reference_cost = self.candidate_cost = initial_cost
        self.acc_reference = self.acc_candidate = 0.0
        self.n_nonmonotonic = 0
        self.max_nonmonotonic = 0

    def step_quality(self, cost, model_cost_change):
        if cost >= np.finfo(np.float64).max:
            return -np.finfo(np.float64).max
        rel = (self.current_cost - cost) / model_cost_change
        hist = (self.reference_cost - cost) / (self.acc_reference + model_cost_change)
        return max(rel, hist)

    def step_accepted(self, cost, model_cost_change):
        self.current_cost = cost
        self.acc_candidate += model_cost_change
        self.acc_reference += model_cost_change
        if self.current_cost < self.minimum_cost:
            self.minimum_cost = self.current_cost
            self.n_nonmonotonic = 0
            self.candidate_cost = self.current_cost
            self.acc_candidate = 0.0
        else:
            self.n_nonmonotonic += 1
            if self.current_cost > self.candidate_cost:
                self.candidate_cost = self.current_cost
                self.acc_candidate = 0.0
        if self.n_nonmonotonic == self.max_nonmonotonic:
            self.reference_cost = self.candidate_cost
            self.acc_reference = self.acc_candidate


class Dogleg:
    """dogleg_strategy.cc (TRADITIONAL_DOGLEG) over SparseNormalCholeskySolver: lhs = J^T J + D^T D, rhs = J^T r."""

    SUCCESS, FAILURE = 0, 1

    def __init__(self):
        self.radius = INITIAL_TRUST_REGION_RADIUS
        self.mu = K_MIN_MU
        self.reuse = False
        self.dogleg_step_norm = 0.0
        self.case = 0
        self.diagonal = self.gradient = self.gauss_newton = None
        self.alpha = 0.0

    def compute_step(self, J, r):
        if self.reuse:                                           # a new interpolant only
            return self.SUCCESS, self._traditional_dogleg()
        self.reuse = True
        col2 = np.einsum("nqc,nqc->nc", J, J)                    # SquaredColumnNorm
        self.diagonal = np.sqrt(np.minimum(np.maximum(col2, MIN_LM_DIAGONAL), MAX_LM_DIAGONAL))
        # ComputeGradient: (J^T r) / diagonal
        self.gradient = np.einsum("nqc,nq->nc", J, r) / self.diagonal
        # ComputeCauchyPoint: alpha = |g|^2 / |J (g / diagonal)|^2
        Jg = np.einsum("nqc,nc->nq", J, self.gradient / self.diagonal)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.alpha = float(np.float64(np.sum(self.gradient ** 2)) / np.float64(np.sum(Jg ** 2)))      # (0 / 0 = NaN, as in C)
        # ComputeGaussNewtonStep: retry with mu * 10 while the factorisation fails / the solution is not finite
        status = self.FAILURE
        while self.mu < K_MAX_MU:
            D = self.diagonal * np.sqrt(self.mu)                 # lm_diagonal
            lhs = np.einsum("nqa,nqb->nab", J, J)
            idx = np.arange(4)
            lhs[:, idx, idx] += D * D
            rhs = np.einsum("nqc,nq->nc", J, r)
            y = None
            try:
                with np.errstate(all="ignore"):
                    L = np.linalg.cholesky(lhs)
                    z = np.linalg.solve(L, rhs[:, :, None])
                    y = np.linalg.solve(np.transpose(L, (0, 2, 1)), z)[:, :, 0]
            except np.linalg.LinAlgError:
                y = None
            if y is None or not np.isfinite(y).all():
                self.mu *= MU_INCREASE_FACTOR
                continue
            self.gauss_newton = y * (-self.diagonal)             # gauss_newton_step_.array() *= -diagonal_.array()
            status = self.SUCCESS
            break
        if status != self.SUCCESS:
            return status, None
        return status, self._traditional_dogleg()

    def _traditional_dogleg(self):
        gradient_norm = float(np.sqrt(np.sum(self.gradient ** 2)))
        gauss_newton_norm = float(np.sqrt(np.sum(self.gauss_newton ** 2)))
        if gauss_newton_norm <= self.radius:                     # case 1
            self.case = 1
            self.dogleg_step_norm = gauss_newton_norm
            return self.gauss_newton / self.diagonal
        if gradient_norm * self.alpha >= self.radius:            # case 2
            self.case = 2
            self.dogleg_step_norm = self.radius
            return (-(self.radius / gradient_norm) * self.gradient) / self.diagonal
        self.case = 3                                            # case 3
        b_dot_a = -self.alpha * float(np.sum(self.gradient * self.gauss_newton))
        a_squared_norm = (self.alpha * gradient_norm) ** 2
        b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gauss_newton_norm ** 2
        c = b_dot_a - a_squared_norm
        d = np.sqrt(c * c + b_minus_a_squared_norm * (self.radius ** 2 - a_squared_norm))
        beta = (d - c) / b_minus_a_squared_norm if c <= 0 else (self.radius * self.radius - a_squared_norm) / (d + c)
        step = (-self.alpha * (1.0 - beta)) * self.gradient + beta * self.gauss_newton
        self.dogleg_step_norm = float(np.sqrt(np.sum(step ** 2)))
        return step / self.diagonal

    def step_accepted(self, step_quality):
        if step_quality < DECREASE_THRESHOLD:
            self.radius *= 0.5
        if step_quality > INCREASE_THRESHOLD:
            self.radius = max(self.radius, 3.0 * self.dogleg_step_norm)
        self.mu = max(K_MIN_MU, 2.0 * self.mu / MU_INCREASE_FACTOR)
        self.reuse = False

    def step_rejected(self, step_quality):
        self.radius *= 0.5
        self.reuse = True

    def step_is_invalid(self):
        self.mu *= MU_INCREASE_FACTOR
        self.reuse = False


class Minimizer:
    """trust_region_minimizer.cc, TrustRegionMinimizer::Minimize and its helpers in the order Minimize calls them."""

    def __init__(self, program, x0):
        self.p = program
        self.x = np.array(x0, np.float64).reshape(program.n, 4)
        self.parameters = self.x.copy()                 # the user's parameter blocks (updated when the cost improves)
        self.x_norm = float(np.sqrt(np.sum(self.x ** 2)))      # Init()
        self.minimum_cost = np.finfo(np.float64).max
        self.strategy = Dogleg()
        self.iteration = 0
        self.num_successful_steps = 0
        self.nonGN = 0
        self.num_consecutive_invalid_steps = 0
        self.termination = TERM_MAX_ITER
        self.scaling = None
        self.step_is_valid = self.step_is_successful = False
        self.gradient_max_norm = 0.0

    # -- EvaluateGradientAndJacobian --
    def _evaluate_gradient_and_jacobian(self):
        ok, self.x_cost, self.residuals, gradient, J = self.p.evaluate(self.x, True)
        if not ok:
            return False
        if JACOBI_SCALING:
            if self.iteration == 0:
                self.scaling = 1.0 / (1.0 + np.sqrt(np.einsum("nqc,nqc->nc", J, J)))
            J = J * self.scaling[:, None, :]
        self.jacobian = J
        projected = self.x + (-gradient)                       # Plus(x, -gradient)
        self.gradient_max_norm = float(np.max(np.abs(self.x - projected))) if self.p.n else 0.0
        return True

    def _iteration_zero(self):
        self.iteration = 0
        if not self._evaluate_gradient_and_jacobian():
            return False
        self.initial_cost = self.x_cost
        self.step_is_valid = True
        self.step_is_successful = VARIANTS["iter0_successful"] == 0
        if not self.step_is_successful:          # (the variant must still record the start values as the best so far)
            self.num_successful_steps += 1
            self.minimum_cost = self.x_cost
            self.parameters = self.x.copy()
        return True

    def _finalize_iteration_and_check_if_minimizer_can_continue(self):
        if self.step_is_successful:
            self.num_successful_steps += 1
            if self.x_cost < self.minimum_cost or (VARIANTS["min_cost_ties"] and self.x_cost == self.minimum_cost):
                self.minimum_cost = self.x_cost
                self.parameters = self.x.copy()
        radius = self.strategy.radius
        if self.iteration >= MAX_NUM_ITERATIONS:                # MaxSolverIterationsReached
            self.termination = TERM_MAX_ITER
            return False
        if self.step_is_successful and self.gradient_max_norm <= GRADIENT_TOLERANCE:    # GradientToleranceReached
            self.termination = TERM_GRADIENT_TOL
            return False
        if radius <= MIN_TRUST_REGION_RADIUS:                   # MinTrustRegionRadiusReached
            self.termination = TERM_MIN_RADIUS
            return False
        return True

    def _compute_trust_region_step(self):
        self.step_is_valid = False
        status, step = self.strategy.compute_step(self.jacobian, self.residuals)
        if status == Dogleg.FAILURE:
            return
        model_residuals = np.einsum("nqc,nc->nq", self.jacobian, step)
        self.model_cost_change = -float(np.sum(model_residuals * (self.residuals + model_residuals / 2.0)))
        self.step_is_valid = self.model_cost_change > 0.0
        if self.step_is_valid:
            self.delta = step * self.scaling
            self.num_consecutive_invalid_steps = 0
            if self.strategy.case != 1:
                self.nonGN += 1

    def _handle_invalid_step(self):
        self.num_consecutive_invalid_steps += 1
        if self.num_consecutive_invalid_steps >= MAX_NUM_CONSECUTIVE_INVALID_STEPS:
            self.termination = TERM_FAILURE
            return False
        self.strategy.step_is_invalid()
        return True

    def _compute_candidate_point_and_evaluate_cost(self):
        self.candidate_x = self.x + self.delta
        ok, cost, _, _, _ = self.p.evaluate(self.candidate_x, False)
        self.candidate_cost = cost if ok else np.finfo(np.float64).max

    def _parameter_tolerance_reached(self):
        self.step_norm = float(np.sqrt(np.sum((self.x - self.candidate_x) ** 2)))
        return self.step_norm <= PARAMETER_TOLERANCE * (self.x_norm + PARAMETER_TOLERANCE)

    def _function_tolerance_reached(self):
        cost_change = self.x_cost - self.candidate_cost
        base = (self.x_cost, self.candidate_cost, self.minimum_cost)[VARIANTS["ftol_base"]]
        return abs(cost_change) <= FUNCTION_TOLERANCE * base

    def minimize(self):
        if not self._iteration_zero():
            self.termination = TERM_FAILURE
            return self
        self.step_evaluator = StepEvaluator(self.x_cost)
        while self._finalize_iteration_and_check_if_minimizer_can_continue():
            self.iteration += 1
            self.step_is_successful = False
            self._compute_trust_region_step()
            if not self.step_is_valid:
                if not self._handle_invalid_step():
                    return self
                continue
            self._compute_candidate_point_and_evaluate_cost()
            if self._parameter_tolerance_reached():
                self.termination = TERM_PARAMETER_TOL
                return self
            if self._function_tolerance_reached():
                self.termination = TERM_FUNCTION_TOL
                return self
            self.relative_decrease = self.step_evaluator.step_quality(self.candidate_cost, self.model_cost_change)
            if self.relative_decrease > MIN_RELATIVE_DECREASE:      # IsStepSuccessful -> HandleSuccessfulStep
                self.x = self.candidate_x
                self.x_norm = float(np.sqrt(np.sum(self.x ** 2)))
                if not self._evaluate_gradient_and_jacobian():
                    self.termination = TERM_FAILURE
                    return self
                self.step_is_successful = True
                self.strategy.step_accepted(self.relative_decrease)
                self.step_evaluator.step_accepted(self.candidate_cost, self.model_cost_change)
            else:
                self.strategy.step_rejected(self.relative_decrease)
        return self


def optimize_location(uv12, uv_ref1, uv_ref2, ref2_scale, flow12_map, total_num, width, height):
    """trajectory_optimize.cpp:30-96, same signature.  Returns ((N,4) f64, stats dict)."""
    n = int(total_num)
    fm = np.asarray(flow12_map)
    assert fm.shape[0] == int(height) and fm.shape[1] == int(width)
    x0 = np.asarray(uv12, np.float64).reshape(-1, 4)[:n]
    if n == 0:
        return np.zeros((0, 4)), {"iterations": 0, "successful_steps": 0, "termination": -1, "dogleg_nonGN": 0,
                                  "initial_cost": 0.0, "final_cost": 0.0}
    prog = Program(np.asarray(uv_ref1, np.float64).reshape(-1, 2)[:n], np.asarray(uv_ref2, np.float64).reshape(-1, 2)[:n],
                   np.asarray(ref2_scale, np.float64).reshape(-1)[:n], fm)
    m = Minimizer(prog, x0).minimize()
    # solver.cc: the user's parameters are written back only when Summary::IsSolutionUsable(); after FAILURE the state is
    # restored -- the reference ignores the summary (trajectory_optimize.cpp:81-82) and returns whatever is there
    out = x0.copy() if (m.termination == TERM_FAILURE and VARIANTS["failure_returns"] == 0) else m.parameters
    stats = {"iterations": m.iteration, "successful_steps": max(m.num_successful_steps - 1, 0),
             "termination": m.termination, "dogleg_nonGN": m.nonGN,
             "initial_cost": getattr(m, "initial_cost", float("nan")),
             "final_cost": m.minimum_cost if m.minimum_cost < np.finfo(np.float64).max else float("nan")}
    return out, stats
