// psfm_pc_control.h -- the SCALAR side of the device's trust-region loop: the control block of a solve and Ceres' control
// flow for one iteration from the global sums (TrustRegionMinimizer + DoglegStrategy/TRADITIONAL_DOGLEG of Ceres 2.0.0 as
// trajectory_optimize.cpp:74-79 configures them).  Run on the device by one thread (the last block of a launch, the reducer
// of the persistent solve, or psfm_pc_control_kernel in track-sharded runs) -- see psfm_solver.hip for who calls what.
// Host-compilable like psfm_pc_core.h: tests/test_pc_chain_host.py runs the WHOLE launch chain of the device (these functions
// + the per-track arithmetic) on the host against the CPU oracle, without a GPU.
#pragma once
#include <math.h>
#include <string.h>

#include "../../include/psfm.h"
#include "psfm_pc_core.h"

struct PsfmSolveCtrl {
    // trust-region state
    double radius, mu, x_cost, x_norm, gmax, initial_cost;
    double g2, jg2, gn2, dot;        // Gauss-Newton system sums at the current x (for the current mu)
    double gnorm, gnn, alpha;        // sqrt(g2), sqrt(gn2), g2 / jg2: what every dogleg step of this iterate starts from (pc_sums_derive)
    double dl_a, dl_b;               // dogleg step = (dl_a * ghat + dl_b * gn) / diag when dl_fixed
    double dl_norm;                  // scaled norm of that step when known a priori (cases 1, 2); < 0 -> from the kernel
    int dl_fixed, done, termination, iteration;   // dl_fixed == 0: the kernel speculates the Gauss-Newton step (0, 1)
    int n_invalid, cur, successful, nonGN;
    int n_tracks, failed, dl_case, launches;
    int fresh_x, k_first;            // x was accepted by the previous control step: gradient test pending; iterations of the
                                     // solve's first fused launch (fixes where its iterates live)
    // launch chain: the two more sums at x that price any dogleg step (psfm_pc_core.h), the model cost change of the step the
    // next launch evaluates, and what that launch is: 0 = the candidate of (dl_a, dl_b) + the system at it, 1 = the system
    // at x again (mu was raised by an invalid step)
    double qud, qdd, mcc;
    int kind_next, written;          // written: the accepted iterate is in buffer 0 / the caller's rows and the statistics are out
};

PC_HD int pc_other(int cur) { return cur == 1 ? 2 : 1; }

// Rows of the launch chain's sums (13 per launch, like the fused solve's, so that the track-sharded exchange is the same):
// the sums AT an iterate keep their fused-solve slots (SUM_XN2, SUM_GMAX, SUM_G2, SUM_JG2, SUM_GN2, SUM_DOT, SUM_FAIL); the
// slots of the per-track step sums carry (J u).(J d) and |J d|^2 instead; SUM_COST / SUM_STEP2 belong to the candidate;
// SUM_CNT / SUM_COST0 to pc_init.
enum { CH_QUD = SUM_MCC, CH_QDD = SUM_DL2 };

// ------------------------------------------------------------------------------------------------
// pc_ctrl: reduce the partials in a fixed order and run Ceres' scalar control logic.
// ------------------------------------------------------------------------------------------------
// |ghat|, |gn| and the Cauchy step length alpha = |ghat|^2 / |Js ghat/diag|^2 (ComputeCauchyPoint) of the iterate whose sums
// are in the control block: computed when the sums are adopted, not in every dogleg step (a rejected step shrinks the radius and
// asks for another step of the SAME iterate -- same values)
PC_HD void pc_sums_derive(PsfmSolveCtrl& C)
{
    C.gnorm = sqrt(C.g2); C.gnn = sqrt(C.gn2);
    C.alpha = C.g2 / C.jg2;
}

PC_HD void pc_choose_dogleg(PsfmSolveCtrl& C)
{
    // ComputeTraditionalDoglegStep
    const double gnorm = C.gnorm, gnn = C.gnn;
    const double alpha = C.alpha;
    if (gnn <= C.radius) {
        C.dl_case = 1; C.dl_a = 0.0; C.dl_b = 1.0; C.dl_norm = gnn;
    } else if (gnorm * alpha >= C.radius) {
        C.dl_case = 2; C.dl_a = -(C.radius / gnorm); C.dl_b = 0.0; C.dl_norm = C.radius;
    } else {
        const double b_dot_a = -alpha * C.dot;
        // dogleg_strategy.cc writes these squares as pow(x, 2.0): host compilers fold that into x * x (exactly rounded);
        // the device library's pow is a log/exp evaluation within 1 ulp, which moved the coefficients by an ulp
        const double ag = alpha * gnorm;
        const double a2 = ag * ag;
        const double bma2 = a2 - 2 * b_dot_a + gnn * gnn;
        const double c = b_dot_a - a2;
        const double dd = sqrt(c * c + bma2 * (C.radius * C.radius - a2));
        const double beta = (c <= 0) ? (dd - c) / bma2 : (C.radius * C.radius - a2) / (dd + c);
        C.dl_case = 3; C.dl_a = -alpha * (1.0 - beta); C.dl_b = beta; C.dl_norm = -1.0;
    }
}

// Ceres' scalar control logic for ONE trust-region iteration whose global sums are tot[PC_NSUM] (thread 0 of the last
// block).  accept_buf: the buffer that holds the candidate, i.e. where the iterate lives if the step is accepted.
PC_HD void pc_control_step(PsfmSolveCtrl& C, const double* tot, int is_init, int accept_buf)
{
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_radius = 1e-32;
    const double min_mu = 1e-8, mu_increase = 10.0;
    const int max_iter = 200, max_invalid = 5;
    if (is_init) {
        memset(&C, 0, sizeof(C));
        C.radius = 1e4; C.mu = min_mu;
        C.n_tracks = (int)tot[SUM_CNT];
        C.x_cost = tot[SUM_COST0]; C.initial_cost = C.x_cost;
        C.termination = PSFM_TERM_MAX_ITER;
        C.fresh_x = 1;   // iteration 0 counts as a successful step: its gradient test is due now
        if (C.n_tracks == 0) { C.done = 1; C.termination = PSFM_TERM_GRADIENT_TOL; }
    }
    if (!C.done && tot[SUM_FAIL] > 0.0) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; }
    if (!C.done) {
        // quantities of the CURRENT iterate evaluated by this launch
        C.x_norm = sqrt(tot[SUM_XN2]);
        C.gmax = tot[SUM_GMAX];
        C.g2 = tot[SUM_G2]; C.jg2 = tot[SUM_JG2]; C.gn2 = tot[SUM_GN2]; C.dot = tot[SUM_DOT];
        pc_sums_derive(C);
        if (C.fresh_x) {   // FinalizeIterationAndCheckIfMinimizerCanContinue after a successful step (max_iter and
            C.fresh_x = 0; // radius were tested when the step was accepted; the gradient is only known now)
            if (C.gmax <= gradient_tolerance) { C.done = 1; C.termination = PSFM_TERM_GRADIENT_TOL; }
        }
    }
    if (!C.done) {
        // Which step did the launch take, and which one does the dogleg prescribe for this radius?
        const bool used_fixed = C.dl_fixed != 0;
        const double used_a = C.dl_a, used_b = C.dl_b;
        pc_choose_dogleg(C);
        const bool step_ok = (C.dl_case == 1) ? !used_fixed : (used_fixed && used_a == C.dl_a && used_b == C.dl_b);
        if (!step_ok) {
            C.dl_fixed = (C.dl_case != 1);   // re-issue this iteration with the prescribed coefficients
        } else {
            bool rejected = false;
            C.iteration += 1;
            if (C.dl_case != 1) C.nonGN += 1;
            const double mcc = -tot[SUM_MCC];
            const double dogleg_step_norm = C.dl_norm >= 0.0 ? C.dl_norm : sqrt(tot[SUM_DL2]);
            if (!(mcc > 0.0)) {
                // HandleInvalidStep / StepIsInvalid: the next launch re-solves with the larger mu
                if (++C.n_invalid >= max_invalid) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; }
                C.mu *= mu_increase;
                C.dl_fixed = 0;
            } else {
                C.n_invalid = 0;
                const double cand = tot[SUM_COST];
                const double step_norm = sqrt(tot[SUM_STEP2]);
                if (step_norm <= parameter_tolerance * (C.x_norm + parameter_tolerance)) {
                    C.done = 1; C.termination = PSFM_TERM_PARAMETER_TOL;
                } else if (fabs(C.x_cost - cand) <= function_tolerance * C.x_cost) {
                    C.done = 1; C.termination = PSFM_TERM_FUNCTION_TOL;
                } else {
                    const double rho = (C.x_cost - cand) / mcc;
                    if (rho > min_relative_decrease) {
                        // HandleSuccessfulStep + DoglegStrategy::StepAccepted
                        C.cur = accept_buf;
                        C.x_cost = cand;
                        C.successful += 1;
                        if (rho < 0.25) C.radius *= 0.5;
                        if (rho > 0.75) C.radius = fmax(C.radius, 3.0 * dogleg_step_norm);
                        C.mu = fmax(min_mu, 2.0 * C.mu / mu_increase);
                        C.fresh_x = 1;
                        C.dl_fixed = 0;
                    } else {
                        rejected = true;
                    }
                }
            }
            // FinalizeIterationAndCheckIfMinimizerCanContinue (the gradient test of an accepted step is deferred)
            if (!C.done) {
                if (C.iteration >= max_iter) { C.done = 1; C.termination = PSFM_TERM_MAX_ITER; }
                else if (!rejected && C.radius <= min_radius) { C.done = 1; C.termination = PSFM_TERM_MIN_RADIUS; }
            }
            // StepRejected: radius /= 2 and the SAME Gauss-Newton system.  While the shrunken region still contains
            // the Gauss-Newton step the dogleg returns the same step, hence the same candidate and the same
            // rejection: replay those iterations here instead of relaunching.
            while (rejected && !C.done) {
                C.radius *= 0.5;
                if (C.radius <= min_radius) { C.done = 1; C.termination = PSFM_TERM_MIN_RADIUS; break; }
                pc_choose_dogleg(C);
                if (C.dl_case != 1) { C.dl_fixed = 1; break; }
                C.iteration += 1;   // identical step, identical rho: rejected again
                if (C.iteration >= max_iter) { C.done = 1; C.termination = PSFM_TERM_MAX_ITER; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The launch chain's control step.  Same decisions, in the same order, as pc_control_step (which replays the fused
// solve's rows); what differs is where the numbers come from: the sums AT the current iterate sit in the control block
// (adopted when that iterate was accepted -- the launch that evaluated it as a candidate reduced them ahead of the
// decision), every dogleg step of the iterate is priced from them, and a launch only contributes the candidate's cost
// and step length.
// ------------------------------------------------------------------------------------------------
// The square roots and quotients a control step of the launch chain may need from a launch's totals -- all of them functions of
// the totals and of what the control block held BEFORE the step, so they can be formed side by side (the resident solve: one
// per lane of the wave that runs the control step) instead of one behind the other on the step's critical path.  Which of them
// a step reads depends on its branch; the unread ones may be anything (NaN included).
struct PcDerived {
    double step_norm;      // sqrt(SUM_STEP2): |x - x'| of the candidate
    double x_norm;         // sqrt(SUM_XN2) at the iterate whose sums the launch reduced
    double gnorm, gnn;     // sqrt(SUM_G2), sqrt(SUM_GN2) there
    double alpha;          // SUM_G2 / SUM_JG2 there (ComputeCauchyPoint)
    double rho;            // (x_cost - SUM_COST) / mcc: the relative decrease of the candidate
};
PC_HD PcDerived pc_derive(const PsfmSolveCtrl& C, const double* tot)
{
    PcDerived D;
    D.step_norm = sqrt(tot[SUM_STEP2]);
    D.x_norm = sqrt(tot[SUM_XN2]);
    D.gnorm = sqrt(tot[SUM_G2]); D.gnn = sqrt(tot[SUM_GN2]);
    D.alpha = tot[SUM_G2] / tot[SUM_JG2];
    D.rho = (C.x_cost - tot[SUM_COST]) / C.mcc;
    return D;
}

PC_HD void pc_chain_adopt(PsfmSolveCtrl& C, const double* tot, const PcDerived& D)
{
    C.x_norm = D.x_norm;
    C.gmax = tot[SUM_GMAX];
    C.g2 = tot[SUM_G2]; C.jg2 = tot[SUM_JG2]; C.gn2 = tot[SUM_GN2]; C.dot = tot[SUM_DOT];
    C.qud = tot[CH_QUD]; C.qdd = tot[CH_QDD];
    C.gnorm = D.gnorm; C.gnn = D.gnn; C.alpha = D.alpha;
}

// The step pc_choose_dogleg has just fixed: its norm and model cost change from the sums at x; an invalid step
// (TrustRegionMinimizer::HandleInvalidStep + DoglegStrategy::StepIsInvalid) is an iteration of its own that raises mu --
// the system at x has to be reduced again before the next step can be chosen.
PC_HD void pc_chain_price(PsfmSolveCtrl& C)
{
    const double min_radius = 1e-32, mu_increase = 10.0;
    const int max_iter = 200, max_invalid = 5;
    const double a = C.dl_a, b = C.dl_b;
    if (C.dl_norm < 0.0) C.dl_norm = sqrt((a * a * C.g2 + 2.0 * a * b * C.dot) + b * b * C.gn2);
    C.mcc = -((a * C.g2 + b * C.dot) + 0.5 * ((a * a * C.jg2 + 2.0 * a * b * C.qud) + b * b * C.qdd));
    C.kind_next = 0;
    if (!(C.mcc > 0.0)) {
        C.iteration += 1;
        if (C.dl_case != 1) C.nonGN += 1;
        if (++C.n_invalid >= max_invalid) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; }
        C.mu *= mu_increase;
        C.kind_next = 1;
        if (!C.done) {
            if (C.iteration >= max_iter) { C.done = 1; C.termination = PSFM_TERM_MAX_ITER; }
            else if (C.radius <= min_radius) { C.done = 1; C.termination = PSFM_TERM_MIN_RADIUS; }
        }
    }
}

// kind 0: behind pc_init; 1: behind pc_iter (which did what C.kind_next said).  D = pc_derive(C, tot), formed before the call.
PC_HD void pc_chain_control_d(PsfmSolveCtrl& C, const double* tot, const PcDerived& D, int kind)
{
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_radius = 1e-32;
    const double min_mu = 1e-8, mu_increase = 10.0;
    const int max_iter = 200;
    if (kind == 0) {
        memset(&C, 0, sizeof(C));
        C.radius = 1e4; C.mu = min_mu;
        C.n_tracks = (int)tot[SUM_CNT];
        C.x_cost = tot[SUM_COST0]; C.initial_cost = C.x_cost;
        C.termination = PSFM_TERM_MAX_ITER;
        if (C.n_tracks == 0) { C.done = 1; C.termination = PSFM_TERM_GRADIENT_TOL; return; }
        if (tot[SUM_FAIL] > 0.0) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; return; }
        pc_chain_adopt(C, tot, D);
        if (C.gmax <= gradient_tolerance) { C.done = 1; C.termination = PSFM_TERM_GRADIENT_TOL; return; }   // iteration 0
        pc_choose_dogleg(C);
        pc_chain_price(C);
        return;
    }
    if (C.done) return;
    if (C.kind_next != 0) {     // the system at x for the raised mu
        if (tot[SUM_FAIL] > 0.0) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; return; }
        pc_chain_adopt(C, tot, D);
        pc_choose_dogleg(C);
        pc_chain_price(C);
        return;
    }
    // ---- the candidate of (dl_a, dl_b) has been evaluated ----
    bool rejected = false, accepted = false;
    C.iteration += 1;
    if (C.dl_case != 1) C.nonGN += 1;
    C.n_invalid = 0;
    const double cand = tot[SUM_COST];
    const double step_norm = D.step_norm;
    if (step_norm <= parameter_tolerance * (C.x_norm + parameter_tolerance)) {
        C.done = 1; C.termination = PSFM_TERM_PARAMETER_TOL;
    } else if (fabs(C.x_cost - cand) <= function_tolerance * C.x_cost) {
        C.done = 1; C.termination = PSFM_TERM_FUNCTION_TOL;
    } else {
        const double rho = D.rho;
        if (rho > min_relative_decrease) {
            // HandleSuccessfulStep + DoglegStrategy::StepAccepted
            C.cur = pc_other(C.cur);
            C.x_cost = cand;
            C.successful += 1;
            if (rho < 0.25) C.radius *= 0.5;
            if (rho > 0.75) C.radius = fmax(C.radius, 3.0 * C.dl_norm);
            C.mu = fmax(min_mu, 2.0 * C.mu / mu_increase);
            accepted = true;
        } else {
            rejected = true;
        }
    }
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (!C.done) {
        if (C.iteration >= max_iter) { C.done = 1; C.termination = PSFM_TERM_MAX_ITER; }
        else if (!rejected && C.radius <= min_radius) { C.done = 1; C.termination = PSFM_TERM_MIN_RADIUS; }
    }
    if (accepted && !C.done) {
        // the launch reduced the system at the candidate with the mu that is in force now: it is the current iterate's
        if (tot[SUM_FAIL] > 0.0) { C.done = 1; C.failed = 1; C.termination = PSFM_TERM_FAILURE; return; }
        pc_chain_adopt(C, tot, D);
        if (C.gmax <= gradient_tolerance) { C.done = 1; C.termination = PSFM_TERM_GRADIENT_TOL; return; }
        pc_choose_dogleg(C);
        pc_chain_price(C);
        return;
    }
    // StepRejected: radius /= 2 and the SAME Gauss-Newton system.  While the shrunken region still contains the
    // Gauss-Newton step the dogleg returns the same step, hence the same candidate and the same rejection: those
    // iterations are counted here; the first radius that prescribes another step goes to the next launch.
    while (rejected && !C.done) {
        C.radius *= 0.5;
        if (C.radius <= min_radius) { C.done = 1; C.termination = PSFM_TERM_MIN_RADIUS; break; }
        pc_choose_dogleg(C);
        if (C.dl_case != 1) { pc_chain_price(C); break; }
        C.iteration += 1;   // identical step, identical rho: rejected again
        if (C.iteration >= max_iter) { C.done = 1; C.termination = PSFM_TERM_MAX_ITER; }
    }
}

PC_HD void pc_chain_control(PsfmSolveCtrl& C, const double* tot, int kind)
{
    const PcDerived D = pc_derive(C, tot);
    pc_chain_control_d(C, tot, D, kind);
}
