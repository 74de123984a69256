"""The device's launch chain -- the trust-region loop psfm_pc_init_kernel / psfm_pc_iter_kernel / psfm_pc_resident_kernel run,
i.e. particle-sfm_amd/csrc/psfm_pc_core.h (per-track arithmetic) + psfm_pc_control.h (Ceres' control flow from the global sums,
evaluate-ahead form) -- compiled for the HOST and run against the CPU oracle (oracle/psfm_oracle.c: the restatement of
trajectory_optimize.cpp:30-96 under Ceres 2.0.0) on the batches the GPU tests use and on random ones.  Every decision of the
loop must agree (iterations, successful steps, termination, dogleg cases) and the positions to rounding; the only thing that
differs from the GPU run is the order in which the per-track terms of a sum are added.  No GPU involved: this is the part of the
solver parity that the CPU suite carries."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _common import SOLVER_BATCHES, solver_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chain(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("pc_chain") / "libpc_chain_host.so")
    cmd = ["g++", "-O2", "-mfma", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "particle-sfm_amd", "csrc"), os.path.join(ROOT, "tests", "host", "pc_chain_host.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    L = ctypes.CDLL(out)
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.pc_host_chain_solve.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_chain_solve.restype = ctypes.c_int
    L.pc_host_chain_solve_ex.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_chain_solve_ex.restype = ctypes.c_int
    L.pc_host_fused_solve.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_fused_solve.restype = ctypes.c_int
    return L


def _solve(L, uv, ref1, ref2, scale, flow12, pair=1, fused_k=0):
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 4)
    n = len(uv)
    r1 = np.ascontiguousarray(ref1, np.float64).reshape(-1, 2); r2 = np.ascontiguousarray(ref2, np.float64).reshape(-1, 2)
    sc = np.ascontiguousarray(scale, np.float64).reshape(-1)
    fl = np.ascontiguousarray(flow12, np.float32)
    H, W = fl.shape[:2]
    out = np.empty((n, 4)); stats = np.zeros(7, np.int32); costs = np.zeros(2)
    fn = L.pc_host_fused_solve if fused_k else L.pc_host_chain_solve
    rc = fn(n, uv.ctypes.data_as(dp), r1.ctypes.data_as(dp), r2.ctypes.data_as(dp), sc.ctypes.data_as(dp), fl.ctypes.data_as(fp), H, W,
            fused_k if fused_k else pair, out.ctypes.data_as(dp), stats.ctypes.data_as(ip), costs.ctypes.data_as(dp))
    return out, {"iterations": int(stats[0]), "successful_steps": int(stats[1]), "termination": int(stats[2]),
                 "dogleg_nonGN": int(stats[3]), "launches": int(stats[4]), "done": int(stats[5]), "failed": int(stats[6]),
                 "initial_cost": float(costs[0]), "final_cost": float(costs[1])}, rc


def _same_solve(a, st_a, b, st_b, tol):
    for k in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
        assert st_a[k] == st_b[k], (k, st_a, st_b)
    assert abs(st_a["initial_cost"] - st_b["initial_cost"]) <= 1e-9 * max(1.0, abs(st_b["initial_cost"]))
    assert abs(st_a["final_cost"] - st_b["final_cost"]) <= 1e-9 * max(1.0, abs(st_b["final_cost"]))
    assert float(np.abs(a - b).max()) <= tol


@pytest.mark.parametrize("H,W,n,seed,sigma,kink", [b if b[2] <= 5000 else (b[0], b[1], 20000) + b[3:] for b in SOLVER_BATCHES])
def test_device_launch_chain_on_the_host_equals_the_oracle(chain, H, W, n, seed, sigma, kink):
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    for pair in (0, 1):                      # the taps as four 8-byte loads / as two 16-byte pairs: the same values
        got, st, rc = _solve(chain, uv, ref1, ref2, scale, flow12, pair)
        assert rc == 0 and st["done"] == 1
        _same_solve(got, st, want, st_o, 1e-6)
    # launches = 1 (iteration 0) + one per iteration that needed the tracks again (replayed rejections need none)
    assert st["launches"] <= st["iterations"] + 1 + st["iterations"]


@pytest.mark.parametrize("all_tracks", [True, False])
def test_device_launch_chain_on_the_host_hands_a_failed_solve_back(chain, all_tracks):
    """Non-finite residuals at the start values: Ceres fails in IterationZero ("Residual and Jacobian evaluation failed",
    residual_block.cc IsEvaluationValid) and hands the parameters back as they came in; the reference ignores the failure
    (trajectory_optimize.cpp:81-82).  Oracle, second restatement and the device's chain: FAILURE, no iteration, input back --
    whether every track or a handful of them sample the NaN patch."""
    from oracle import oracle as orc
    from oracle import ceres_tr_numpy as ct
    uv, ref1, ref2, scale, flow12 = solver_batch(40, 50, 200, 9, 0.1, False)
    flow12 = flow12.copy(); flow12[10:14, 20:24] = np.nan
    uv[:, 0] = np.clip(uv[:, 0], 20.2, 22.8); uv[:, 1] = np.clip(uv[:, 1], 10.2, 12.8)       # p1 samples the NaN patch
    if not all_tracks:
        uv[5:, 0] += 10.3
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    second, st_n = ct.optimize_location(uv, ref1, ref2, scale, flow12, len(uv), 50, 40)
    got, st, rc = _solve(chain, uv, ref1, ref2, scale, flow12)
    assert st["failed"] == 1
    assert st["termination"] == st_o["termination"] == st_n["termination"] == 5
    assert st["iterations"] == st_o["iterations"] == st_n["iterations"] == 0
    assert np.array_equal(got, uv) and np.array_equal(want, uv) and np.array_equal(second, uv)


@pytest.mark.parametrize("after", [1, 3])
def test_device_launch_chain_on_the_host_hands_the_start_values_back_when_it_fails_behind_accepted_steps(chain, after):
    """Ceres' FAILURE behind accepted steps (a system that lost definiteness at an accepted iterate, five invalid steps in a row):
    Summary::IsSolutionUsable() is false and Solve() leaves the parameter blocks as they came in -- the reference ignores the
    failure (trajectory_optimize.cpp:81-82) and carries on with the INPUT, not with the last accepted iterate.  No batch of these
    tests gets there by itself (the 4x4 blocks stay positive definite), so the harness reports a failed factorisation in the round
    that accepts step number `after`: termination 5, `after` successful steps on record, the start values handed back.  The device's
    write-back (pc_writeback_tracks, the resident solve's) applies the same rule: buffer 0 is never written before a solve is done."""
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 2000, 3, 0.3, False)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 4)
    clean, st_clean, _ = _solve(chain, uv, ref1, ref2, scale, flow12)
    assert st_clean["successful_steps"] >= 3 and float(np.abs(clean - uv).max()) > 1e-3      # (the solve does move the tracks)
    n = len(uv)
    r1 = np.ascontiguousarray(ref1, np.float64).reshape(-1, 2); r2 = np.ascontiguousarray(ref2, np.float64).reshape(-1, 2)
    sc = np.ascontiguousarray(scale, np.float64).reshape(-1); fl = np.ascontiguousarray(flow12, np.float32)
    out = np.empty((n, 4)); stats = np.zeros(7, np.int32); costs = np.zeros(2)
    rc = chain.pc_host_chain_solve_ex(n, uv.ctypes.data_as(dp), r1.ctypes.data_as(dp), r2.ctypes.data_as(dp), sc.ctypes.data_as(dp),
                                      fl.ctypes.data_as(fp), fl.shape[0], fl.shape[1], 1, after, out.ctypes.data_as(dp),
                                      stats.ctypes.data_as(ip), costs.ctypes.data_as(dp))
    assert rc == 0 and stats[6] == 1 and stats[2] == 5 and stats[1] == after
    assert np.array_equal(out, uv)


def test_device_launch_chain_on_the_host_survives_candidates_that_do_not_evaluate(chain):
    """A NaN patch the tracks only walk INTO: the start values evaluate, a candidate does not -- Ceres treats that step as one of
    infinite cost (rejected), the radius shrinks, the solve goes on.  Same decisions in the device's loop."""
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 3000, 7, 0.05, False)
    bad = flow12.copy()
    bad[20:30, 30:50, :] = np.nan
    inside = (uv[:, 0] > 28.5) & (uv[:, 0] < 50.5) & (uv[:, 1] > 18.5) & (uv[:, 1] < 30.5)
    uv, ref1, ref2, scale = uv[~inside], ref1[~inside], ref2[~inside], scale[~inside]     # nobody STARTS in the patch
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, bad, return_stats=True)
    got, st, rc = _solve(chain, uv, ref1, ref2, scale, bad)
    assert np.isfinite(st_o["initial_cost"])
    for k in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
        assert st[k] == st_o[k], (k, st, st_o)
    if st_o["termination"] == 5:
        assert np.array_equal(got, uv) and np.array_equal(want, uv)
    else:
        assert float(np.abs(got - want).max()) <= 1e-6


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 400), sigma=st.sampled_from([0.0, 0.02, 0.1, 0.3, 0.6, 1.5]),
       kink=st.booleans(), hw=st.sampled_from([(24, 31), (40, 56), (9, 120), (64, 64)]))
def test_device_launch_chain_on_the_host_random_batches(chain, seed, n, sigma, kink, hw):
    """Small random batches -- noisy flows, points outside the image, zero scales, single tracks: every trust-region decision of
    the device's loop equals the oracle's (a batch is ONE Ceres problem: its tracks share the radius, the accept / reject
    decision and the termination test)."""
    from oracle import oracle as orc
    H, W = hw
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    got, st, rc = _solve(chain, uv, ref1, ref2, scale, flow12)
    assert rc == 0
    _same_solve(got, st, want, st_o, 1e-6)


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 300), sigma=st.sampled_from([0.0, 0.01, 0.03, 0.06, 0.1, 0.3]),
       k=st.integers(1, 8), hw=st.sampled_from([(24, 31), (40, 56), (64, 64)]), spread=st.sampled_from([0.02, 0.1, 0.5]))
def test_device_fused_solve_on_the_host_is_the_chain_or_hands_over(chain, seed, n, sigma, k, hw, spread):
    """The speculated form (what the frame kernels run: K Gauss-Newton iterations per launch, the sums replayed by the control
    step, continuation launches while every step is accepted).  Whenever it finishes, the solve is the oracle's, decision for
    decision, and bit-identical to the launch chain's on the same tracks; otherwise it hands over (return 1: the product then runs
    the chain from the start values) and has not written anything.  Clean flows with nearby start values finish this way."""
    from oracle import oracle as orc
    H, W = hw
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, False)
    rng = np.random.default_rng(seed)
    uv = np.concatenate([ref1, ref2], 1) + rng.normal(0, spread, (len(uv), 4))       # start values near the references
    got, sg, rc = _solve(chain, uv, ref1, ref2, scale, flow12, fused_k=k)
    want, so = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    ch, sc, _ = _solve(chain, uv, ref1, ref2, scale, flow12, pair=0)
    if rc == 0:
        _same_solve(got, sg, want, so, 1e-6)
        assert np.array_equal(got, ch) and all(sg[q] == sc[q] for q in ("iterations", "successful_steps", "termination", "dogleg_nonGN"))
        assert sg["final_cost"] == sc["final_cost"]
    else:
        assert rc == 1
    if so["iterations"] <= 7:        # (8 iterates per solve are buffered: longer clean solves hand over too)
        assert (rc == 0) == _clean(so), so
    _same_solve(ch, sc, want, so, 1e-6)


def _clean(so):
    """what the fused solve speculates: every iteration an accepted Gauss-Newton step at min_mu, but the one that ends the solve"""
    extra = so["iterations"] - so["successful_steps"]
    return so["dogleg_nonGN"] == 0 and (extra == 1 or (extra == 0 and so["termination"] in (2, 5)))


def test_device_fused_solve_on_the_host_finishes_exactly_the_clean_solves(chain):
    """K = 3 like the bench's sequences: solves of 4-5 clean iterations are finished by continuation launches, solves with a
    rejected or interpolated step hand over -- the split is exactly the oracle's statistics."""
    from oracle import oracle as orc
    done = 0
    for seed in range(12):
        uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 2000, 100 + seed, 0.02, False)
        uv = np.concatenate([ref1, ref2], 1) + np.random.default_rng(seed).normal(0, 0.05, (len(uv), 4))
        got, sg, rc = _solve(chain, uv, ref1, ref2, scale, flow12, fused_k=3)
        want, so = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
        assert (rc == 0) == _clean(so), so
        if rc == 0:
            done += 1
            assert so["iterations"] > 3          # more than one launch's worth: the continuation ran
            _same_solve(got, sg, want, so, 1e-6)
    assert 3 <= done <= 9
