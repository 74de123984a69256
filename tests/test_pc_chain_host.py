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
           "-I", os.path.join(ROOT, "particle-sfm_amd", "csrc"), os.path.join(ROOT, "tests", "host", "pc_chain_host.cpp"),
           os.path.join(ROOT, "tests", "host", "pc_resident_host.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    L = ctypes.CDLL(out)
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.pc_host_chain_solve.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_chain_solve.restype = ctypes.c_int
    L.pc_host_chain_solve_ex.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_chain_solve_ex.restype = ctypes.c_int
    L.pc_host_fused_solve.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_fused_solve.restype = ctypes.c_int
    ubp = ctypes.POINTER(ctypes.c_ubyte)
    L.pc_host_resident_solve.argtypes = [ctypes.c_long, ubp, dp, dp, dp, dp, fp] + [ctypes.c_int] * 8 + [dp, ip, dp, ip]
    L.pc_host_resident_solve.restype = ctypes.c_int
    L.pc_host_peer_solve.argtypes = [ctypes.c_long, ubp, ubp, ctypes.c_int, dp, dp, dp, dp, fp] + [ctypes.c_int] * 8 + [dp, ip, dp, ip]
    L.pc_host_peer_solve.restype = ctypes.c_int
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


# ---- the RESIDENT solve's bookkeeping (csrc/psfm_pc_resident.h: lists, slots, the streamed tail, the accepted-step update, the
#      block tree, the write-back, giving up) on the host ----
def _resident(L, uv, ref1, ref2, scale, flow12, n_blocks, ns, banded=1, init_inside=1, part=None, quit=(-1, -1)):
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 4)
    n = len(uv)
    r1 = np.ascontiguousarray(ref1, np.float64).reshape(-1, 2); r2 = np.ascontiguousarray(ref2, np.float64).reshape(-1, 2)
    sc = np.ascontiguousarray(scale, np.float64).reshape(-1)
    fl = np.ascontiguousarray(flow12, np.float32)
    H, W = fl.shape[:2]
    out = np.full((n, 4), -777.0); stats = np.zeros(7, np.int32); costs = np.zeros(2); info = np.zeros(4, np.int32)
    pm = None if part is None else np.ascontiguousarray(part, np.uint8)
    rc = L.pc_host_resident_solve(n, None if pm is None else pm.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), uv.ctypes.data_as(dp),
                                  r1.ctypes.data_as(dp), r2.ctypes.data_as(dp), sc.ctypes.data_as(dp), fl.ctypes.data_as(fp), H, W,
                                  int(n_blocks), int(ns), int(banded), int(init_inside), int(quit[0]), int(quit[1]),
                                  out.ctypes.data_as(dp), stats.ctypes.data_as(ip), costs.ctypes.data_as(dp), info.ctypes.data_as(ip))
    return out, {"iterations": int(stats[0]), "successful_steps": int(stats[1]), "termination": int(stats[2]),
                 "dogleg_nonGN": int(stats[3]), "launches": int(stats[4]), "done": int(stats[5]), "failed": int(stats[6]),
                 "initial_cost": float(costs[0]), "final_cost": float(costs[1])}, rc, \
        {"longest_list": int(info[0]), "streamed": int(info[1]), "rounds": int(info[2]), "empty_blocks": int(info[3])}


RESIDENT_SHAPES = [
    # n_blocks, slots per thread, banded lists, iteration 0 inside the launch
    (1, 1, 1, 1), (1, 3, 1, 1),            # one block holds everything: 256 / 768 tracks on chip, the rest streamed
    (5, 2, 0, 1), (12, 1, 1, 0),           # (12 blocks: what a batch of 3000 tracks runs on; iteration 0 as its own launch)
    (40, 3, 1, 1),                         # more blocks than leaders: the first level of the tree has members
    (64, 2, 1, 1), (64, 1, 1, 0),          # a grid that is a multiple of 8: XCD-banded lists once there are >= 64 chunks
    (100, 3, 1, 1),                        # more blocks than chunks: empty lists
]


@pytest.mark.parametrize("n_blocks,ns,banded,init_inside", RESIDENT_SHAPES)
@pytest.mark.parametrize("H,W,n,seed,sigma,kink", [(60, 80, 3000, 2, 0.5, False), (45, 70, 5000, 3, 0.3, True), (270, 480, 20000, 4, 0.05, False),
                                                   (33, 47, 1, 5, 0.1, False)])
def test_resident_bookkeeping_on_the_host_is_the_launch_chain(chain, n_blocks, ns, banded, init_inside, H, W, n, seed, sigma, kink):
    """psfm_pc_resident_kernel's slot / list / streamed-tail / write-back logic with the device's own functions
    (csrc/psfm_pc_resident.h), for block counts and slot counts that put the tracks everywhere a launch can put them -- all on chip,
    most of them streamed behind the slots, lists of different lengths, empty blocks: the same solve as the launch chain's (every
    decision; positions to the rounding of another order of summation) and as the oracle's."""
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    ch, sc, _ = _solve(chain, uv, ref1, ref2, scale, flow12, pair=1)
    got, st, rc, info = _resident(chain, uv, ref1, ref2, scale, flow12, n_blocks, ns, banded, init_inside)
    assert rc == 0 and st["done"] == 1
    _same_solve(got, st, ch, sc, 1e-7)
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    _same_solve(got, st, want, st_o, 1e-6)
    if n_blocks == 1 and n > ns * 256:
        assert info["streamed"] == n - ns * 256 and info["longest_list"] == n
    if n_blocks == 100 and n <= 5000:
        assert info["empty_blocks"] > 0


def test_resident_bookkeeping_rows_that_do_not_take_part(chain):
    """Frame mode's lists hold only the lanes whose track has three buffered points: a random 55 % of the rows take part; their solve
    is the launch chain's on exactly those rows, the others are never touched."""
    uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 6000, 12, 0.4, False)
    rng = np.random.default_rng(5)
    part = rng.uniform(size=len(uv)) < 0.55
    ch, sc, _ = _solve(chain, uv[part], ref1[part], ref2[part], scale[part], flow12, pair=1)
    for n_blocks, ns in ((7, 1), (24, 3)):
        got, st, rc, info = _resident(chain, uv, ref1, ref2, scale, flow12, n_blocks, ns, part=part)
        assert rc == 0
        _same_solve(got[part], st, ch, sc, 1e-7)
        assert np.all(got[~part] == -777.0)


@pytest.mark.parametrize("quit", [(0, 0), (3, 1), (11, 4)])
def test_resident_bookkeeping_giving_up_writes_nothing(chain, quit):
    """A block that leaves in round r (its hand-off timed out: the grid was not co-resident) takes the solve with it -- and nothing
    has been written: the caller's rows are as they were, the launches redo the solve from iteration 0."""
    uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 3000, 2, 0.5, False)
    got, st, rc, info = _resident(chain, uv, ref1, ref2, scale, flow12, 12, 3, quit=quit)
    assert rc == 2 and np.all(got == -777.0)


def test_resident_bookkeeping_failed_solve_hands_the_start_values_back(chain):
    """Non-finite residuals at the start values (Ceres' FAILURE in IterationZero): slots and streamed entries alike keep the input."""
    uv, ref1, ref2, scale, flow12 = solver_batch(40, 50, 900, 9, 0.1, False)
    flow12 = flow12.copy(); flow12[10:14, 20:24] = np.nan
    uv[:, 0] = np.clip(uv[:, 0], 20.2, 22.8); uv[:, 1] = np.clip(uv[:, 1], 10.2, 12.8)
    got, st, rc, info = _resident(chain, uv, ref1, ref2, scale, flow12, 2, 1)
    assert rc == 0 and st["failed"] == 1 and st["termination"] == 5 and info["streamed"] > 0
    assert np.array_equal(got, np.ascontiguousarray(uv, np.float64).reshape(-1, 4))


def test_block_tree_order_is_what_the_device_adds(chain):
    """pc_tree_totals (the order every form of the chain adds the blocks' sums in: members -> 32 leaders -> four groups of eight)
    against the same order written out with NumPy, for block counts on both sides of the leader count."""
    import numpy as np
    src = r"""
    #include "psfm_pc_resident.h"
    extern "C" void tree(const double* rows, int n_blocks, double* tot) { pc_tree_totals(rows, PC_NSUM, n_blocks, PC_NSUM, tot); }
    """
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, "t.cpp"), "w").write(src)
    so = os.path.join(d, "t.so")
    subprocess.run(["g++", "-O2", "-mfma", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "particle-sfm_amd", "csrc"),
                    os.path.join(d, "t.cpp"), "-o", so], check=True)
    T = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    T.tree.argtypes = [dp, ctypes.c_int, dp]
    rng = np.random.default_rng(3)
    for nb in (1, 2, 31, 32, 33, 100, 128, 129, 512, 1000):
        rows = rng.standard_normal((nb, 13)) * 10.0 ** rng.integers(-6, 7, size=(nb, 13))
        tot = np.zeros(13)
        T.tree(np.ascontiguousarray(rows).ctypes.data_as(dp), nb, tot.ctypes.data_as(dp))
        Lq, Q = min(32, nb), ((nb + 31) // 32 + 3) // 4
        for k in range(13):
            mx = k == 5
            red = (lambda a, b: max(a, b)) if mx else (lambda a, b: a + b)
            S = []
            for x in range(Lq):
                cnt = (nb - x + 31) // 32
                sj = []
                for j in range(4):
                    v = 0.0
                    for u in range(Q):
                        m = j * Q + u
                        if m < cnt:
                            v = red(v, rows[x + 32 * m, k])
                    sj.append(v)
                S.append(red(red(red(sj[0], sj[1]), sj[2]), sj[3]))
            t = []
            for j in range(4):
                v = 0.0
                for x in range(8 * j, min(8 * j + 8, Lq)):
                    v = red(v, S[x])
                t.append(v)
            assert tot[k] == red(red(red(t[0], t[1]), t[2]), t[3]), (nb, k)


def _peer(L, uv, ref1, ref2, scale, flow12, owner, world, n_blocks, ns, banded=1, init_inside=1, quit=(-1, -1)):
    dp, fp, ip, ubp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ubyte)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 4)
    n = len(uv)
    r1 = np.ascontiguousarray(ref1, np.float64).reshape(-1, 2); r2 = np.ascontiguousarray(ref2, np.float64).reshape(-1, 2)
    sc = np.ascontiguousarray(scale, np.float64).reshape(-1)
    fl = np.ascontiguousarray(flow12, np.float32)
    H, W = fl.shape[:2]
    ow = np.ascontiguousarray(owner, np.uint8)
    out = np.full((n, 4), -777.0); stats = np.zeros(7, np.int32); costs = np.zeros(2); info = np.zeros(4, np.int32)
    rc = L.pc_host_peer_solve(n, None, ow.ctypes.data_as(ubp), int(world), uv.ctypes.data_as(dp), r1.ctypes.data_as(dp), r2.ctypes.data_as(dp),
                              sc.ctypes.data_as(dp), fl.ctypes.data_as(fp), H, W, int(n_blocks), int(ns), int(banded), int(init_inside),
                              int(quit[0]), int(quit[1]), out.ctypes.data_as(dp), stats.ctypes.data_as(ip), costs.ctypes.data_as(dp),
                              info.ctypes.data_as(ip))
    return out, {"iterations": int(stats[0]), "successful_steps": int(stats[1]), "termination": int(stats[2]), "dogleg_nonGN": int(stats[3]),
                 "done": int(stats[5]), "failed": int(stats[6]), "initial_cost": float(costs[0]), "final_cost": float(costs[1])}, rc


@pytest.mark.parametrize("world,n_blocks,ns", [(2, 5, 2), (3, 12, 1), (8, 4, 3), (4, 40, 3)])
@pytest.mark.parametrize("H,W,n,seed,sigma,kink", [(60, 80, 3000, 2, 0.5, False), (45, 70, 5000, 3, 0.3, True)])
def test_cross_rank_bookkeeping_on_the_host(chain, world, n_blocks, ns, H, W, n, seed, sigma, kink):
    """psfm_shard_solve_peer's arithmetic without a GPU: the tracks of ONE solve dealt to `world` ranks (bands of rows, as connect_sharded
    deals them), every rank the resident launch's bookkeeping over its own rows (csrc/psfm_pc_resident.h), every rank's tree total, the
    totals added in RANK ORDER (pc_peer_totals: what every block of every rank does with the leader rows in its area), one control
    step on them -- same decisions as one rank and as the oracle, positions to the rounding of another grouping of the sums; a rank
    whose block gives up takes every rank's solve with it."""
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    owner = (np.arange(n) * world // n).astype(np.uint8)            # contiguous bands of rows
    one, st1, rc1, _ = _resident(chain, uv, ref1, ref2, scale, flow12, n_blocks, ns)
    got, st, rc = _peer(chain, uv, ref1, ref2, scale, flow12, owner, world, n_blocks, ns)
    assert rc == 0 and rc1 == 0 and st["done"] == 1
    for key in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
        assert st[key] == st1[key], key
    assert float(np.abs(got - one).max()) <= 1e-9
    want, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    _same_solve(got, st, want, st_o, 1e-6)
    # ... and with ONE rank it is the one-GPU bookkeeping bit for bit
    same, st_s, rc_s = _peer(chain, uv, ref1, ref2, scale, flow12, np.zeros(n, np.uint8), 1, n_blocks, ns)
    assert rc_s == 0 and np.array_equal(same, one) and st_s["iterations"] == st1["iterations"]
    # a block of the LAST rank leaves in round 2: nothing is written anywhere
    out, _, rc_q = _peer(chain, uv, ref1, ref2, scale, flow12, owner, world, n_blocks, ns, quit=(world * n_blocks - 1, 2))
    assert rc_q == 2 and (out == -777.0).all()
