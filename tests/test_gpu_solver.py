"""GPU parity of the path-consistency solver (psfm_optimize_location / track_optimize) against the CPU oracle.
Tolerance: ids / lengths bit-exact, positions within 1e-4 px (north_star); in practice ~1e-10 because both
sides run the same Ceres-compatible control flow in f64 and differ only in rounding (operation and summation order)."""
import numpy as np
import pytest

from _common import golden, regen_inputs, assert_csr_equal, solver_batch, SOLVER_BATCHES, LARGE_MOTION, REALISTIC_OPT
import psfm_synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def pt():
    import torch
    assert torch.cuda.is_available()
    from point_trajectory import utils, trajectory, _hip
    from point_trajectory.track_optimize import track_optimize
    from point_trajectory.optimize.build import particlesfm
    _hip.context()
    class NS: pass
    ns = NS()
    ns.utils, ns.trajectory, ns.track_optimize, ns.particlesfm = utils, trajectory, track_optimize, particlesfm
    return ns


_batch = solver_batch


# How a frame's solve is run (psfm_ctx_set_solver): adaptive, the launch chain, the fused solve (one launch that
# speculates k Gauss-Newton iterations), and the fused solve with k = 1 -- which no real solve fits, so every one of
# them is redone by the chain from the untouched buffer.  The results must not depend on it.
SOLVER_MODES = [(0, 0, 1), (1, 0, 1), (2, 0, 1), (2, 1, 1), (2, 0, 0), (2, 1, 0)]


@pytest.fixture(params=SOLVER_MODES, ids=["adaptive", "launch-chain", "fused", "fused-k1-continued", "fused-host-paced", "fused-k1-redone"])
def solver_mode(request, pt):
    """(mode, k, device-paced): the last flag is PSFM_SEQ -- 0 keeps one host-parameterised frame kernel per frame, where a
    solve that needs more iterations than speculated stalls and is redone (with k = 1: every solve)."""
    import os
    from point_trajectory import _hip
    ctx = _hip.context()
    ctx.set_solver(request.param[0], request.param[1])
    if not request.param[2]:
        os.environ["PSFM_SEQ"] = "0"
    yield request.param
    os.environ.pop("PSFM_SEQ", None)
    ctx.set_solver(0, 0)


@pytest.mark.parametrize("H,W,n,seed,sigma,kink", SOLVER_BATCHES)
def test_optimize_location_vs_oracle(pt, H, W, n, seed, sigma, kink):
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = _batch(H, W, n, seed, sigma, kink)
    out_o, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    out_g = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, n, W, H)
    st_g = pt.particlesfm.optimize_location.last_stats
    assert st_g["iterations"] == st_o["iterations"], (st_g, st_o)
    assert st_g["successful_steps"] == st_o["successful_steps"] and st_g["termination"] == st_o["termination"]
    assert st_g["dogleg_nonGN"] == st_o["dogleg_nonGN"]
    assert abs(st_g["final_cost"] - st_o["final_cost"]) <= 1e-9 * max(1.0, st_o["final_cost"])
    # far inside the 1e-4 px bar: the device solves the same normal equations in another order (psfm_pc_core.h: unscaled system,
    # 2x2 Schur complement, contracted multiply-adds) than the C restatement (scaled system, dense Cholesky), so the iterates
    # differ by rounding; every decision of the loop above is the same
    assert float(np.abs(out_g - out_o).max()) <= 1e-8


def test_optimize_location_exercises_dogleg(pt):
    """The noisy batch must actually leave the pure Gauss-Newton path, otherwise cases 2/3 are untested."""
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = _batch(60, 80, 3000, 2, 0.5)
    _, st = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    assert st["dogleg_nonGN"] > 0 and st["iterations"] > st["successful_steps"]


@pytest.mark.parametrize("name", ["opt_48x64_r2", "opt_45x70_r3"])
def test_track_optimize_golden(pt, solver_mode, name):
    g = golden(name)
    d = regen_inputs(g, stride2=True)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = pt.utils.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g, tol=TOL)
    assert float(np.abs(R.xy - g["xy"]).max()) <= 1e-7


@pytest.mark.parametrize("name", LARGE_MOTION)
def test_track_optimize_large_motion_golden(pt, solver_mode, name):
    """`loss02_scale = (1 - occ02) * (|flow02| < 20)` (trajectory.py:179; csrc/psfm_solver.hip) on both sides of the 20 px gate, with
    fractional occ02 weights and tracks drifting out of the image: fixtures from the reference's own Python (the fixture counts the
    gate decisions it went through), every way of running the solve; the masks of the large flows are compared too."""
    g = golden(name)
    assert int(g["gate_closed"]) > 1000 and int(g["gate_open"]) > 1000 and int(g["scale_fractional"]) > 100
    d = regen_inputs(g, stride2=True)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = pt.utils.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    assert np.array_equal(np.packbits(np.stack(occ)), g["occ"]) and np.array_equal(np.packbits(np.stack(occ2)), g["occ2"])
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g, tol=TOL)
    # (sigma 0.25 on 90 x 140: solves of 20-30 iterations through bilinear kinks amplify the rounding differences between the
    # device's and the restatement's arithmetic to a few 1e-7 px; the bar is 1e-4)
    assert float(np.abs(R.xy - g["xy"]).max()) <= 1e-5


@pytest.mark.parametrize("name", REALISTIC_OPT)
def test_track_optimize_realistic_golden(pt, solver_mode, name):
    """The third distribution (psfm_synth.REALISTIC: layers with true (dis)occlusion, correlated flow error, outlier blobs) against
    the fixture of the reference's own Python, every way of running the solve: masks of both strides, ids, lengths, positions, and
    the iteration count of every solve (all of them reject steps: the launch chain / resident solve, or the redo of a fused solve)."""
    g = golden(name)
    d = regen_inputs(g, stride2=True)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = pt.utils.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    assert np.array_equal(np.packbits(np.stack(occ)), g["occ"]) and np.array_equal(np.packbits(np.stack(occ2)), g["occ2"])
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g, tol=TOL)
    assert float(np.abs(R.xy - g["xy"]).max()) <= 1e-5
    assert [s["iterations"] for s in R.solve_stats] == list(g["solve_iterations"])
    assert [s["successful_steps"] for s in R.solve_stats] == list(g["solve_successful"])


@pytest.mark.parametrize("H,W,T,r,seed,sigma,nocc", [
    (120, 200, 9, 2, 51, 0.05, 2),
    (90, 140, 8, 3, 52, 0.3, 3),
    (436, 1024, 8, 2, 53, 0.05, 2),    # configs[2] shape (Sintel alley_1), fewer frames
    (64, 96, 20, 1, 54, 0.15, 1),
])
def test_track_optimize_vs_oracle(pt, solver_mode, H, W, T, r, seed, sigma, nocc):
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert len(R) == O.n_traj
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    err = float(np.abs(R.xy - O.xy).max())
    assert err <= TOL, err
    assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
    assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves]


def test_failed_solve_leaves_parameters_untouched(pt):
    """A solve Ceres would end in FAILURE (here: non-finite residuals at iteration 0, and a map that is NaN where the
    tracks walk to after a few accepted steps) is ignored by the reference (trajectory_optimize.cpp:81-82) and Ceres
    hands the parameters back as they came in (solver.cc Minimize / IsSolutionUsable): same here, status OK,
    termination 5 in the statistics."""
    from oracle import oracle as orc
    uv, ref1, ref2, scale, flow12 = _batch(60, 80, 3000, 7, 0.05)
    bad = flow12.copy()
    bad[20:30, 30:50, :] = np.nan
    out_o, st_o = orc.optimize_location(uv, ref1, ref2, scale, bad, return_stats=True)
    out_g = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, bad, uv.shape[0], 80, 60)
    st_g = pt.particlesfm.optimize_location.last_stats
    assert st_o["termination"] == 5 and st_g["termination"] == 5
    assert np.array_equal(out_o, uv) and np.array_equal(out_g, uv)
    # and the next solve on the same context is unaffected
    out_o, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    out_g = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, uv.shape[0], 80, 60)
    assert st_o["termination"] != 5 and pt.particlesfm.optimize_location.last_stats["termination"] == st_o["termination"]
    assert float(np.abs(out_g - out_o).max()) <= 1e-8


def test_track_optimize_carries_on_after_failed_solves(pt, solver_mode):
    """Non-finite flow components in all four stacks: the frames whose solve fails keep their chained positions, the
    sequence goes on -- ids, lengths, positions and per-solve terminations as in the CPU restatement."""
    from oracle import oracle as orc
    T, H, W, r = 9, 60, 84, 2
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(T, H, W, seed=77, sigma=0.1, n_occluders=1, stride2=True),
                                    seed=78, per_field=6)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert 5 in [s["termination"] for s in O.solves]
    assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves]
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    both = np.isfinite(O.xy)
    assert np.array_equal(np.isfinite(R.xy), both) and float(np.abs(R.xy[both] - O.xy[both]).max()) <= TOL


def test_track_optimize_stalled_solves_are_resumed(pt, monkeypatch):
    """With ONE unrolled iteration per frame every solve of the launch chain runs out of launches: its write-back raises
    the device-side stall flag (everything enqueued behind turns into no-ops), the next checkpoint redoes it with host
    polling and re-enqueues the frames after it.  Same trajectories, same per-solve statistics."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    monkeypatch.setenv("PSFM_SOLVE_UNROLL", "1")
    _hip.context().set_solver(1, 0)
    for (H, W, T, r, seed, sigma) in [(90, 140, 8, 3, 52, 0.3), (64, 96, 21, 1, 54, 0.15)]:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=2, stride2=True)
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        assert max(s["iterations"] for s in O.solves) > 2          # more than init + one launch can do
        R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
        assert float(np.abs(R.xy - O.xy).max()) <= TOL
        assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
        assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves]
    _hip.context().set_solver(0, 0)


def test_fused_solve_is_what_runs_on_clean_sequences(pt):
    """Adaptive mode on a sequence whose solves converge without a rejection: (nearly) every solve is ONE fused launch,
    k settles at accepted steps + 1; with k forced to 1 every solve is redone by the chain.  A noisy sequence (rejected
    steps, dogleg interpolation) goes to the chain after its first window."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    ctx = _hip.context()
    d = psfm_synth.synth_sequence(40, 72, 100, seed=91, sigma=0.03, n_occluders=1, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    clean = [s["dogleg_nonGN"] == 0 and s["iterations"] == s["successful_steps"] + 1 for s in O.solves]
    assert sum(clean) >= len(clean) - 2
    ctx.set_solver(0, 0)
    for _ in range(2):   # (the second run starts from the k the first one learnt)
        R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    cnt = ctx.solver_counters()
    assert cnt["fused"] >= len(O.solves) - 4 and cnt["chain"] == 0, cnt
    assert cnt["k"] == max(s["successful_steps"] for s in O.solves) + 1, cnt
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and float(np.abs(R.xy - O.xy).max()) <= TOL
    assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
    # k forced to 1: the device-paced sequence finishes every solve with continuation launches (no redo) ...
    ctx.set_solver(2, 1)
    R1 = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    cnt1 = ctx.solver_counters()
    assert cnt1["fused"] == len(O.solves) and cnt1["fused_redone"] == 0, cnt1
    assert np.array_equal(R1.xy, R.xy)
    # ... and the host-paced form (PSFM_SEQ=0: one frame kernel per frame) has every one of them redone by the chain
    import os
    os.environ["PSFM_SEQ"] = "0"
    try:
        R2 = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
        cnt2 = ctx.solver_counters()
    finally:
        del os.environ["PSFM_SEQ"]
    assert cnt2["fused"] == 0 and cnt2["fused_redone"] == len(O.solves), cnt2
    assert np.array_equal(R2.xy, R.xy)
    ctx.set_solver(0, 0)
    dn = psfm_synth.synth_sequence(40, 72, 100, seed=92, sigma=0.5, n_occluders=2, stride2=True)
    _, occ = orc.flow_check(dn["flows_f"], dn["flows_b"], 1.0)
    _, occ2 = orc.flow_check(dn["flows_f2"], dn["flows_b2"], 1.0)
    On = orc.track_optimize(dn["flows_f"], dn["flows_f2"], occ, occ2, 2)
    assert sum(s["iterations"] != s["successful_steps"] + 1 or s["dogleg_nonGN"] > 0 for s in On.solves) > len(On.solves) // 2
    Rn = pt.track_optimize(dn["flows_f"], dn["flows_f2"], occ, occ2, 2)
    cntn = ctx.solver_counters()
    assert cntn["chain"] > cntn["fused"] + cntn["fused_redone"], cntn
    assert np.array_equal(Rn.birth, On.birth) and np.array_equal(Rn.length, On.length) and float(np.abs(Rn.xy - On.xy).max()) <= TOL
    assert [s["iterations"] for s in Rn.solve_stats] == [s["iterations"] for s in On.solves]


def test_track_optimize_two_flows_only(pt, solver_mode):
    """n_flows = 2: exactly one solve; n_flows = 1: none (the stride-2 stack is empty)."""
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(3, 40, 60, seed=61, sigma=0.05, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    assert float(np.abs(R.xy - O.xy).max()) <= TOL and len(R.solve_stats) == 1
    O1 = orc.track_optimize(d["flows_f"][:1], [], occ[:1], [], 2)
    R1 = pt.track_optimize(d["flows_f"][:1], [], occ[:1], [], 2)
    assert np.array_equal(R1.birth, O1.birth) and np.array_equal(R1.xy, O1.xy)


@pytest.mark.parametrize("mode,k", [(0, 1), (0, 2), (0, 3), (2, 2), (0, 0)])
@pytest.mark.parametrize("batch", [False, True])
def test_solve_that_outlasts_the_launches_of_its_window(pt, mode, k, batch):
    """A device-paced window has (frames left + 2) launches; a sequence's LAST frame therefore has three -- K + 2 + 2 iterations --
    and a solve whose every step is accepted without terminating for longer than that leaves the window with no frame completed and
    nothing stalled.  The next window's launches must go on with that solve (rounds 2-4 ran the frame again from the top: its newborn
    tracks came out twice -- found by scripts/stress_batch.py, seed 3 batch 86).  The sequence: two flows, one solve of 8 iterations with
    7 accepted steps; also as the tail of a longer sequence and as a batch of one."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    import torch
    d = psfm_synth.synth_sequence(3, 57, 26, seed=410372702, sigma=0.4, n_occluders=2, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    assert [(s["iterations"], s["successful_steps"]) for s in O.solves] == [(8, 7)]
    dev = {k2: torch.from_numpy(np.stack(d[k2])).cuda() for k2 in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    ctx = _hip.batch_contexts(1)[0] if batch else _hip.context()
    ctx.set_solver(mode, k)
    try:
        if batch:
            ctxs, infos = pt.trajectory.run_connect_batch([(dev["flows_f"], dev["flows_b"], dev["flows_f2"], dev["flows_b2"])], 1.0, 2)
            R = pt.trajectory._result_to_host(ctxs[0], infos[0])
        else:
            R = pt.trajectory.run_connect(dev["flows_f"], dev["flows_b"], dev["flows_f2"], dev["flows_b2"], 1.0, 2)
    finally:
        ctx.set_solver(0, 0)
    assert len(R.birth) == O.n_traj == 488 and np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    assert float(np.abs(R.xy - O.xy).max()) <= TOL
    assert [(s["iterations"], s["successful_steps"], s["termination"]) for s in R.solve_stats] == \
           [(s["iterations"], s["successful_steps"], s["termination"]) for s in O.solves]


def test_lane_table_full_inside_a_device_paced_window(pt):
    """A sequence that outgrows its lane table (a 17 x 15 grid whose lanes creep past twice the grid in 24 frames) inside a device-paced
    window: the fused launch behind that point counts on blocks the grid does not have -- it never sees its last arrival, raises no stall
    flag and left its tickets mid-count for every later sequence of the context (found by scripts/stress_batch.py, seed 54).  The
    checkpoint now ends such a run with PSFM_ERR_CAPACITY (run_connect raises the capacity and runs it again) and every sequence starts
    with its tickets at zero: the sequence equals the oracle, and so does the next one on the same context."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    import ctypes
    import torch
    ctx = _hip.context()
    for seed, T in ((88003367, 25), (494952463, 7)):
        d = psfm_synth.synth_sequence(T, 34, 30, seed=seed, sigma=0.03, n_occluders=0, stride2=True)
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 3.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 3.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
        dev = {k2: torch.from_numpy(np.stack(d[k2])).cuda() for k2 in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
        if T == 25:      # the call itself, with the default tables: a capacity error, not a solver error and not a wrong result
            ctx.set_capacity(2.0, 8.0)
            info = _hip.TrackInfo()
            st = _hip.lib().psfm_connect(ctx.handle, _hip.ptr(dev["flows_f"]), _hip.ptr(dev["flows_b"]), _hip.ptr(dev["flows_f2"]),
                                         _hip.ptr(dev["flows_b2"]), T - 1, 34, 30, 3.0, 2, None, None, ctypes.byref(info),
                                         _hip.current_stream_ptr(ctx.device))
            assert st == _hip.PSFM_ERR_CAPACITY, (st, _hip.lib().psfm_last_error())
        R = pt.trajectory.run_connect(dev["flows_f"], dev["flows_b"], dev["flows_f2"], dev["flows_b2"], 3.0, 2)
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and float(np.abs(R.xy - O.xy).max()) <= TOL
        assert [(q["iterations"], q["termination"]) for q in R.solve_stats] == [(q["iterations"], q["termination"]) for q in O.solves]


def test_track_optimize_full_size_properties(pt):
    """configs[3]/[4] shapes with fewer frames: 1080p r=2 and 480x640 r=1 (dense), full path-consistency optimise.
    Size-independent invariants + the first frames against the oracle."""
    import torch
    from oracle import oracle as orc
    for (H, W, T, r, k) in [(1080, 1920, 10, 2, 4), (480, 640, 40, 1, 4)]:
        d = psfm_synth.synth_sequence_torch(T, H, W, seed=5, sigma=0.05, n_occluders=2, stride2=True)
        R = pt.trajectory.run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], 1.0, r)
        _, occ = pt.utils.flow_check_device(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = pt.utils.flow_check_device(d["flows_f2"], d["flows_b2"], 1.0)
        R_seq = pt.trajectory.run_track(d["flows_f"], occ, d["flows_f2"], occ2, r)
        # the pipelined stage entry (psfm_connect) and the two-call form give the same bits
        assert np.array_equal(R.birth, R_seq.birth) and np.array_equal(R.length, R_seq.length) and np.array_equal(R.xy, R_seq.xy)
        GW, GH = (W + r - 1) // r, (H + r - 1) // r
        last = R.birth + R.length - 1
        assert R.off[0] == 0 and np.array_equal(np.diff(R.off), R.length) and R.off[-1] == R.n_points
        assert int((R.birth == 0).sum()) == GW * GH and last.max() == T - 1 and (np.diff(last) >= 0).all()
        assert len(R.solve_stats) == T - 2 and all(s["termination"] in (0, 1, 2) for s in R.solve_stats)
        assert np.isfinite(R.xy).all()
        ff, f2 = list(d["flows_f"][:k].cpu().numpy()), list(d["flows_f2"][:k - 1].cpu().numpy())
        oo, o2 = list(occ[:k].cpu().numpy()), list(occ2[:k - 1].cpu().numpy())
        O = orc.track_optimize(ff, f2, oo, o2, r)
        Rk = pt.trajectory.run_track(d["flows_f"][:k], occ[:k], d["flows_f2"][:k - 1], occ2[:k - 1], r)
        assert np.array_equal(Rk.birth, O.birth) and np.array_equal(Rk.length, O.length)
        assert float(np.abs(Rk.xy - O.xy).max()) <= TOL
        assert [s["iterations"] for s in Rk.solve_stats] == [s["iterations"] for s in O.solves]
        del d, R, R_seq
        torch.cuda.empty_cache()


# (the dense 436 x 640 grid: 545 tracks per block -- with ONE slot per thread more than half of them are streamed behind the slots)
@pytest.mark.parametrize("H,W,T,r,seed", [(120, 200, 9, 2, 61), (90, 140, 8, 3, 62), (436, 1024, 6, 2, 63), (436, 640, 5, 1, 64)])
def test_launch_chain_as_one_persistent_launch_is_the_same_solve(pt, monkeypatch, H, W, T, r, seed):
    """Solves that reject steps run the launch chain; with the device to itself the chain's loop is ONE persistent launch
    (psfm_pc_resident_kernel: the tracks' state on chip, a two-hop all-reduce per trust-region round) instead of one launch per iteration.  Both forms
    run the same per-track code, the same reduction order and the same control step: identical bits -- and the oracle's
    decisions.  Hard flows (sigma 0.3, 5 % occluders): every solve takes 20-40 iterations with rejections and dogleg steps."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert sum(s["iterations"] - s["successful_steps"] for s in O.solves) > T and sum(s["dogleg_nonGN"] for s in O.solves) > T
    ctx = _hip.context()
    ctx.set_solver(1, 0)            # the launch chain for every solve
    try:
        res = {}
        for persist in ("1", "0"):
            monkeypatch.setenv("PSFM_PC_PERSIST", persist)
            res[persist] = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        A, B = res["1"], res["0"]
        # the resident launch behind a separate iteration-0 launch, and with ONE slot per thread (the rest of a block's tracks
        # streamed behind the slots): the same sums in the same order
        for env, val in (("PSFM_PC_INIT_INSIDE", "0"), ("PSFM_PC_SLOTS", "1")):
            monkeypatch.setenv("PSFM_PC_PERSIST", "1")
            monkeypatch.setenv(env, val)
            V = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            monkeypatch.delenv(env)
            assert np.array_equal(V.birth, A.birth) and np.array_equal(V.length, A.length) and float(np.abs(V.xy - A.xy).max()) <= 1e-9
            if H * W < 100000:
                assert np.array_equal(V.xy, A.xy), env
            assert [s["iterations"] for s in V.solve_stats] == [s["iterations"] for s in A.solve_stats]
        # (bit-equal where the lanes are placed deterministically; on larger grids births pop lanes from the shared stacks in
        # atomic order, which permutes the partial sums of two runs of the SAME form as well)
        assert np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length)
        assert float(np.abs(A.xy - B.xy).max()) <= 1e-9
        if H * W < 100000:
            assert np.array_equal(A.xy, B.xy)
        keys = ("iterations", "successful_steps", "termination", "dogleg_nonGN")
        assert [[s[k] for k in keys] for s in A.solve_stats] == [[s[k] for k in keys] for s in B.solve_stats]
        assert np.array_equal(A.birth, O.birth) and np.array_equal(A.length, O.length)
        assert float(np.abs(A.xy - O.xy).max()) <= TOL
        assert [s["iterations"] for s in A.solve_stats] == [s["iterations"] for s in O.solves]
        assert [s["termination"] for s in A.solve_stats] == [s["termination"] for s in O.solves]
        # the batch entry point too
        uv, ref1, ref2, scale, flow12 = _batch(60, 80, 3000, 2, 0.5)
        outs = {}
        for persist in ("1", "0"):
            monkeypatch.setenv("PSFM_PC_PERSIST", persist)
            outs[persist] = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, uv.shape[0], 80, 60)
            outs[persist + "s"] = dict(pt.particlesfm.optimize_location.last_stats)
        assert np.array_equal(outs["1"], outs["0"]) and outs["1s"] == outs["0s"] and outs["1s"]["iterations"] > outs["1s"]["successful_steps"]
    finally:
        ctx.set_solver(0, 0)


def test_persistent_solve_that_gives_up_is_redone_with_launches(pt, monkeypatch):
    """A persistent solve whose barrier times out (PSFM_PC_SPIN=0: every block but the last arriver of round 1 leaves at once -- what
    happens when the grid is not co-resident) leaves the control block "not done"; the write-back kernel behind it raises the stall
    flag, the host's checkpoint redoes the solve (the batch entry point: its polling loop carries on with launches).  Same result."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    T, H, W, r = 7, 120, 200, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=71, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    ctx = _hip.context()
    ctx.set_solver(1, 0)
    try:
        monkeypatch.setenv("PSFM_PC_SPIN", "0")
        R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and float(np.abs(R.xy - O.xy).max()) <= TOL
        assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
        uv, ref1, ref2, scale, flow12 = _batch(60, 80, 3000, 2, 0.5)
        out_o, st_o = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
        out_g = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, uv.shape[0], 80, 60)
        st_g = pt.particlesfm.optimize_location.last_stats
        assert st_g["iterations"] == st_o["iterations"] and st_g["termination"] == st_o["termination"]
        assert float(np.abs(out_g - out_o).max()) <= 1e-8
    finally:
        ctx.set_solver(0, 0)


@pytest.mark.parametrize("quit", ["40,2", "5,3", "0,1"])
def test_resident_solve_with_one_block_giving_up_mid_solve(pt, monkeypatch, quit):
    """PSFM_PC_QUIT="block,round": that block gives up in that round of every resident solve (what a grid that is not co-resident
    looks like to the others) -- a member (block 40), a leader of the all-reduce (block 5), block 0.  It poisons its granules, its
    leader passes the poison on, every block leaves at its next poll without having written anything and raises the stall flag;
    the host redoes the solve with launches (and stops trying the resident form after two give-ups).  Same bits as launches only."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    T, H, W, r = 7, 120, 200, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=72, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    ctx = _hip.context()
    ctx.set_solver(1, 0)
    try:
        monkeypatch.setenv("PSFM_PC_PERSIST", "0")
        ref = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        monkeypatch.setenv("PSFM_PC_PERSIST", "1")
        monkeypatch.setenv("PSFM_PC_QUIT", quit)
        R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        assert np.array_equal(R.birth, ref.birth) and np.array_equal(R.length, ref.length) and np.array_equal(R.xy, ref.xy)
        assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in ref.solve_stats]
        assert max(s["iterations"] for s in ref.solve_stats) > int(quit.split(",")[1]) + 1      # (the round is reached)
        uv, ref1, ref2, scale, flow12 = _batch(60, 80, 3000, 2, 0.5)
        # (the batch of 3000 tracks runs on 12 blocks)
        monkeypatch.setenv("PSFM_PC_QUIT", ("3," if int(quit.split(",")[0]) >= 12 else quit.split(",")[0] + ",") + quit.split(",")[1])
        out_q = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, uv.shape[0], 80, 60)
        st_q = dict(pt.particlesfm.optimize_location.last_stats)
        monkeypatch.delenv("PSFM_PC_QUIT")
        monkeypatch.setenv("PSFM_PC_PERSIST", "0")
        out_l = pt.particlesfm.optimize_location(uv, ref1, ref2, scale, flow12, uv.shape[0], 80, 60)
        assert np.array_equal(out_q, out_l) and st_q == dict(pt.particlesfm.optimize_location.last_stats)
    finally:
        ctx.set_solver(0, 0)


@pytest.mark.parametrize("env,val", [("PSFM_PC_SPIN", "0"), ("PSFM_PC_QUIT", "0,1"), ("PSFM_PC_QUIT", "5,2")])
def test_redo_of_a_fused_solve_survives_a_resident_launch_that_gives_up(pt, monkeypatch, env, val):
    """The DEFAULT adaptive path on flows whose solves reject steps: the fused solve of a frame stalls at the first rejection and
    the host redoes it (psfm_solve_frame_resume, chain_stalled = false) with iteration 0 + the resident launch, polling behind it.
    When that launch gives up (spin limit 0 / one block quitting: a grid that is not co-resident) it must leave the stall flag
    alone -- raised, every pc_iter launch of the polling loop would return at once and the call would end in PSFM_ERR_SOLVER
    ("did not terminate") -- and the launches take the solve from iteration 0.  Same results as the oracle."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    T, H, W, r = 7, 120, 200, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=73, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert sum(s["iterations"] - s["successful_steps"] for s in O.solves) > 0
    ctx = _hip.context()
    ctx.set_solver(0, 0)
    # (the adaptive mode remembers what the last window of the context's previous sequence looked like: a clean sequence first, so
    # that this one starts with the fused solve)
    c = psfm_synth.synth_sequence(5, H, W, seed=74, sigma=0.02, n_occluders=0, stride2=True)
    _, co = orc.flow_check(c["flows_f"], c["flows_b"], 1.0)
    _, co2 = orc.flow_check(c["flows_f2"], c["flows_b2"], 1.0)
    pt.track_optimize(c["flows_f"], c["flows_f2"], co, co2, r)        # (may itself run as the chain; its clean window brings the fused solve back)
    monkeypatch.setenv("PSFM_PC_PERSIST", "1")
    monkeypatch.setenv(env, val)
    R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    cnt = ctx.solver_counters()
    assert cnt["fused_redone"] >= 1, cnt        # (the path under test ran)
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and float(np.abs(R.xy - O.xy).max()) <= TOL
    assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
    assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves]


def test_resident_solves_of_several_threads_side_by_side(pt):
    """psfm_ctx_set_resident_budget: four host threads, each with its own context and a quarter of the device's co-resident block
    slots, run sequences whose solves reject steps at the same time -- every solve as ONE resident launch (shared gate) instead of
    the launch chain.  Same results as the oracle; the counters say the resident form ran."""
    import threading
    from oracle import oracle as orc
    from point_trajectory import _hip
    H, W, r, n_thr = 120, 200, 2, 4
    data, want = [], []
    for k in range(n_thr):
        d = psfm_synth.synth_sequence(8 + k, H, W, seed=85 + k, stride2=True, **psfm_synth.HARD)
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        data.append((d, occ, occ2))
        want.append(orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r))
    res, err, cnts = {}, [], {}

    def run(k):
        import torch
        try:
            torch.cuda.set_device(0)
            ctx = _hip.context()
            ctx.set_chain_mode(1)
            ctx.set_resident_budget(ctx.resident_capacity() // n_thr)
            d, occ, occ2 = data[k]
            for _ in range(2):
                res[k] = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            cnts[k] = ctx.solver_counters()
        except Exception as e:      # noqa: BLE001
            err.append(e)
        finally:
            _hip.release_thread_contexts()

    ths = [threading.Thread(target=run, args=(k,)) for k in range(n_thr)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not err, err
    for k in range(n_thr):
        R, O = res[k], want[k]
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and float(np.abs(R.xy - O.xy).max()) <= TOL
        assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
        assert cnts[k]["resident_launches"] > 0 and cnts[k]["resident_giveups"] <= 1, cnts[k]


def test_exclusive_sequence_lets_other_host_threads_in(pt):
    """A track_optimize call takes the device gate exclusively when it is free (its hard solves then run as resident launches);
    a psfm call of another host thread that arrives meanwhile announces itself and is let in at the sequence's next checkpoint
    (PsfmGate::yield_exclusive: the rest of the sequence runs its solves as launches).  Same results as when run alone."""
    import threading
    import time
    from oracle import oracle as orc
    from point_trajectory.track import track
    T, H, W, r = 41, 120, 200, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=81, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    alone_a = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    alone_b = track(d["flows_f"], occ, r)
    res, err = {}, []

    def run_a():
        import torch
        from point_trajectory import _hip
        try:
            torch.cuda.set_device(0)
            res["a"] = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        except Exception as e:      # noqa: BLE001
            err.append(e)
        finally:
            _hip.release_thread_contexts()

    def run_b():
        import torch
        from point_trajectory import _hip
        try:
            torch.cuda.set_device(0)
            time.sleep(0.004)
            res["b"] = [track(d["flows_f"], occ, r) for _ in range(3)]
        except Exception as e:      # noqa: BLE001
            err.append(e)
        finally:
            _hip.release_thread_contexts()

    for _ in range(3):
        ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
        ta.start(); tb.start(); ta.join(); tb.join()
        assert not err, err
        A = res["a"]
        assert np.array_equal(A.birth, alone_a.birth) and np.array_equal(A.length, alone_a.length) and np.array_equal(A.xy, alone_a.xy)
        assert [s["iterations"] for s in A.solve_stats] == [s["iterations"] for s in alone_a.solve_stats]
        for B in res["b"]:
            assert np.array_equal(B.birth, alone_b.birth) and np.array_equal(B.xy, alone_b.xy)



def test_budget_below_the_solves_block_count(pt):
    """psfm_ctx_set_resident_budget with a budget BELOW the solve's own block count (lane capacity / 256: ~870 blocks at 436 x 1024,
    sample_ratio 2): the solver's launches shrink to the budget, so the f64 sums over the tracks are grouped by other blocks than in
    the unbudgeted run (psfm.h used to promise identical results).  What holds, against the unbudgeted run AND the oracle:
    ids, lengths and every decision of every solve (iterations, accepted steps, terminations) equal; positions to rounding."""
    from oracle import oracle as orc
    from point_trajectory import _hip
    H, W, r, T = 436, 1024, 2, 7
    d = psfm_synth.synth_sequence(T, H, W, seed=91, stride2=True, **psfm_synth.HARD)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    ctx = _hip.context()
    runs = {}
    try:
        for budget in (0, 128, 48):
            ctx.set_resident_budget(budget)
            R = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            runs[budget] = (R, ctx.solver_counters())
    finally:
        ctx.set_resident_budget(0)
    assert ((W + r - 1) // r) * ((H + r - 1) // r) * 2 // 256 > 128          # the budgets really are below the block count
    R0 = runs[0][0]
    for budget, (R, cnt) in runs.items():
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
        for key in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
            assert [s[key] for s in R.solve_stats] == [s[key] for s in O.solves], (budget, key)
        assert float(np.abs(R.xy - O.xy).max()) <= TOL
        assert float(np.abs(R.xy - R0.xy).max()) <= 1e-9, budget
        assert cnt["resident_launches"] > 0, (budget, cnt)
