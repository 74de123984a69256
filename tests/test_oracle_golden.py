"""Pin the CPU oracle (oracle/psfm_oracle.c) against vectors produced by the reference's own
Python (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc
from _common import solver_batch, SOLVER_BATCHES, LARGE_MOTION, REALISTIC_OPT, REALISTIC_TRACK, golden, regen_inputs, assert_csr_equal
import psfm_synth


def test_sampler_bit_exact():
    g = golden("sampler")
    rng = np.random.default_rng(int(g["seed"]))
    H, W = int(g["H"]), int(g["W"])
    m2 = rng.standard_normal((H, W, 2)).astype(np.float32)
    m1 = (rng.uniform(size=(H, W)) < 0.3)
    s2 = orc.grid_sample(m2, g["pts"])
    s1 = orc.grid_sample(m1.astype(np.float32), g["pts"])
    assert np.array_equal(s2.view(np.uint32), g["s2"].view(np.uint32))
    assert np.array_equal(s1.view(np.uint32), g["s1"].view(np.uint32))


def test_sampler_1080p_bit_exact():
    g = golden("sampler")
    rng = np.random.default_rng(int(g["seed"]))
    H, W = int(g["H"]), int(g["W"])
    m2 = rng.standard_normal((H, W, 2)).astype(np.float32)
    m1 = (rng.uniform(size=(H, W)) < 0.3)
    # the generator consumed the point draws between the maps: replay them
    rng.uniform([-3, -3], [W + 2, H + 2], size=(4000, 2))
    Hb, Wb = int(g["Hb"]), int(g["Wb"])
    mb = rng.standard_normal((Hb, Wb, 2)).astype(np.float32)
    assert hashlib.sha256(m2.tobytes() + m1.tobytes() + mb.tobytes()).hexdigest() == str(g["map_hash"])
    sb = orc.grid_sample(mb, g["pb"])
    assert np.array_equal(sb.view(np.uint32), g["sb"].view(np.uint32))


def test_flow_check_bit_exact():
    g = golden("flow_check")
    d = psfm_synth.synth_sequence(4, 64, 96, seed=21, sigma=0.4, n_occluders=2, stride2=False)
    for thres in (1.0, 3.0):
        err, occ = orc.flow_check(d["flows_f"], d["flows_b"], thres)
        assert np.array_equal(np.stack(err).view(np.uint32), g["fc_err_%g" % thres].view(np.uint32))
        assert np.array_equal(np.packbits(np.stack(occ)), g["fc_occ_%g" % thres])
    dd = psfm_synth.synth_sequence(3, 40, 56, seed=22, amp=9.0, sigma=0.0, stride2=False)
    err, occ = orc.flow_check(dd["flows_f"], dd["flows_b"], 1.0)
    assert np.array_equal(np.stack(err).view(np.uint32), g["big_err"].view(np.uint32))
    assert np.array_equal(np.packbits(np.stack(occ)), g["big_occ"])
    assert 0.02 < np.stack(occ).mean() < 0.98


@pytest.mark.parametrize("name", ["track_48x64_r2", "track_45x70_r1", "track_50x66_r3", "track_52x61_r4",
                                  "track_largemotion_80x120_r2", "track_largemotion_75x110_r1"] + REALISTIC_TRACK)   # (~10 px of drift per frame; layered scene)
def test_track_bit_exact(name):
    g = golden(name)
    d = regen_inputs(g, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = orc.track(d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)


def test_track_all_tracks_die():
    """SciPy's EDT with no background pixel: the reference respawns every grid point but (0,0)."""
    g = golden("track_alldie_24x30_r2")
    d = regen_inputs(g, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    occ = [o.copy() for o in occ]
    occ[1][:] = True
    R = orc.track(d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)


def _nonfinite_inputs(g):
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(int(g["T"]), int(g["H"]), int(g["W"]), seed=int(g["seed"]),
                                                              sigma=float(g["sigma"]), n_occluders=int(g["n_occluders"]),
                                                              stride2=False), seed=int(g["seed"]) + 1)
    from _common import input_hash
    assert input_hash(d) == str(g["input_hash"])
    return d


def test_nonfinite_flows_bit_exact():
    """NaN / +-Inf / huge flow components (fixture made by the reference's torch ops): NaN errors stay NaN and compare
    false, tracks stepping to a non-finite position end there; ids, lengths, positions equal."""
    g = golden("nonfinite_40x56_r2")
    d = _nonfinite_inputs(g)
    err, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    assert np.array_equal(np.stack(err).view(np.uint32), g["fc_err"].view(np.uint32))
    assert np.isnan(g["fc_err"]).sum() > 50
    assert np.array_equal(np.packbits(np.stack(occ)), g["fc_occ"])
    R = orc.track(d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)
    assert np.isfinite(R.xy).all()


@pytest.mark.parametrize("name", ["opt_48x64_r2", "opt_45x70_r3"])
def test_track_optimize_orchestration(name):
    """Buffer / index / scale semantics are the reference's (pinned); the solver iterate inside is the
    C restatement on both sides (parity unpinned w.r.t. real Ceres)."""
    g = golden(name)
    d = regen_inputs(g, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    R = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)
    assert len(R.solves) == int(g["T"]) - 2


@pytest.mark.parametrize("name", LARGE_MOTION)
def test_track_optimize_large_motion(name):
    """The 20 px gate of `loss02_scale` (trajectory.py:179) from both sides and fractional occ02 weights: the fixture is the
    reference's own Python on flows drifting ~10 px per frame, and records how often each branch was taken."""
    g = golden(name)
    assert int(g["gate_closed"]) > 1000 and int(g["gate_open"]) > 1000 and int(g["scale_fractional"]) > 100
    d = regen_inputs(g, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    assert np.array_equal(np.packbits(np.stack(occ)), g["occ"]) and np.array_equal(np.packbits(np.stack(occ2)), g["occ2"])
    R = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)
    assert len(R.solves) == int(g["T"]) - 2


@pytest.mark.parametrize("name", REALISTIC_OPT)
def test_track_optimize_realistic(name):
    """psfm_synth.REALISTIC -- depth-ordered layers with true (dis)occlusion, correlated flow error, outlier blobs -- through the
    reference's own flow_check + track_optimize: masks of both strides, ids, lengths, positions; and the fixture says what the solver
    went through (every solve rejects steps at the motion boundaries: the launch chain's / the resident solve's case)."""
    g = golden(name)
    d = regen_inputs(g, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    assert np.array_equal(np.packbits(np.stack(occ)), g["occ"]) and np.array_equal(np.packbits(np.stack(occ2)), g["occ2"])
    assert 0.05 < float(g["occluded"]) < 0.3 and int(g["ended_early"]) > 1000
    R = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)
    assert [s["iterations"] for s in R.solves] == list(g["solve_iterations"])
    assert sum(s["iterations"] - s["successful_steps"] for s in R.solves) > 20


def test_solver_properties():
    """Known-answer style checks of the restated Ceres loop that need no Ceres."""
    rng = np.random.default_rng(5)
    H, W, n = 30, 40, 500
    # constant flow: the objective is an exact linear least squares -> one Gauss-Newton step solves it
    flow = np.zeros((H, W, 2), np.float32)
    flow[..., 0], flow[..., 1] = 1.5, -0.5
    p0 = rng.uniform([5, 5], [W - 6, H - 6], size=(n, 2))
    ref1 = p0 + [1.5, -0.5]
    ref2 = p0 + [3.0, -1.0]
    uv = np.concatenate([ref1, ref2], 1) + rng.normal(0, 0.3, size=(n, 4))
    s = np.ones(n)
    out, st = orc.optimize_location(uv, ref1, ref2, s, flow, return_stats=True)
    # (the dogleg's mu=1e-8 regularisation leaves a ~1e-8 relative remainder per step)
    assert np.abs(out - np.concatenate([ref1, ref2], 1)).max() < 1e-6
    assert st["successful_steps"] >= 1 and st["final_cost"] < 1e-12 * st["initial_cost"]
    # scale = 0 removes the stride-2 term; p1 -> ref1, p2 -> p1 + flow
    out0 = orc.optimize_location(uv, ref1, ref2 + 7.0, np.zeros(n), flow)
    assert np.abs(out0[:, :2] - ref1).max() < 1e-6
    assert np.abs(out0[:, 2:] - (ref1 + [1.5, -0.5])).max() < 1e-6
    # already optimal input: terminates immediately, returns the input bits
    exact = np.concatenate([ref1, ref2], 1)
    out1, st1 = orc.optimize_location(exact, ref1, ref2, s, flow, return_stats=True)
    assert np.abs(out1 - exact).max() < 1e-12


@pytest.mark.parametrize("H,W,n,seed,sigma,kink", SOLVER_BATCHES)
def test_second_restatement_agrees(H, W, n, seed, sigma, kink):
    """oracle/ceres_tr_numpy.py -- Ceres 2.0.0's trust-region loop restated a second time, component by component from
    the library's own structure, with different arithmetic (dense einsum Jacobians, LAPACK Cholesky, pairwise sums) --
    against the C oracle: every decision equal (iterations, accepted steps, termination, dogleg case count), positions
    within 1e-9 px.  Two separately written restatements agreeing is the evidence available until real Ceres pins them
    (tests/test_ref_ceres.py)."""
    from oracle import ceres_tr_numpy as ct
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    out_c, st_c = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
    out_n, st_n = ct.optimize_location(uv, ref1, ref2, scale, flow12, n, W, H)
    for k in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
        assert st_c[k] == st_n[k], (k, st_c, st_n)
    assert abs(st_c["initial_cost"] - st_n["initial_cost"]) <= 1e-12 * max(1.0, st_c["initial_cost"])
    assert abs(st_c["final_cost"] - st_n["final_cost"]) <= 1e-9 * max(1.0, st_c["final_cost"])
    assert float(np.abs(out_c - out_n).max()) <= 1e-9


def test_second_restatement_agrees_on_failure_and_trivial_cases():
    from oracle import ceres_tr_numpy as ct
    uv, ref1, ref2, scale, flow12 = solver_batch(60, 80, 3000, 7, 0.05)
    bad = flow12.copy()
    bad[20:30, 30:50, :] = np.nan
    out_c, st_c = orc.optimize_location(uv, ref1, ref2, scale, bad, return_stats=True)
    out_n, st_n = ct.optimize_location(uv, ref1, ref2, scale, bad, uv.shape[0], 80, 60)
    assert st_c["termination"] == 5 and st_n["termination"] == 5          # FAILURE: parameters come back untouched
    assert st_c["iterations"] == st_n["iterations"] == 0                   # in IterationZero: residual_block.cc IsEvaluationValid
    assert np.array_equal(out_c, uv) and np.array_equal(out_n, uv)
    # constant flow = linear least squares; already-optimal input
    H, W, n = 30, 40, 200
    rng = np.random.default_rng(8)
    flow = np.zeros((H, W, 2), np.float32)
    flow[..., 0], flow[..., 1] = 1.5, -0.5
    p0 = rng.uniform([5, 5], [W - 6, H - 6], size=(n, 2))
    r1, r2 = p0 + [1.5, -0.5], p0 + [3.0, -1.0]
    x0 = np.concatenate([r1, r2], 1) + rng.normal(0, 0.3, size=(n, 4))
    for start in (x0, np.concatenate([r1, r2], 1)):
        oc, sc = orc.optimize_location(start, r1, r2, np.ones(n), flow, return_stats=True)
        on, sn = ct.optimize_location(start, r1, r2, np.ones(n), flow, n, W, H)
        assert sc["iterations"] == sn["iterations"] and sc["termination"] == sn["termination"], (sc, sn)
        assert float(np.abs(oc - on).max()) <= 1e-10


def test_reference_python_with_the_second_restatement_as_solver():
    """The reference's own track_optimize (track_optimize.py:24-53, trajectory.py:161-194), unmodified, with the NumPy
    restatement as `particlesfm.optimize_location`, against the C oracle's whole-sequence result: ids / lengths equal,
    positions within 1e-9 px.  (The committed opt_* fixtures were produced with the C restatement in that seat.)"""
    from oracle import ref_shim
    from oracle import ceres_tr_numpy as ct
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    ref = ref_shim.load_reference(optimize_location=lambda *a: ct.optimize_location(*a)[0])
    d = psfm_synth.synth_sequence(7, 60, 84, seed=91, sigma=0.25, n_occluders=2, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    birth, length, off, xy = ref_shim.trajs_to_csr(full)
    assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
    assert float(np.abs(xy - O.xy).max()) <= 1e-9


def test_ceres_assumption_switches_are_mirrored_in_both_restatements():
    """The places where the restatements rest on memory of trust_region_minimizer.cc / solver.cc are switches
    (psfm_oracle.c orc_set_variant, ceres_tr_numpy.VARIANTS); tests/test_ref_ceres.py walks them against a real Ceres build when
    one exists.  Here: under every setting of every switch the two restatements still agree with each other, and the switches
    do what they say on batches built to hit them."""
    import itertools
    from oracle import oracle as orc
    from oracle import ceres_tr_numpy as ct
    from _common import solver_batch
    batches = [solver_batch(40, 50, 300, 3, 0.3, False), solver_batch(40, 50, 200, 4, 0.05, True)]
    # tracks that sit in the optimum already (zero field, refs = positions): gradient 0 at iteration 0
    n = 50
    rng = np.random.default_rng(0)
    p1 = rng.uniform(5, 30, (n, 2)); uv0 = np.concatenate([p1, p1], 1)
    batches.append((uv0, p1.copy(), p1.copy(), np.ones((n, 1)), np.zeros((40, 50, 2), np.float32)))
    # a NaN patch under some tracks: FAILURE in IterationZero
    uvn, r1n, r2n, scn, fln = solver_batch(40, 50, 100, 9, 0.1, False)
    fln = fln.copy(); fln[10:14, 20:24] = np.nan
    uvn[:, 0] = np.clip(uvn[:, 0], 20.2, 22.8); uvn[:, 1] = np.clip(uvn[:, 1], 10.2, 12.8)
    batches.append((uvn, r1n, r2n, scn, fln))
    seen = {}
    try:
        for key in ct.VARIANT_KEYS:
            for val in ct.VARIANT_VALUES[key]:
                orc.set_variant(key, val); ct.VARIANTS[key] = val
                for bi, (uv, r1, r2, sc, fl) in enumerate(batches):
                    a, sa = orc.optimize_location(uv, r1, r2, sc, fl, return_stats=True)
                    b, sb = ct.optimize_location(uv, r1, r2, sc, fl, len(uv), fl.shape[1], fl.shape[0])
                    for k in ("iterations", "successful_steps", "termination"):
                        assert sa[k] == sb[k], (key, val, bi, k, sa, sb)
                    fin = np.isfinite(a) & np.isfinite(b)
                    assert np.array_equal(np.isfinite(a), np.isfinite(b)) and float(np.abs(a[fin] - b[fin]).max()) <= 1e-9
                    seen[(key, val, bi)] = (sa["iterations"], sa["termination"], a.copy())
                orc.set_variant(key, 0); ct.VARIANTS[key] = 0
    finally:
        for key in ct.VARIANT_KEYS:
            orc.set_variant(key, 0); ct.VARIANTS[key] = 0
    # iteration 0 in the optimum: gradient convergence before the first iteration by default, one iteration otherwise
    assert seen[("iter0_successful", 0, 2)][:2] == (0, 2) and seen[("iter0_successful", 1, 2)][0] > 0      # (five invalid steps: no decrease to model)
    # FAILURE: the input handed back by default
    assert seen[("failure_returns", 0, 3)][1] == 5 and np.array_equal(seen[("failure_returns", 0, 3)][2], uvn)

