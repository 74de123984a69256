"""Hook for pinning the Ceres-defined part of the path against the REAL reference module.

Skipped unless PSFM_REF_PARTICLESFM_SO points at a built point_trajectory/optimize/build/particlesfm*.so of the
reference (recipe: oracle/_ref/BUILD.md; impossible in the build image -- no Ceres / Eigen / glog, no network).  With it,
the batches of tests/test_gpu_solver.py go through the real `optimize_location` (optimize/src/bindings.cc:31 ->
trajectory_optimize.cpp:30-96, Ceres 2.0.0) and are compared with the C oracle, the second (NumPy) restatement and, on
a GPU box, libpsfm_hip: <= 1e-4 px (north_star), iteration-level decisions reported.
"""
import importlib.machinery
import importlib.util
import os
import sys

import numpy as np
import pytest

from _common import solver_batch, SOLVER_BATCHES
import psfm_synth

SO = os.environ.get("PSFM_REF_PARTICLESFM_SO", "")
if not SO:      # a module dropped into oracle/_ref/ (BUILD.md step 3) is found without the variable
    import glob
    _found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "particlesfm*.so")))
    SO = _found[0] if _found else ""
pytestmark = pytest.mark.skipif(not (SO and os.path.isfile(SO)),
                                reason="real reference module not available (set PSFM_REF_PARTICLESFM_SO, see oracle/_ref/BUILD.md)")
TOL = 1e-4


def _real_module():
    loader = importlib.machinery.ExtensionFileLoader("particlesfm", SO)
    spec = importlib.util.spec_from_file_location("particlesfm", SO, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("H,W,n,seed,sigma,kink", SOLVER_BATCHES)
def test_real_ceres_vs_both_restatements(H, W, n, seed, sigma, kink):
    from oracle import oracle as orc
    from oracle import ceres_tr_numpy as ct
    real = _real_module()
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    out_r = np.asarray(real.optimize_location(uv, ref1, ref2, scale, flow12.astype(np.float64), n, W, H))
    out_c = orc.optimize_location(uv, ref1, ref2, scale, flow12)
    out_n, _ = ct.optimize_location(uv, ref1, ref2, scale, flow12, n, W, H)
    assert float(np.abs(out_r - out_c).max()) <= TOL
    assert float(np.abs(out_r - out_n).max()) <= TOL


def test_real_ceres_on_the_hand_derived_cases():
    """tests/test_ceres_hand_cases.py's three solves through the real module: the derived positions (the summary is not returned by
    trajectory_optimize.cpp:84-95, so iterations / termination cannot be read -- the positions after 2 / 5 / 0 iterations can)."""
    import test_ceres_hand_cases as hc
    real = _real_module()

    def engine(uv12, ref1, ref2, scale, flow):
        n = len(uv12)
        out = np.asarray(real.optimize_location(uv12, ref1, ref2, scale, flow.astype(np.float64), n, flow.shape[1], flow.shape[0]))
        _, st = hc.run_numpy(uv12, ref1, ref2, scale, flow)      # (decisions: from the restatement, positions: from real Ceres)
        return out, st

    for near in (False, True):
        hc.test_linear_problem_one_gauss_newton_step_then_a_tolerance(engine, near)
    for n in (1, 4):
        hc.test_radius_binds_three_times_then_gauss_newton(engine, n)
    hc.test_zero_gradient_start_ends_at_iteration_zero(engine)


def test_which_reading_of_ceres_the_real_module_agrees_with(capsys):
    """Every combination of the switches of oracle/ceres_tr_numpy.VARIANTS (= psfm_oracle.c orc_set_variant: the places the
    restatements rest on memory of Ceres' sources) against the real module on the solver batches: prints the table of max |dx|
    per combination, and requires the SHIPPED reading (all zeros) to be among the combinations that agree to 1e-9 px."""
    import itertools
    import json
    from oracle import oracle as orc
    from oracle import ceres_tr_numpy as ct
    real = _real_module()
    rows = {}
    try:
        for combo in itertools.product(*(ct.VARIANT_VALUES[k] for k in ct.VARIANT_KEYS)):
            for k, v in zip(ct.VARIANT_KEYS, combo):
                orc.set_variant(k, v); ct.VARIANTS[k] = v
            worst = 0.0
            for (H, W, n, seed, sigma, kink) in SOLVER_BATCHES:
                uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
                out_r = np.asarray(real.optimize_location(uv, ref1, ref2, scale, flow12.astype(np.float64), n, W, H))
                out_c = orc.optimize_location(uv, ref1, ref2, scale, flow12)
                worst = max(worst, float(np.abs(out_r - out_c).max()))
            rows["".join(str(v) for v in combo)] = worst
    finally:
        for k in ct.VARIANT_KEYS:
            orc.set_variant(k, 0); ct.VARIANTS[k] = 0
    with capsys.disabled():
        print("\nreal Ceres vs the C oracle, max |dx| px per reading %s:\n%s" % ("/".join(ct.VARIANT_KEYS), json.dumps(rows, indent=1)))
    assert rows["0" * len(ct.VARIANT_KEYS)] <= 1e-9, rows


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,n,seed,sigma,kink", SOLVER_BATCHES)
def test_real_ceres_vs_hip(H, W, n, seed, sigma, kink):
    from point_trajectory.optimize.build import particlesfm as ours
    real = _real_module()
    uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, seed, sigma, kink)
    out_r = np.asarray(real.optimize_location(uv, ref1, ref2, scale, flow12.astype(np.float64), n, W, H))
    out_g = ours.optimize_location(uv, ref1, ref2, scale, flow12, n, W, H)
    assert float(np.abs(out_r - out_g).max()) <= TOL


def test_reference_track_optimize_with_the_real_module():
    """The reference's own Python (track_optimize.py:24-53, trajectory.py:161-194) with the REAL pybind module in place
    of the stand-in of oracle/ref_shim.py, against the C oracle: ids / lengths equal, positions <= 1e-4 px."""
    ref_root = os.environ.get("PSFM_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref_root, "point_trajectory")):
        pytest.skip("reference tree not present")
    from oracle import oracle as orc
    from oracle import ref_shim
    ref = ref_shim.load_reference(ref_root, particlesfm_module=_real_module())
    d = psfm_synth.synth_sequence(8, 90, 140, seed=52, sigma=0.3, n_occluders=3, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 3)
    full = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 3)
    assert len(full) == O.n_traj
    for i, t in enumerate(full):
        assert t.length() == O.length[i] and t.times[0] == O.birth[i]
        assert float(np.abs(np.array(t.xys) - O.xy[O.off[i]:O.off[i + 1]]).max()) <= TOL
