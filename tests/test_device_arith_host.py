"""The DEVICE's arithmetic against the reference's golden vectors, without a GPU: particle-sfm_amd/csrc/psfm_device.h and
psfm_chain.h -- the fp32 sampler (trajectory.py:25-37), the flow_check verdict of a pixel in its three forms (utils.py:58-105),
a chain step (trajectory.py:45-62) and the EDT respawn rule folded into grid-resolution marks (trajectory.py:129-152) -- are
compiled for the host through a stand-in for <hip/hip_runtime.h> (tests/host/shim: every *_rn intrinsic is the IEEE operation it
names) and driven by tests/host/device_arith_host.cpp, which keeps plain lists where the kernels have lanes, ballots and
atomics.  The vectors are the ones the reference's own Python produced (tests/golden/make_golden.py): bit-exact, like the GPU
tests of the same kernels."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import psfm_synth
from _common import golden, regen_inputs, assert_csr_equal, input_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dev_arith") / "libdev_arith_host.so")
    cmd = ["g++", "-O2", "-mfma", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "tests", "host", "shim"),
           "-I", os.path.join(ROOT, "particle-sfm_amd", "csrc"), os.path.join(ROOT, "tests", "host", "device_arith_host.cpp"),
           os.path.join(ROOT, "tests", "host", "pc_chain_host.cpp"), "-o", out]
    subprocess.run(cmd, check=True)
    L = ctypes.CDLL(out)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    L.psfm_host_grid_sample.argtypes = [vp, i32, i32, vp, i64, vp]
    L.psfm_host_grid_sample1.argtypes = [vp, i32, i32, vp, i64, vp]
    L.psfm_host_flow_check.argtypes = [vp, vp, i32, i32, ctypes.c_float, i32, vp, vp]
    L.psfm_host_track.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i64, i64, ctypes.POINTER(i64), vp, vp, vp, vp]
    L.psfm_host_track.restype = i64
    return L


def _sample(L, m, pts):
    m = np.ascontiguousarray(m, np.float32)
    pts = np.ascontiguousarray(pts, np.float32)
    if m.ndim == 3:
        out = np.empty((len(pts), 2), np.float32)
        L.psfm_host_grid_sample(m.ctypes.data, m.shape[0], m.shape[1], pts.ctypes.data, len(pts), out.ctypes.data)
    else:
        out = np.empty(len(pts), np.float32)
        L.psfm_host_grid_sample1(m.ctypes.data, m.shape[0], m.shape[1], pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def _flow_check(L, f, b, thres, form):
    f, b = np.ascontiguousarray(f, np.float32), np.ascontiguousarray(b, np.float32)
    H, W = f.shape[:2]
    occ, err = np.empty((H, W), np.uint8), np.empty((H, W), np.float32)
    L.psfm_host_flow_check(f.ctypes.data, b.ctypes.data, H, W, thres, form, occ.ctypes.data, err.ctypes.data)
    return err, occ.astype(bool)


def _track(L, flows, occ, ratio, flows2=None, occ2=None):
    fl = [np.ascontiguousarray(f, np.float32) for f in flows]
    oc = [np.ascontiguousarray(o, np.uint8) for o in occ]
    n = len(fl)
    H, W = fl[0].shape[:2]
    G = ((H + ratio - 1) // ratio) * ((W + ratio - 1) // ratio)
    cap_t, cap_p = G * (n + 1), G * (n + 1) * 2
    birth, length, xy = np.empty(cap_t, np.int32), np.empty(cap_t, np.int32), np.empty((cap_p, 2), np.float64)
    npts = ctypes.c_long(0)
    ptrs = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    f2 = o2 = its = terms = None
    if flows2 is not None:
        f2l = [np.ascontiguousarray(f, np.float32) for f in flows2]
        o2l = [np.ascontiguousarray(o, np.uint8) for o in occ2]
        f2, o2 = ptrs(f2l), ptrs(o2l)
        its, terms = np.full(n - 1, -2, np.int32), np.full(n - 1, -2, np.int32)
    nt = L.psfm_host_track(ptrs(fl), ptrs(oc), n, H, W, ratio, birth.ctypes.data, length.ctypes.data, xy.ctypes.data, cap_t, cap_p,
                           ctypes.byref(npts), f2, o2, its.ctypes.data if its is not None else None,
                           terms.ctypes.data if terms is not None else None)
    assert nt >= 0, nt
    out = (birth[:nt].copy(), length[:nt].copy(), xy[:npts.value].copy())
    return out + (its, terms) if flows2 is not None else out


def test_device_sampler_bit_exact(dev):
    g = golden("sampler")
    rng = np.random.default_rng(int(g["seed"]))
    H, W = int(g["H"]), int(g["W"])
    m2 = rng.standard_normal((H, W, 2)).astype(np.float32)
    m1 = (rng.uniform(size=(H, W)) < 0.3)
    assert np.array_equal(_sample(dev, m2, g["pts"]).view(np.uint32), g["s2"].view(np.uint32))
    assert np.array_equal(_sample(dev, m1.astype(np.float32), g["pts"]).view(np.uint32), g["s1"].reshape(-1).view(np.uint32))
    rng.uniform([-3, -3], [W + 2, H + 2], size=(4000, 2))        # (the generator's point draws between the maps)
    mb = rng.standard_normal((int(g["Hb"]), int(g["Wb"]), 2)).astype(np.float32)
    assert hashlib.sha256(m2.tobytes() + m1.tobytes() + mb.tobytes()).hexdigest() == str(g["map_hash"])
    assert np.array_equal(_sample(dev, mb, g["pb"]).view(np.uint32), g["sb"].view(np.uint32))       # 1080p coordinates


def test_device_flow_check_bit_exact_in_its_three_forms(dev):
    g = golden("flow_check")
    d = psfm_synth.synth_sequence(4, 64, 96, seed=21, sigma=0.4, n_occluders=2, stride2=False)
    for thres in (1.0, 3.0):
        res = [[_flow_check(dev, f, b, thres, form) for f, b in zip(d["flows_f"], d["flows_b"])] for form in (0, 1, 2)]
        assert np.array_equal(np.stack([e for e, _ in res[0]]).view(np.uint32), g["fc_err_%g" % thres].view(np.uint32))
        for form in (0, 1, 2):       # the mask-only forms never take the root or the true division: same masks
            assert np.array_equal(np.packbits(np.stack([o for _, o in res[form]])), g["fc_occ_%g" % thres]), form
    dd = psfm_synth.synth_sequence(3, 40, 56, seed=22, amp=9.0, sigma=0.0, stride2=False)
    for form in (0, 1, 2):
        res = [_flow_check(dev, f, b, 1.0, form) for f, b in zip(dd["flows_f"], dd["flows_b"])]
        if form == 0:
            assert np.array_equal(np.stack([e for e, _ in res]).view(np.uint32), g["big_err"].view(np.uint32))
        assert np.array_equal(np.packbits(np.stack([o for _, o in res])), g["big_occ"])


@pytest.mark.parametrize("name", ["track_48x64_r2", "track_45x70_r1", "track_50x66_r3", "track_52x61_r4",
                                  "track_largemotion_80x120_r2", "track_largemotion_75x110_r1", "track_realistic_100x150_r2"])
def test_device_chain_arithmetic_reproduces_the_reference_track(dev, name):
    """Births on the stride-r grid, the step, deaths, the respawn rule as grid-resolution marks, ids by the key: every trajectory
    of the reference's own track() -- ids, lengths, f64 positions -- bit for bit."""
    g = golden(name)
    d = regen_inputs(g, stride2=False)
    occ = [_flow_check(dev, f, b, 1.0, 1)[1] for f, b in zip(d["flows_f"], d["flows_b"])]
    birth, length, xy = _track(dev, d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(birth, length, xy, g)


def test_device_chain_arithmetic_all_tracks_die(dev):
    g = golden("track_alldie_24x30_r2")
    d = regen_inputs(g, stride2=False)
    occ = [_flow_check(dev, f, b, 1.0, 2)[1] for f, b in zip(d["flows_f"], d["flows_b"])]
    occ[1][:] = True
    birth, length, xy = _track(dev, d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(birth, length, xy, g)


def test_device_arithmetic_on_nonfinite_flows(dev):
    g = golden("nonfinite_40x56_r2")
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(int(g["T"]), int(g["H"]), int(g["W"]), seed=int(g["seed"]),
                                                              sigma=float(g["sigma"]), n_occluders=int(g["n_occluders"]),
                                                              stride2=False), seed=int(g["seed"]) + 1)
    assert input_hash(d) == str(g["input_hash"])
    res = [_flow_check(dev, f, b, 1.0, 0) for f, b in zip(d["flows_f"], d["flows_b"])]
    assert np.array_equal(np.stack([e for e, _ in res]).view(np.uint32), g["fc_err"].view(np.uint32))
    occ = [o for _, o in res]
    assert np.array_equal(np.packbits(np.stack(occ)), g["fc_occ"])
    for form in (1, 2):
        assert np.array_equal(np.packbits(np.stack([_flow_check(dev, f, b, 1.0, form)[1] for f, b in zip(d["flows_f"], d["flows_b"])])), g["fc_occ"])
    birth, length, xy = _track(dev, d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(birth, length, xy, g)


@pytest.mark.parametrize("name", ["opt_48x64_r2", "opt_45x70_r3", "opt_largemotion_96x128_r2", "opt_largemotion_90x140_r3",
                                  "opt_realistic_96x160_r2"])
def test_device_arithmetic_reproduces_the_reference_track_optimize(dev, name):
    """track_optimize.py:24-53 with the device's arithmetic end to end: chain step, references / weight of optimize_buffer from
    the fp32 sampler (both sides of the 20 px gate of trajectory.py:179 in the large-motion fixtures), every frame's solve by the
    device's launch chain -- against the fixtures the reference's own Python produced (with the C restatement in the solver's
    seat): ids and lengths exact, positions to 1e-5 px, and every solve's iteration count and termination equal to the oracle's."""
    from oracle import oracle as orc
    g = golden(name)
    d = regen_inputs(g, stride2=True)
    occ = [_flow_check(dev, f, b, 1.0, 1)[1] for f, b in zip(d["flows_f"], d["flows_b"])]
    occ2 = [_flow_check(dev, f, b, 1.0, 1)[1] for f, b in zip(d["flows_f2"], d["flows_b2"])]
    birth, length, xy, its, terms = _track(dev, d["flows_f"], occ, int(g["ratio"]), d["flows_f2"], occ2)
    assert_csr_equal(birth, length, xy, g, tol=1e-5)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, int(g["ratio"]))
    assert [int(v) for v in its] == [s["iterations"] for s in O.solves]
    assert [int(v) for v in terms] == [s["termination"] for s in O.solves]


@pytest.mark.parametrize("T,H,W,r,seed,sigma,nocc,drift", [(12, 120, 214, 2, 31, 0.3, 3, (0.0, 0.0)), (10, 97, 131, 1, 32, 0.15, 2, (6.0, -4.0)),
                                                          (14, 160, 200, 4, 33, 0.5, 4, (0.0, 0.0)), (9, 75, 203, 3, 34, 0.05, 1, (-9.0, 2.0))])
def test_device_chain_arithmetic_equals_the_oracle_on_larger_sequences(dev, T, H, W, r, seed, sigma, nocc, drift):
    """Beyond the committed fixtures: noisy / drifting sequences with thousands of deaths and respawns, against the oracle (itself
    pinned bit-exact to the reference's Python): masks, ids, lengths and f64 positions bit for bit."""
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=False, drift=drift)
    err_o, occ_o = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    res = [_flow_check(dev, f, b, 1.0, 0) for f, b in zip(d["flows_f"], d["flows_b"])]
    assert np.array_equal(np.stack([e for e, _ in res]).view(np.uint32), np.stack(err_o).view(np.uint32))
    occ = [o for _, o in res]
    assert np.array_equal(np.stack(occ), np.stack(occ_o))
    O = orc.track(d["flows_f"], occ_o, r)
    birth, length, xy = _track(dev, d["flows_f"], occ, r)
    assert len(birth) == O.n_traj > 2000 and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
    assert np.array_equal(xy, O.xy)
    assert int((O.length < T).sum()) > 500          # plenty of tracks that died or were respawned


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**31 - 1), h=st.integers(2, 19), w=st.integers(2, 23), thres=st.sampled_from([0.0, 0.5, 1.0, 3.0, 1e-30, 1e30]))
def test_device_flow_check_forms_agree_with_the_oracle_on_absurd_fields(dev, seed, h, w, thres):
    """Flow components of every magnitude -- subnormal, ordinary, 1e10, 1e30, 3e38, +-Inf, NaN, -0 -- in both fields: the error map of
    the device's code is the oracle's bit for bit (NaN where the oracle has NaN), and the two mask-only forms (reciprocal division,
    threshold under the root, 16-byte interior taps) return the same mask wherever they promise to: the fast division is exact
    for |x| < 1e30 and +0 only, beyond that all four taps are out of bounds for both quotients."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    mags = np.array([0.0, -0.0, 1e-42, 1e-38, 1e-3, 1.0, 7.5, 1e3, 1e10, 1e30, 3e38, np.inf, -np.inf, np.nan], np.float32)

    def field():
        base = rng.normal(0, 2.0, (h, w, 2)).astype(np.float32)
        pick = rng.random((h, w, 2)) < 0.25
        sel = mags[rng.integers(0, len(mags), (h, w, 2))] * np.where(rng.random((h, w, 2)) < 0.5, 1.0, -1.0).astype(np.float32)
        return np.where(pick, sel, base).astype(np.float32)

    f, b = field(), field()
    with np.errstate(all="ignore"):
        err_o, occ_o = orc.flow_check([f], [b], thres)
    e0, o0 = _flow_check(dev, f, b, thres, 0)
    nan = np.isnan(err_o[0])              # (which NaN a host build produces -- sign, payload -- says nothing about the GPU's)
    assert np.array_equal(np.isnan(e0), nan) and np.array_equal(e0[~nan].view(np.uint32), err_o[0][~nan].view(np.uint32))
    assert np.array_equal(o0, occ_o[0])
    for form in (1, 2):
        assert np.array_equal(_flow_check(dev, f, b, thres, form)[1], occ_o[0]), form


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**20), r=st.integers(1, 4), per_field=st.integers(1, 12), sigma=st.sampled_from([0.05, 0.3]))
def test_device_chain_arithmetic_on_random_poisoned_sequences(dev, seed, r, per_field, sigma):
    """NaN / +-Inf / 1e30 components sprinkled over every flow field: masks, ids, lengths and positions of the device's arithmetic
    equal the oracle's (a track stepping to a non-finite position ends there; NaN errors compare false)."""
    from oracle import oracle as orc
    T, H, W = 6, 34, 45
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=1, stride2=False),
                                    seed=seed + 1, per_field=per_field)
    with np.errstate(all="ignore"):
        _, occ_o = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        O = orc.track(d["flows_f"], occ_o, r)
    occ = [_flow_check(dev, f, b, 1.0, 2)[1] for f, b in zip(d["flows_f"], d["flows_b"])]
    assert np.array_equal(np.stack(occ), np.stack(occ_o))
    birth, length, xy = _track(dev, d["flows_f"], occ, r)
    assert len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length) and np.array_equal(xy, O.xy)


@pytest.mark.parametrize("seed,per_field", [(77, 6), (5, 3), (19, 10)])
def test_device_arithmetic_track_optimize_carries_on_after_failed_solves(dev, seed, per_field):
    """Non-finite flow components in all four stacks: a frame whose solve fails (FAILURE in IterationZero) keeps its chained
    positions and the sequence goes on -- ids, lengths, per-solve iterations and terminations as in the oracle, positions to 1e-6 px
    where finite and non-finite in the same places."""
    from oracle import oracle as orc
    T, H, W, r = 9, 60, 84, 2
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.1, n_occluders=1, stride2=True),
                                    seed=seed + 1, per_field=per_field)
    with np.errstate(all="ignore"):
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    birth, length, xy, its, terms = _track(dev, d["flows_f"], occ, r, d["flows_f2"], occ2)
    assert [int(v) for v in terms] == [s["termination"] for s in O.solves]
    assert [int(v) for v in its] == [s["iterations"] for s in O.solves]
    assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
    both = np.isfinite(O.xy)
    assert np.array_equal(np.isfinite(xy), both) and float(np.abs(xy[both] - O.xy[both]).max()) <= 1e-6
