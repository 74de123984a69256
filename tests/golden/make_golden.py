#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every vector below is an output of the unmodified reference files
point_trajectory/{utils,trajectory,track,track_optimize}.py (torch-CPU
grid_sample, SciPy EDT, NumPy), executed through oracle/ref_shim.py.  The one
piece that is NOT the reference is `optimize_location` (pybind11 + Ceres, cannot
be built here): the shim forwards it to the C restatement, so the solver
iterate inside the track_optimize fixtures is "parity unpinned"; everything
around it (buffer/index/scale semantics, ids, lengths) is pinned.

Inputs are regenerated from seeds by psfm_synth; each fixture stores a sha256 of
its inputs so a drifting generator is detected instead of silently mis-compared.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))

import psfm_synth  # noqa: E402
from oracle import ref_shim  # noqa: E402

TRACK_CASES = [
    # name, T(frames), H, W, ratio, seed, sigma, occluders
    ("track_48x64_r2", 8, 48, 64, 2, 3, 0.3, 2),
    ("track_45x70_r1", 7, 45, 70, 1, 4, 0.2, 1),
    ("track_50x66_r3", 8, 50, 66, 3, 5, 0.35, 2),
    ("track_52x61_r4", 6, 52, 61, 4, 6, 0.1, 1),
]
OPT_CASES = [
    ("opt_48x64_r2", 8, 48, 64, 2, 7, 0.05, 1),
    ("opt_45x70_r3", 7, 45, 70, 3, 8, 0.3, 2),
]


def input_hash(d):
    h = hashlib.sha256()
    for k in sorted(d):
        for a in d[k]:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_nonfinite(ref):
    """NaN / Inf / huge flow components: the reference's torch ops define what happens (NaN errors compare false, a
    track whose next position is not finite fails the strict bounds test and ends) and nothing may fault."""
    T, H, W, r, seed = 7, 40, 56, 2, 41
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.2, n_occluders=1, stride2=False),
                                    seed=seed + 1)
    err, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
    tr = ref.track(d["flows_f"], occ, r)
    b, l, off, xy = ref_shim.trajs_to_csr(tr)
    np.savez_compressed(os.path.join(HERE, "nonfinite_40x56_r2.npz"), T=T, H=H, W=W, ratio=r, seed=seed, sigma=0.2,
                        n_occluders=1, input_hash=input_hash(d), fc_err=np.stack(err), fc_occ=np.packbits(np.stack(occ)),
                        birth=b, length=l, xy=xy)
    print("nonfinite", len(tr), "tracks,", int(l.sum()), "points,", int(np.isnan(np.stack(err)).sum()), "NaN errors,",
          int((~np.isfinite(np.concatenate([np.ravel(a) for a in d["flows_f"]]))).sum()), "non-finite forward components")


LARGE_MOTION_CASES = [
    # name, T, H, W, ratio, seed, amp, sigma, occluders, drift
    ("opt_largemotion_96x128_r2", 9, 96, 128, 2, 61, 1.5, 0.05, 3, (9.6, 1.5)),
    ("opt_largemotion_90x140_r3", 8, 90, 140, 3, 62, 2.0, 0.25, 2, (-8.8, 4.0)),
]


TRACK_LARGE_MOTION = [
    # name, T, H, W, ratio, seed, amp, sigma, occluders, drift: track() with tracks crossing the image in ~10 frames and leaving it
    ("track_largemotion_80x120_r2", 9, 80, 120, 2, 63, 1.5, 0.1, 2, (10.4, -2.2)),
    ("track_largemotion_75x110_r1", 7, 75, 110, 1, 64, 2.5, 0.2, 1, (-7.6, 6.1)),
]


def make_track_large_motion(ref):
    for name, T, H, W, r, seed, amp, sigma, nocc, drift in TRACK_LARGE_MOTION:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, amp=amp, sigma=sigma, n_occluders=nocc, stride2=False, drift=drift, warp_b=True)
        _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
        tr = ref.track(d["flows_f"], occ, r)
        b, l, off, xy = ref_shim.trajs_to_csr(tr)
        left = int((l[b + l - 1 < T - 1]).shape[0])          # trajectories that ended before the last frame
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, amp=amp, sigma=sigma,
                            n_occluders=nocc, drift=np.asarray(drift, np.float64), warp_b=1, input_hash=input_hash(d),
                            birth=b, length=l, xy=xy, occ=np.packbits(np.stack(occ)), ended_early=left)
        print(name, len(tr), "tracks,", int(l.sum()), "points,", left, "ended before the last frame, occluded fraction",
              round(float(np.stack(occ).mean()), 3))


def make_large_motion(ref):
    """`loss02_scale = (1 - occ02) * (|flow02| < 20)` (trajectory.py:179) on BOTH sides of the gate, with
    continuous occ02 weights at occluder borders, and tracks whose solver parameters / chain taps leave the image (a drift of
    ~10 px per frame).  The reference's own optimize_buffer is watched from outside (its `optimize_location` argument
    `ref2_scale` and the samples its `grid_sample` returns) so that the fixture records what it actually exercised."""
    seen = {"scale_zero": 0, "scale_one": 0, "scale_frac": 0, "norm_ge20": 0, "norm_lt20_s2": 0, "solves": 0}
    real_opt = ref.particlesfm.optimize_location
    real_gs = ref.trajectory.grid_sample
    state = {"in_opt": 0, "calls": []}

    def gs(tensor, xy):
        out = real_gs(tensor, xy)
        state["calls"].append(np.asarray(out))
        return out

    def opt(uv12, ref1, ref2, scale, *a, **k):
        sc = np.asarray(scale, np.float64).reshape(-1)
        seen["solves"] += 1
        seen["scale_zero"] += int((sc == 0).sum())
        seen["scale_one"] += int((sc == 1).sum())
        seen["scale_frac"] += int(((sc > 0) & (sc < 1)).sum())
        # optimize_buffer samples flow01, flow02, occ02 in this order right before the call (trajectory.py:172-178)
        flow02 = state["calls"][-2]
        nrm = np.sqrt((flow02.astype(np.float32) ** 2).sum(-1))
        assert flow02.shape == (len(sc), 2)
        seen["norm_ge20"] += int((nrm >= 20).sum())
        seen["norm_lt20_s2"] += int((nrm < 20).sum())
        state["calls"].clear()
        return real_opt(uv12, ref1, ref2, scale, *a, **k)

    ref.trajectory.grid_sample = gs
    ref.particlesfm.optimize_location = opt
    try:
        for name, T, H, W, r, seed, amp, sigma, nocc, drift in LARGE_MOTION_CASES:
            for k in seen:
                seen[k] = 0
            d = psfm_synth.synth_sequence(T, H, W, seed=seed, amp=amp, sigma=sigma, n_occluders=nocc, stride2=True, drift=drift,
                                          warp_b=True)
            _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
            _, occ2 = ref.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
            tr = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            b, l, off, xy = ref_shim.trajs_to_csr(tr)
            outside = int(((xy[:, 0] < 0) | (xy[:, 0] > W - 1) | (xy[:, 1] < 0) | (xy[:, 1] > H - 1)).sum())
            assert seen["norm_ge20"] > 100 and seen["norm_lt20_s2"] > 100 and seen["scale_frac"] > 100, seen
            np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, amp=amp, sigma=sigma,
                                n_occluders=nocc, drift=np.asarray(drift, np.float64), warp_b=1, input_hash=input_hash(d),
                                birth=b, length=l, xy=xy, occ=np.packbits(np.stack(occ)), occ2=np.packbits(np.stack(occ2)),
                                gate_closed=seen["norm_ge20"], gate_open=seen["norm_lt20_s2"], scale_fractional=seen["scale_frac"],
                                scale_zero=seen["scale_zero"], scale_one=seen["scale_one"], points_outside_image=outside)
            print(name, len(tr), "tracks,", int(l.sum()), "points;", dict(seen), "points outside the image:", outside)
    finally:
        ref.trajectory.grid_sample = real_gs
        ref.particlesfm.optimize_location = real_opt


REALISTIC_CASES = [
    # name, T, H, W, ratio, seed, optimize: psfm_synth.REALISTIC (layers in depth order with true (dis)occlusion, correlated flow error,
    # outlier blobs -- what RAFT on real video looks like to flow_check and to the solver)
    ("opt_realistic_96x160_r2", 8, 96, 160, 2, 91, True),
    ("opt_realistic_84x132_r1", 6, 84, 132, 1, 92, True),
    ("track_realistic_100x150_r2", 9, 100, 150, 2, 93, False),
]


def make_realistic(ref):
    """the reference's own flow_check + track / track_optimize on the third distribution.  The fixture records what the
    sequence exercised: occluded fractions of both strides, trajectories that ended early, solves that rejected steps (statistics of
    the C restatement the shim forwards optimize_location to)."""
    from oracle import oracle as orc
    for name, T, H, W, r, seed, opt in REALISTIC_CASES:
        d = psfm_synth.synth_realistic(T, H, W, seed=seed, stride2=opt, **psfm_synth.REALISTIC)
        _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
        extra = {}
        if opt:
            _, occ2 = ref.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
            tr = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            O = orc.track_optimize(d["flows_f"], d["flows_f2"], [np.asarray(o) for o in occ], [np.asarray(o) for o in occ2], r)
            extra = dict(occ2=np.packbits(np.stack(occ2)), occluded2=float(np.stack(occ2).mean()),
                         solve_iterations=np.asarray([s_["iterations"] for s_ in O.solves], np.int32),
                         solve_successful=np.asarray([s_["successful_steps"] for s_ in O.solves], np.int32))
        else:
            tr = ref.track(d["flows_f"], occ, r)
        b, l, off, xy = ref_shim.trajs_to_csr(tr)
        early = int((b + l - 1 < T - 1).sum())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, realistic=1, input_hash=input_hash(d),
                            birth=b, length=l, xy=xy, occ=np.packbits(np.stack(occ)), occluded=float(np.stack(occ).mean()),
                            ended_early=early, **extra)
        print(name, len(tr), "tracks,", int(l.sum()), "points,", early, "ended early, occluded", round(float(np.stack(occ).mean()), 3),
              ("occluded (stride 2) %.3f, iterations %s, accepted %s" % (extra["occluded2"], list(extra["solve_iterations"]),
                                                                           list(extra["solve_successful"]))) if opt else "")


def main():
    import torch
    ref = ref_shim.load()
    if sys.argv[1:] == ["realistic"]:
        return make_realistic(ref)
    if sys.argv[1:] == ["nonfinite"]:
        return make_nonfinite(ref)
    if sys.argv[1:] == ["largemotion"]:
        make_track_large_motion(ref)
        return make_large_motion(ref)
    out = {}

    # ---- 1. sampler known answers (trajectory.py:25-37) -------------------
    rng = np.random.default_rng(11)
    H, W = 37, 53
    m2 = rng.standard_normal((H, W, 2)).astype(np.float32)
    m1 = (rng.uniform(size=(H, W)) < 0.3)
    pts = [rng.uniform([-3, -3], [W + 2, H + 2], size=(4000, 2))]
    pts.append(np.stack(np.meshgrid(np.arange(-1, W + 1), np.arange(-1, H + 1)), -1).reshape(-1, 2).astype(np.float64))
    pts.append(np.array([[0, 0], [W - 1, H - 1], [W - 1, 0], [0, H - 1], [W - 1 - 1e-9, 3.5], [1e-12, 1e-12],
                         [-1e-12, 5.0], [W - 1 + 1e-7, H - 1 + 1e-7], [25.5, 17.5], [1e6, 1e6], [-1e6, 2.0]], np.float64))
    pts = np.concatenate(pts, 0)
    s2 = ref.grid_sample(torch.from_numpy(m2).permute(2, 0, 1).float(), pts.copy())
    s1 = ref.grid_sample(torch.from_numpy(m1).unsqueeze(0).float(), pts.copy())
    # a 1080p-sized map exercises the fp32 normalise/un-normalise round trip at x~1900
    Hb, Wb = 1080, 1920
    mb = rng.standard_normal((Hb, Wb, 2)).astype(np.float32)
    pb = rng.uniform([-2, -2], [Wb + 1, Hb + 1], size=(6000, 2))
    sb = ref.grid_sample(torch.from_numpy(mb).permute(2, 0, 1).float(), pb.copy())
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), seed=11, H=H, W=W, pts=pts, s2=s2, s1=s1,
                        Hb=Hb, Wb=Wb, pb=pb, sb=sb,
                        map_hash=hashlib.sha256(m2.tobytes() + m1.tobytes() + mb.tobytes()).hexdigest())

    # ---- 2. flow_check (utils.py:94-105) -----------------------------------
    d = psfm_synth.synth_sequence(4, 64, 96, seed=21, sigma=0.4, n_occluders=2, stride2=False)
    for thres in (1.0, 3.0):
        err, occ = ref.flow_check(d["flows_f"], d["flows_b"], thres)
        out["fc_err_%g" % thres] = np.stack(err)
        out["fc_occ_%g" % thres] = np.packbits(np.stack(occ))
    # large flows near the border exercise the out-of-bounds branch
    dd = psfm_synth.synth_sequence(3, 40, 56, seed=22, amp=9.0, sigma=0.0, stride2=False)
    err, occ = ref.flow_check(dd["flows_f"], dd["flows_b"], 1.0)
    np.savez_compressed(os.path.join(HERE, "flow_check.npz"), hash_a=input_hash(d), hash_b=input_hash(dd),
                        big_err=np.stack(err), big_occ=np.packbits(np.stack(occ)), **out)

    # ---- 3. track (track.py:24-50) ------------------------------------------
    for name, T, H, W, r, seed, sigma, nocc in TRACK_CASES:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=False)
        _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
        tr = ref.track(d["flows_f"], occ, r)
        b, l, off, xy = ref_shim.trajs_to_csr(tr)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, sigma=sigma,
                            n_occluders=nocc, input_hash=input_hash(d), birth=b, length=l, xy=xy)
        print(name, len(tr), "tracks,", int(l.sum()), "points, short(<3):", int((l < 3).sum()))

    # degenerate: every track dies at step 1 (all-occluded map) -> SciPy's no-background EDT
    T, H, W, r = 5, 24, 30, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=31, sigma=0.05, stride2=False)
    _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
    occ = [o.copy() for o in occ]
    occ[1][:] = True
    tr = ref.track(d["flows_f"], occ, r)
    b, l, off, xy = ref_shim.trajs_to_csr(tr)
    np.savez_compressed(os.path.join(HERE, "track_alldie_24x30_r2.npz"), T=T, H=H, W=W, ratio=r, seed=31,
                        input_hash=input_hash(d), birth=b, length=l, xy=xy)
    print("alldie", len(tr))

    # ---- 4. track_optimize (track_optimize.py:24-53); solver iterate unpinned ------------
    for name, T, H, W, r, seed, sigma, nocc in OPT_CASES:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
        _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = ref.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        tr = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        b, l, off, xy = ref_shim.trajs_to_csr(tr)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, sigma=sigma,
                            n_occluders=nocc, input_hash=input_hash(d), birth=b, length=l, xy=xy)
        print(name, len(tr), "tracks,", int(l.sum()), "points")

    # ---- 5. EDT rule (trajectory.py:150) == integer disc, asserted at generation time ----
    import scipy.ndimage
    for r in (1, 2, 3, 4, 5):
        occm = (rng.uniform(size=(40, 50, 1)) < 0.03).astype(np.float64)
        edt = scipy.ndimage.distance_transform_edt(1.0 - occm)
        ref_mask = (edt > r)[::r, ::r, 0]
        ys, xs = np.nonzero(occm[:, :, 0])
        gy, gx = np.meshgrid(np.arange(0, 40, r), np.arange(0, 50, r), indexing="ij")
        d2 = (gy[..., None] - ys) ** 2 + (gx[..., None] - xs) ** 2
        assert ((d2.min(-1) > r * r) == ref_mask).all(), r
    print("EDT == integer-disc rule verified for r=1..5")

    make_nonfinite(ref)
    make_track_large_motion(ref)
    make_large_motion(ref)
    make_realistic(ref)


if __name__ == "__main__":
    main()
