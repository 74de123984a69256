"""Whole-sequence parity at BASELINE.json's full sizes, against the CPU oracle on the same tensors: configs[0]'s shape in
track mode (last test), and the path-consistency path (flow_check x2 + track_optimize + id order) on

    configs[2] shape   436 x 1024 x   50 frames, sample_ratio 2            (Sintel alley_1 stand-in)
    configs[3] shape  1080 x 1920 x  401 frames, sample_ratio 2            (the 8-GPU config, here on ONE GPU)
    configs[4] shape   480 x  640 x 1000 frames, sample_ratio 1, thres 3.0 (ScanNet stand-in, dense grid)

Bar (north_star): ids / lengths bit-exact, |dxy| <= 1e-4 px on every point, and -- stronger -- every solve ends after the
same number of trust-region iterations with the same termination as the oracle's Ceres-compatible loop.  The flows are
synthesised on the device (seeded), the occlusion maps come from the device (spot-checked against the oracle's here,
bit-exactly pinned in test_gpu_parity.py); the oracle walks the whole sequence on one host core (~2e6 points/s), which
is what makes these the slow tests of the suite.
"""
import os

import numpy as np
import pytest

import psfm_synth

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
TOL = 1e-4

EASY = dict(sigma=0.05, n_occluders=2)
CASES = [
    pytest.param(436, 1024, 50, 2, 1.0, 3, EASY, id="configs2-436x1024x50-r2"),
    pytest.param(1080, 1920, 401, 2, 1.0, 1, EASY, id="configs3-1080x1920x401-r2"),
    pytest.param(480, 640, 1000, 1, 3.0, 4, EASY, id="configs4-480x640x1000-r1-thres3"),
    # SURVEY 8(d)'s second distribution (sigma 0.3, 5 % occluder area): mean track life ~8 frames, every solve takes 20-40
    # trust-region iterations of which a third are rejected and half of the accepted steps are interpolated dogleg steps --
    # nothing of it goes as the fused solve speculates, the launch chain walks the whole sequence
    pytest.param(436, 1024, 50, 2, 1.0, 13, psfm_synth.HARD, id="configs2-hard-sigma0.3-occluders5pct"),
    pytest.param(1080, 1920, 401, 2, 1.0, 11, psfm_synth.HARD, id="configs3-hard-sigma0.3-occluders5pct"),
    # large motion at sequence scale: ~10 px of drift per frame (stride-2 flows on both sides of the 20 px gate of
    # trajectory.py:179, tracks crossing and leaving the image)
    pytest.param(436, 1024, 50, 2, 1.0, 14, dict(sigma=0.05, n_occluders=2, amp=2.0, drift=(9.7, -1.2), warp_b=True),
                 id="configs2-largemotion-drift10px"),
    # the third distribution (psfm_synth.REALISTIC: depth-ordered layers with true (dis)occlusion, spatially correlated flow error,
    # outlier blobs -- what RAFT on real video looks like to this path): ~10 % / ~20 % of the pixels fail the stride-1 / stride-2
    # check, every solve rejects steps at the motion boundaries and is walked by the resident solve / the launch chain
    pytest.param(436, 1024, 50, 2, 1.0, 15, dict(psfm_synth.REALISTIC, realistic=True), id="configs2-realistic-layers-disocclusion"),
]


def _host_bytes_needed(H, W, T, r):
    G = ((H + r - 1) // r) * ((W + r - 1) // r)
    return 2 * T * H * W * 8 + 2 * T * H * W + 5 * 16 * G * T       # two flow stacks, two mask stacks, points (oracle + copies)


@pytest.mark.parametrize("H,W,T,r,thres,seed,dist", CASES)
def test_whole_sequence_track_optimize_vs_oracle(H, W, T, r, thres, seed, dist):
    import psutil
    import torch
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.trajectory import run_connect
    from point_trajectory.utils import flow_check_device
    need = _host_bytes_needed(H, W, T, r)
    if psutil.virtual_memory().available < 1.3 * need:
        pytest.skip("host memory: %.1f GB needed for the oracle's copy of the sequence" % (need / 1e9))
    ctx = _hip.context()
    ctx.set_solver(0, 0)
    orc.set_num_threads(min(16, os.cpu_count() or 1))      # (measured on a 256-core box: all cores are slower than 16)
    if dist.get("realistic"):
        d = psfm_synth.synth_realistic_torch(T, H, W, seed=seed, stride2=True, device="cuda", **{k: v for k, v in dist.items() if k != "realistic"})
    else:
        d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, stride2=True, device="cuda", **dist)
    R = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], thres, r)
    cnt = ctx.solver_counters()
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], thres)
    _, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], thres)
    # the device's maps against the oracle's on a few pairs (first, middle, last of both strides)
    for stack_f, stack_b, maps in ((d["flows_f"], d["flows_b"], occ), (d["flows_f2"], d["flows_b2"], occ2)):
        for k in (0, len(maps) // 2, len(maps) - 1):
            _, o = orc.flow_check([stack_f[k].cpu().numpy()], [stack_b[k].cpu().numpy()], thres)
            assert np.array_equal(o[0], maps[k].cpu().numpy().astype(bool))
    ff, f2 = d["flows_f"].cpu().numpy(), d["flows_f2"].cpu().numpy()
    oo, o2 = occ.cpu().numpy(), occ2.cpu().numpy()
    del d, occ, occ2
    torch.cuda.empty_cache()
    O = orc.track_optimize(list(ff), list(f2), list(oo), list(o2), r)
    del ff, f2, oo, o2
    assert len(R) == O.n_traj and R.n_points == O.n_points
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    err = float(np.abs(R.xy - O.xy).max())
    assert err <= TOL, err
    assert len(R.solve_stats) == len(O.solves) == T - 2
    assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
    assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves]
    assert [s["successful_steps"] for s in R.solve_stats] == [s["successful_steps"] for s in O.solves]
    # what ran: the fused solve on (nearly) every frame of the well-behaved sequences; on the hard distribution the solves that
    # reject steps / leave the Gauss-Newton path must really have been walked by the non-speculated path
    assert cnt["fused"] + cnt["fused_redone"] + cnt["chain"] == T - 2
    rejected = sum(s["iterations"] - s["successful_steps"] for s in O.solves)
    if dist.get("realistic"):
        assert rejected > T and cnt["chain"] + cnt["fused_redone"] >= (T - 2) // 2, cnt
    if dist is psfm_synth.HARD:
        assert rejected > T and sum(s["dogleg_nonGN"] for s in O.solves) > T
        assert cnt["chain"] + cnt["fused_redone"] >= (T - 2) // 2, cnt
    out = os.environ.get("PSFM_WHOLE_SEQ_REPORT")
    if out:
        import json
        with open(out, "a") as fh:
            fh.write(json.dumps({"shape": [H, W, T, r], "thres": thres, "trajectories": int(O.n_traj), "points": int(O.n_points),
                                 "ids_lengths_equal": True, "max_abs_dxy_px": err, "solves": T - 2,
                                 "trust_region_iterations": int(sum(s["iterations"] for s in O.solves)),
                                 "rejected_steps": int(rejected), "dogleg_nonGN": int(sum(s["dogleg_nonGN"] for s in O.solves)),
                                 "distribution": {k: (list(v) if isinstance(v, tuple) else v) for k, v in dist.items()},
                                 "iterations_and_terminations_equal": True, "solver_counters": cnt}) + "\n")


def test_whole_sequence_track_configs0_vs_oracle():
    """BASELINE configs[0] shape (DAVIS 'snowboard' stand-in: 480 x 854 x 50 frames, sample_ratio 4, `track` only): the whole
    sequence through flow_check + track on the device -- in every way of running the recurrence (persistent loop, one launch
    per frame, psfm_connect with flow_check fused / on the side stream) -- against the CPU oracle: occlusion maps bit-equal,
    ids / lengths / positions bit-equal."""
    import torch
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.trajectory import run_connect, run_track
    from point_trajectory.utils import flow_check_device
    T, H, W, r = 50, 480, 854, 4
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=2, sigma=0.05, n_occluders=2, stride2=False, device="cuda")
    ff, fb = d["flows_f"].cpu().numpy(), d["flows_b"].cpu().numpy()
    _, occ_o = orc.flow_check(list(ff), list(fb), 1.0)
    O = orc.track(list(ff), occ_o, r)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    assert np.array_equal(occ.cpu().numpy().astype(bool), np.stack(occ_o))
    ctx = _hip.context()
    try:
        for mode in (1, 2):
            ctx.set_chain_mode(mode)
            for R in (run_track(d["flows_f"], occ, None, None, r), run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r)):
                assert R.info["chain_mode"] == mode
                assert len(R) == O.n_traj and R.n_points == O.n_points
                assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
    finally:
        ctx.set_chain_mode(0)
