"""GPU stress: random sequences through psfm_connect / psfm_track in persistent (fused) mode vs per-frame launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_connect, run_track, _result_to_host

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _hip.context()
bad = 0
t0 = time.time()
for i in range(n_cases):
    big = len(sys.argv) > 3
    H = int(rng.integers(300, 1081)) if big else int(rng.integers(20, 400))
    W = int(rng.integers(400, 1921)) if big else int(rng.integers(20, 500))
    T = int(rng.integers(2, 25 if big else 40)); r = int(rng.choice([2, 2, 3, 4] if big else [1, 1, 2, 2, 3, 4]))
    sigma = float(rng.choice([0.05, 0.2, 0.4, 0.8])); nocc = int(rng.integers(0, 4)); seed = int(rng.integers(0, 1 << 30))
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=False)
    out = {}
    for mode in (1, 2):
        ctx.set_chain_mode(mode)
        info = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True)
        out[("c", mode)] = (_result_to_host(ctx, info), info.chain_mode)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    ctx.set_chain_mode(2)
    info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    out[("t", 2)] = (_result_to_host(ctx, info), info.chain_mode)
    for key in (("c", 2), ("t", 2)):
        if out[key][1] != 2:
            print("fell back: case %d %s T=%d %dx%d r=%d sigma=%.2f nocc=%d seed=%d  n_traj %d lanes_peak %s cap %s" % (
                i, key, T, H, W, r, sigma, nocc, seed, len(out[key][0]), out[key][0].info.get("n_lanes_peak"), out[key][0].info.get("lane_capacity")))
    A = out[("c", 1)][0]
    for key in (("c", 2), ("t", 2)):
        B = out[key][0]
        same = len(A) == len(B) and np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length) and np.array_equal(A.xy, B.xy)
        if not same:
            bad += 1
            print("MISMATCH case %d %s: T=%d %dx%d r=%d sigma=%.2f nocc=%d seed=%d  n_traj %d vs %d (modes %d/%d)" % (
                i, key, T, H, W, r, sigma, nocc, seed, len(A), len(B), out[("c", 1)][1], out[key][1]))
ctx.set_chain_mode(0)
print("%d cases, %d mismatches, %.1f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
