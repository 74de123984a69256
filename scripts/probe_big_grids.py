#!/usr/bin/env python3
"""The per-frame chain step on grids the persistent loop cannot host (more than 524 288 grid points): launch time and
achieved bandwidth on K2's algorithmic bytes (min(8P,32A) + min(P,4A) + 33A)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_track, _result_to_host
from point_trajectory.utils import flow_check_device
ctx = _hip.context()
for (H, W, r, T) in [(1080, 1920, 1, 25), (2160, 3840, 2, 13), (720, 1280, 1, 41), (1080, 1920, 2, 101)]:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=1, sigma=0.05, n_occluders=2, stride2=False)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    ctx.set_chain_mode(1)
    for _ in range(2):
        info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    ctx.set_profiling(1)
    info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    pr = ctx.profile(); ctx.set_profiling(0)
    R = _result_to_host(ctx, info)
    nf = T - 1
    last = R.birth.astype(np.int64) + R.length - 1
    A = float(R.n_points - int((last == nf).sum())) / nf
    P = float(H * W)
    bytes_ = min(8 * P, 32 * A) + min(P, 4 * A) + 33 * A
    us = 1e3 * pr["chain_step"]["total_ms"] / pr["chain_step"]["launches"]
    print(json.dumps({"shape": [H, W, r, T], "grid_points": ((H + r - 1) // r) * ((W + r - 1) // r), "avg_alive": A, "chain_step_us": us,
                      "bytes_per_launch": bytes_, "GBs": bytes_ / (us * 1e-6) / 1e9, "frac_of_8TBs": bytes_ / (us * 1e-6) / 8e12,
                      "chain_mode": int(info.chain_mode)}))
    del d, occ, R
    torch.cuda.empty_cache()
ctx.set_chain_mode(0)
