"""GPU probe of ONE build of the library (PSFM_HIP_LIB selects it): is the persistent frame loop still exact, and how fast?
  * small shapes (deaths, respawns, every sample ratio) and the 1080p x 101 headline sequence: psfm_track and psfm_connect in
    chain mode 2 (persistent loop) vs mode 1 (one launch per frame) -- identical arrays?
  * HIP-event time of the persistent launch on ready maps (K2 on its own bytes) and of the fused launch, end-to-end time of
    psfm_connect; medians over 10 runs.
Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track, run_connect

ctx = _hip.context()
out = {"lib": os.path.basename(_hip.LIB_PATH), "cases": [], "ok": True}


def same(A, B):
    return bool(len(A) == len(B) and np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length)
                and np.array_equal(A.off, B.off) and np.array_equal(A.xy, B.xy))


cases = [(7, 48, 64, 2, 3, 0.3), (9, 45, 70, 1, 5, 0.3), (12, 50, 66, 3, 7, 0.3), (10, 52, 61, 4, 9, 0.3), (21, 200, 300, 2, 11, 0.3),
         (31, 270, 480, 1, 12, 0.3), (3, 40, 56, 2, 13, 0.3), (2, 40, 56, 2, 14, 0.3), (41, 436, 1024, 2, 15, 0.1), (101, 1080, 1920, 2, 0, 0.05)]
for (T, H, W, r, seed, sigma) in cases:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, sigma=sigma, n_occluders=2, stride2=False)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    res = {}
    for mode in (1, 2):
        ctx.set_chain_mode(mode)
        res["t%d" % mode] = run_track(d["flows_f"], occ, None, None, r)
        res["c%d" % mode] = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r)
    ctx.set_chain_mode(0)
    row = {"shape": [T, H, W, r], "track_equal": same(res["t1"], res["t2"]), "connect_equal": same(res["c1"], res["c2"]),
           "track_vs_connect": same(res["t1"], res["c1"]), "modes": [int(res[k].info["chain_mode"]) for k in ("t1", "t2", "c1", "c2")],
           "n_traj": len(res["t1"])}
    out["ok"] &= row["track_equal"] and row["connect_equal"] and row["track_vs_connect"] and row["modes"] == [1, 2, 1, 2]
    out["cases"].append(row)
    if (T, H, W) != (101, 1080, 1920):
        del d, occ
        continue
    # ---- timing on the headline shape ----
    us = lambda pr, k: 1e3 * pr[k]["total_ms"] / max(pr[k]["launches"], 1)
    def timed(fn, mode, n=10):
        ctx.set_chain_mode(mode)
        fn(); fn()
        ctx.set_profiling(1)
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            inf = fn()
            torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        pr = ctx.profile()
        ctx.set_profiling(False)
        ctx.set_chain_mode(0)
        return float(np.median(ts)), pr, inf
    ms_t, pr_t, inf_t = timed(lambda: run_track(d["flows_f"], occ, None, None, r, return_device=True), 2)
    ms_c, pr_c, inf_c = timed(lambda: run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True), 0)
    out["timing_1080p"] = {"track_ms": ms_t, "track_chain_launch_us": us(pr_t, "chain_step"), "track_chain_us_per_step": us(pr_t, "chain_step") / (T - 1),
                           "track_mode": int(inf_t.chain_mode), "connect_ms": ms_c, "connect_chain_launch_us": us(pr_c, "chain_step"),
                           "connect_us_per_step": us(pr_c, "chain_step") / (T - 1), "connect_mode": int(inf_c.chain_mode),
                           "finalize_us": us(pr_c, "finalize")}
print(json.dumps(out))
