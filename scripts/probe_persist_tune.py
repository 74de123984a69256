"""GPU probe: the persistent loop's measurement knobs (PSFM_PP_TUNE=a,b,c,d, read at every launch) on the 1080p x 101 headline
sequence: HIP-event time per step of the loop on ready maps (psfm_track) and with flow_check fused (psfm_connect).
    python scripts/probe_persist_tune.py "0,0,0,0" "4,8,0,0" ..."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track, run_connect

T, H, W, r = 101, 1080, 1920, 2
ctx = _hip.context()
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
us = lambda pr, k: 1e3 * pr[k]["total_ms"] / max(pr[k]["launches"], 1)


def timed(fn, mode, n=12):
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


ref = None
for tune in (sys.argv[1:] or ["0,0,0,0"]):
    os.environ["PSFM_PP_TUNE"] = tune
    ms_t, pr_t, inf_t = timed(lambda: run_track(d["flows_f"], occ, None, None, r, return_device=True), 2)
    ms_c, pr_c, inf_c = timed(lambda: run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True), 0)
    key = (int(inf_t.n_traj), int(inf_t.n_points), int(inf_c.n_traj), int(inf_c.n_points))
    ref = ref or key
    print(json.dumps({"tune": tune, "track_us_per_step": round(us(pr_t, "chain_step") / (T - 1), 3), "track_ms": round(ms_t, 3),
                      "connect_us_per_step": round(us(pr_c, "chain_step") / (T - 1), 3), "connect_ms": round(ms_c, 3),
                      "modes": [int(inf_t.chain_mode), int(inf_c.chain_mode)], "counts_ok": key == ref}), flush=True)
