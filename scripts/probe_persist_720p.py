"""One build of the library (PSFM_HIP_LIB) at 720p x 101, sample_ratio 2 (230 k lanes: resident at 4 waves per SIMD too): exactness
of the persistent loop against per-frame launches, us per step on ready maps and with flow_check fused.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track, run_connect

H, W, T, r = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (720, 1280, 101, 2)
ctx = _hip.context()
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
same = lambda A, B: bool(len(A) == len(B) and np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length) and np.array_equal(A.xy, B.xy))
ctx.set_chain_mode(1); ref = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r)
ctx.set_chain_mode(2); got = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r); got_t = run_track(d["flows_f"], occ, None, None, r)
out = {"lib": os.path.basename(_hip.LIB_PATH), "shape": [H, W, T, r], "mode": int(got.info["chain_mode"]), "connect_equal": same(ref, got),
       "track_equal": same(ref, got_t)}
us = lambda pr, k: 1e3 * pr[k]["total_ms"] / max(pr[k]["launches"], 1)
def timed(fn, n=10):
    fn(); fn()
    ctx.set_profiling(1)
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); inf = fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    pr = ctx.profile(); ctx.set_profiling(False)
    return float(np.median(ts)), pr, inf
ms_t, pr_t, inf_t = timed(lambda: run_track(d["flows_f"], occ, None, None, r, return_device=True))
ms_c, pr_c, inf_c = timed(lambda: run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True))
out.update({"track_ms": ms_t, "track_us_per_step": us(pr_t, "chain_step") / (T - 1), "track_mode": int(inf_t.chain_mode),
            "connect_ms": ms_c, "connect_us_per_step": us(pr_c, "chain_step") / (T - 1), "connect_mode": int(inf_c.chain_mode)})
ctx.set_chain_mode(0)
print(json.dumps(out))
