"""GPU probe: psfm_connect with one chain_step launch per frame (chain mode 1) and flow_check on the side stream, 1080p x 101:
end-to-end ms and the chain step's average launch time, for the environment it is started with (PSFM_FC_BG, PSFM_FC_BG_LDS_KB)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect
shapes = [(101, 1080, 1920, 2), (101, 720, 1280, 2), (101, 1080, 1920, 4)]
ctx = _hip.context()
for (T, H, W, r) in shapes:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
    fn = lambda: run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True)
    ctx.set_chain_mode(1)
    fn(); fn()
    ctx.set_profiling(8)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        info = fn()
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    pr = ctx.profile(); ctx.set_profiling(False); ctx.set_chain_mode(0)
    print(json.dumps({"shape": [T, H, W, r], "env": {k: os.environ.get(k) for k in ("PSFM_FC_BG", "PSFM_FC_BG_LDS_KB")},
                      "connect_mode1_ms": round(float(np.median(ts)), 3),
                      "chain_step_us": round(1e3 * pr["chain_step"]["total_ms"] / max(pr["chain_step"]["launches"], 1), 2),
                      "points": int(info.n_points)}), flush=True)
    del d
