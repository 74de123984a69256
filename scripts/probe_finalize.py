"""GPU probe: finalize of the persistent loop (records -> sort -> offsets -> transpose gather) on the 1080p x 101 headline
sequence for the library PSFM_HIP_LIB selects: HIP-event time of the finalize span, end-to-end psfm_connect, and a checksum of
the result (so that builds can be compared)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect

T, H, W, r = 101, 1080, 1920, 2
ctx = _hip.context()
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
fn = lambda **kw: run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, **kw)
R = fn()
chk = (len(R), int(R.n_points), float(R.xy.sum()), int(R.length.astype(np.int64).dot(np.arange(len(R)) % 1000003)), int(R.birth.sum()))
fn(return_device=True)
ctx.set_profiling(1)
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(return_device=True)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
pr = ctx.profile()
ctx.set_profiling(False)
us = lambda k: 1e3 * pr[k]["total_ms"] / max(pr[k]["launches"], 1)
print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "connect_ms": round(float(np.median(ts)), 3), "finalize_us": round(us("finalize"), 1),
                  "loop_us": round(us("chain_step"), 1), "checksum": chk}))
