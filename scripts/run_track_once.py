"""Profiling target: the persistent loop on ready maps (psfm_track, chain mode 2) or fused (psfm_connect), 1080p x 101, N runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track, run_connect
what = sys.argv[1] if len(sys.argv) > 1 else "track"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T, H, W, r = 101, 1080, 1920, 2
ctx = _hip.context()
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
ctx.set_chain_mode(2 if what == "track" else 0)
for _ in range(n):
    if what == "track":
        info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    else:
        info = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True)
torch.cuda.synchronize()
print("mode", int(info.chain_mode), "points", int(info.n_points))
