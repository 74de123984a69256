"""Quick GPU timing probe for the track path (not the official bench)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track

T = int(sys.argv[1]) if len(sys.argv) > 1 else 21
H, W, r = 1080, 1920, 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
torch.cuda.synchronize()
ctx = _hip.context()
for it in range(3):
    ctx.set_profiling(it == 2)
    torch.cuda.synchronize(); t0 = time.time()
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    torch.cuda.synchronize(); t1 = time.time()
    info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    torch.cuda.synchronize(); t2 = time.time()
    print("iter", it, "flow_check %.3f ms  track %.3f ms  points %d trajs %d lanes_peak %d  -> %.3e points/s" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, info.n_points, info.n_traj, info.n_lanes_peak, info.n_points / (t2 - t0)))
print(ctx.profile())
print("occ frac", float(occ.float().mean()))
