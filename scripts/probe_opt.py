"""GPU timing probe for track_optimize (configs[2]/[3] shapes); not the official bench."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track

H, W, T, r = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (436, 1024, 50, 2)))
sigma = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
d = psfm_synth.synth_sequence_torch(T, H, W, seed=2, sigma=sigma, n_occluders=2, stride2=True)
torch.cuda.synchronize()
ctx = _hip.context()
for it in range(3):
    ctx.set_profiling(1 if it == 2 else 0)
    torch.cuda.synchronize(); t0 = time.time()
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], 1.0)
    torch.cuda.synchronize(); t1 = time.time()
    info = run_track(d["flows_f"], occ, d["flows_f2"], occ2, r, return_device=True)
    torch.cuda.synchronize(); t2 = time.time()
    print("iter", it, "flow_check x2 %.3f ms  track_optimize %.3f ms  points %d trajs %d solves %d iters %d -> %.3e points/s" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, info.n_points, info.n_traj, info.n_solves, info.solver_iterations, info.n_points / (t2 - t0)))
print(ctx.profile())
if "--cpu" in sys.argv:
    from oracle import oracle as orc
    k = 6
    ff = list(d["flows_f"][:k].cpu().numpy()); f2 = list(d["flows_f2"][:k - 1].cpu().numpy())
    oo = list(occ[:k].cpu().numpy()); o2 = list(occ2[:k - 1].cpu().numpy())
    t0 = time.time(); R = orc.track_optimize(ff, f2, oo, o2, r); dt = time.time() - t0
    print("oracle (1 core) %d flows: %.2f s, %d points -> %.3e points/s; iters %s" % (k, dt, R.n_points, R.n_points / dt, [s["iterations"] for s in R.solves]))
