"""Full-sequence parity of track_optimize against the CPU oracle on the configs[2] shape (436x1024, 50 frames, r=2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track
from oracle import oracle as orc
H, W, T, r = 436, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 50, 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=2, sigma=0.05, n_occluders=2, stride2=True)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
_, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], 1.0)
R = run_track(d["flows_f"], occ, d["flows_f2"], occ2, r)
ff = list(d["flows_f"].cpu().numpy()); f2 = list(d["flows_f2"].cpu().numpy())
oo = list(occ.cpu().numpy()); o2 = list(occ2.cpu().numpy())
t0 = time.time(); O = orc.track_optimize(ff, f2, oo, o2, r); dt = time.time() - t0
same = len(R) == O.n_traj and np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
print("oracle %.1f s; trajectories %d / %d, points %d; ids/lengths equal %s; max|dxy| %.3e px; solver iterations gpu %s oracle %s" % (
    dt, len(R), O.n_traj, R.n_points, same, float(np.abs(R.xy - O.xy).max()) if same else float("nan"),
    sum(s["iterations"] for s in R.solve_stats), sum(s["iterations"] for s in O.solves)))
