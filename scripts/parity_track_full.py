"""Whole-sequence parity of track() against the CPU oracle on the configs[0] shape (480x854, 50 frames, r=4), every
way of running the recurrence."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track, run_connect
from oracle import oracle as orc
H, W, T, r = 480, 854, 50, 4
d = psfm_synth.synth_sequence_torch(T, H, W, seed=3, sigma=0.3, n_occluders=3, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
ff = list(d["flows_f"].cpu().numpy()); fb = list(d["flows_b"].cpu().numpy())
t0 = time.time(); _, occ_o = orc.flow_check(ff, fb, 1.0); O = orc.track(ff, occ_o, r); dt = time.time() - t0
assert np.array_equal(np.stack(occ_o).astype(np.uint8), occ.cpu().numpy())
ctx = _hip.context()
for mode in (1, 2):
    ctx.set_chain_mode(mode)
    for name, R in (("track", run_track(d["flows_f"], occ, None, None, r)), ("connect", run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r))):
        same = len(R) == O.n_traj and np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
        print("mode %d %-7s chain_mode %d: %d trajectories, %d points, bit-identical to the oracle: %s" % (mode, name, R.info["chain_mode"], len(R), R.n_points, same))
ctx.set_chain_mode(0)
print("oracle: %.1f s" % dt)
