import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import numpy as np
import psfm_synth
from oracle import oracle as orc
from point_trajectory import _hip
from point_trajectory.track_optimize import track_optimize
H, W, T, r, seed, sigma, nocc = 64, 96, 20, 1, 54, 0.15, 1
d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
_, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
_, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
print("oracle", O.n_traj, [(s["iterations"], s["successful_steps"], s["dogleg_nonGN"], s["termination"]) for s in O.solves])
ctx = _hip.context()
for mode, k in [(1, 0), (2, 0), (2, 1), (2, 2), (2, 3), (2, 4), (2, 6), (2, 8), (0, 0)]:
    ctx.set_solver(mode, k)
    for T2 in (T,):
        R = track_optimize(d["flows_f"][:T2 - 1], d["flows_f2"][:T2 - 2], occ[:T2 - 1], occ2[:T2 - 2], r)
    same = len(R) == O.n_traj and np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    print(mode, k, "n_traj", len(R), "same", same, "maxerr", float(np.abs(R.xy - O.xy).max()) if same else None, ctx.solver_counters(),
          [s["iterations"] for s in R.solve_stats])
# shorter prefixes in forced fused mode: where does it start to differ?
ctx.set_solver(2, 0)
for T2 in range(4, T + 1):
    Ok = orc.track_optimize(d["flows_f"][:T2 - 1], d["flows_f2"][:T2 - 2], occ[:T2 - 1], occ2[:T2 - 2], r)
    R = track_optimize(d["flows_f"][:T2 - 1], d["flows_f2"][:T2 - 2], occ[:T2 - 1], occ2[:T2 - 2], r)
    same = len(R) == Ok.n_traj and np.array_equal(R.birth, Ok.birth) and np.array_equal(R.length, Ok.length)
    print("T", T2, same, float(np.abs(R.xy - Ok.xy).max()) if same else None, ctx.solver_counters())
