"""Random sequences through track_optimize on the device vs the CPU oracle: ids / lengths / per-solve iteration counts
and terminations must be equal, positions within 1e-9 px (bit-equal whenever no solve left the Gauss-Newton path).
Usage: python scripts/stress_optimize.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import psfm_synth
from point_trajectory.track_optimize import track_optimize
from oracle import oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = exact = 0
t0 = time.time()
for k in range(n_cases):
    H, W = int(rng.integers(40, 200)), int(rng.integers(40, 240))
    T, r = int(rng.integers(4, 14)), int(rng.integers(1, 5))
    sigma, nocc = float(rng.uniform(0.02, 0.6)), int(rng.integers(0, 4))
    d = psfm_synth.synth_sequence(T, H, W, seed=int(rng.integers(1 << 30)), sigma=sigma, n_occluders=nocc, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    R = track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    same = (len(R) == O.n_traj and np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
            and [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves]
            and [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves])
    err = float(np.abs(R.xy - O.xy).max()) if same else float("inf")
    non_gn = sum(s["dogleg_nonGN"] for s in O.solves)
    if not same or err > 1e-9 or (non_gn == 0 and err != 0.0):
        bad += 1
        print("MISMATCH case %d: %dx%d T=%d r=%d sigma=%.3f occluders=%d: same=%s err=%.3e nonGN=%d" % (k, H, W, T, r, sigma, nocc, same, err, non_gn))
    exact += int(err == 0.0)
print("%d cases, %d mismatches, %d bit-equal, %.1f s" % (n_cases, bad, exact, time.time() - t0))
