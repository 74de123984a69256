"""Full-length runs of BASELINE.json configs[3] and configs[4] shapes on ONE MI355X (synthetic stand-ins):
  cfg 4: 1080p, 401 frames, sample_ratio 2, full path-consistency optimise
  cfg 5: 480x640, 1000 frames, sample_ratio 1 (dense), flow_check_thres 3.0, full optimise
Prints time per sequence, throughput, solver statistics and size-independent invariants; compares the first frames
with the CPU oracle.  (Not part of the test suite: ~40 GB of HBM, a minute of data synthesis.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch
import psfm_synth
from oracle import oracle as orc
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_connect, run_track, _result_to_host
from point_trajectory import _hip

for name, (H, W, T, r, thres) in {"cfg4": (1080, 1920, 401, 2, 1.0), "cfg5": (480, 640, 1000, 1, 3.0)}.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    t0 = time.time()
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=1 if name == "cfg4" else 4, sigma=0.05, n_occluders=2, stride2=True)
    torch.cuda.synchronize()
    print("%s: synthesised %d frames %dx%d in %.1f s" % (name, T, H, W, time.time() - t0), flush=True)
    ctx = _hip.context()
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], thres, r, return_device=True)
        torch.cuda.synchronize(); dt = time.time() - t0
    print("%s: %.1f ms per sequence, %.3e trajectory-points/s; trajectories %d points %d lanes_peak %d/%d; solves %d iterations %d"
          % (name, dt * 1e3, info.n_points / dt, info.n_traj, info.n_points, info.n_lanes_peak, info.lane_capacity,
             info.n_solves, info.solver_iterations), flush=True)
    R = _result_to_host(ctx, info)
    last = R.birth + R.length - 1
    GW, GH = (W + r - 1) // r, (H + r - 1) // r
    ok = (R.off[-1] == R.n_points and np.array_equal(np.diff(R.off), R.length) and (np.diff(last) >= 0).all()
          and int((R.birth == 0).sum()) == GW * GH and last.max() == T - 1 and np.isfinite(R.xy).all()
          and all(s["termination"] in (0, 1, 2) for s in R.solve_stats))
    k = 4
    _, occ = flow_check_device(d["flows_f"][:k], d["flows_b"][:k], thres)
    _, occ2 = flow_check_device(d["flows_f2"][:k - 1], d["flows_b2"][:k - 1], thres)
    O = orc.track_optimize(list(d["flows_f"][:k].cpu().numpy()), list(d["flows_f2"][:k - 1].cpu().numpy()),
                           list(occ.cpu().numpy()), list(occ2.cpu().numpy()), r)
    Rk = run_track(d["flows_f"][:k], occ, d["flows_f2"][:k - 1], occ2, r)
    same = np.array_equal(Rk.birth, O.birth) and np.array_equal(Rk.length, O.length)
    print("%s: invariants %s; first %d flows vs oracle: ids/lengths equal %s, max|dxy| %.2e px; iterations/solve min %d max %d"
          % (name, ok, k, same, float(np.abs(Rk.xy - O.xy).max()) if same else float("nan"),
             min(s["iterations"] for s in R.solve_stats), max(s["iterations"] for s in R.solve_stats)), flush=True)
    del d, R, Rk
    torch.cuda.empty_cache()
