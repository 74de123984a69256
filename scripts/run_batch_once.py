"""One shape, one batch size, a few repetitions of psfm_connect_batch: what scripts/profile_round5.sh runs under rocprofv3
(kernel-trace stats / PMC passes of the batched frame launches).
    python scripts/run_batch_once.py davis|sintel|scannet|sintel_real B [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory.trajectory import run_connect_batch

SHAPES = {"davis": (480, 854, 50, 4, False, 1.0), "sintel": (436, 1024, 50, 2, True, 1.0), "scannet": (480, 640, 200, 1, True, 3.0),
          "sintel_real": (436, 1024, 50, 2, True, 1.0)}
key, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H, W, T, R, opt, thres = SHAPES[key]
if key.endswith("_real"):
    data = [psfm_synth.synth_realistic_torch(T, H, W, seed=100 + k, stride2=opt, **psfm_synth.REALISTIC) for k in range(B)]
else:
    data = [psfm_synth.synth_sequence_torch(T, H, W, seed=100 + k, sigma=0.05, n_occluders=2, stride2=opt) for k in range(B)]
seqs = [(d["flows_f"], d["flows_b"], d.get("flows_f2") if opt else None, d.get("flows_b2") if opt else None) for d in data]
ctxs, infos = run_connect_batch(seqs, thres, R)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctxs, infos = run_connect_batch(seqs, thres, R)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(json.dumps({"shape": key, "H": H, "W": W, "frames": T, "sample_ratio": R, "optimize": opt, "batch": B, "ms_per_batch": 1e3 * dt,
                  "ms_per_sequence": 1e3 * dt / B, "points": int(sum(int(i.n_points) for i in infos)),
                  "trajectories": int(sum(int(i.n_traj) for i in infos)), "solves": int(sum(int(i.n_solves) for i in infos)),
                  "iterations": int(sum(int(i.solver_iterations) for i in infos)), "modes": sorted(set(int(i.chain_mode) for i in infos))}))
