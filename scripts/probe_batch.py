"""psfm_connect_batch against one psfm_connect per sequence on the small BASELINE shapes (configs[0] / [2] / [4]):
    python scripts/probe_batch.py [out.json] [shape ...]
ms per sequence and points/s for B = 1, 2, 4, 8, 16 sequences per batch (different seeds), the per-launch time of the batched frame
kernels from HIP events, and the counts of every sequence against its single-sequence run."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, run_connect_batch

SHAPES = {"davis": ("configs[0] DAVIS 480x854 r4 track", 480, 854, 50, 4, False, 1.0, (1, 2, 4, 8, 16, 32)),
          "sintel": ("configs[2] Sintel 436x1024 r2 optimize", 436, 1024, 50, 2, True, 1.0, (1, 2, 4, 8, 16)),
          "scannet": ("configs[4] ScanNet 480x640 r1 optimize (200 frames)", 480, 640, 200, 1, True, 3.0, (1, 2, 4, 8)),
          "sintel_track": ("Sintel 436x1024 r2 track", 436, 1024, 50, 2, False, 1.0, (1, 4, 16)),
          # psfm_synth.REALISTIC: every solve rejects steps -> the sequences leave the batch behind their first window and are run by
          # psfm_connect, several at a time with shares of the resident block slots
          "sintel_real": ("configs[2] Sintel 436x1024 r2 optimize, REALISTIC flows", 436, 1024, 50, 2, True, 1.0, (1, 2, 4, 8)),
          "davis_real": ("DAVIS 480x854 r4 optimize, REALISTIC flows", 480, 854, 50, 4, True, 1.0, (1, 4, 8))}
which = [a for a in sys.argv[2:]] or ["davis", "sintel", "scannet"]
out = []
for key in which:
    label, H, W, T, R, opt, thres, Bs = SHAPES[key]
    NMAX = max(Bs)
    if key.endswith("_real"):
        data = [psfm_synth.synth_realistic_torch(T, H, W, seed=100 + k, stride2=opt, **psfm_synth.REALISTIC) for k in range(NMAX)]
    else:
        data = [psfm_synth.synth_sequence_torch(T, H, W, seed=100 + k, sigma=0.05, n_occluders=2, stride2=opt) for k in range(NMAX)]
    seqs = [(d["flows_f"], d["flows_b"], d.get("flows_f2") if opt else None, d.get("flows_b2") if opt else None) for d in data]
    # one psfm_connect per sequence (the default mode), one after the other
    ctx = _hip.context()
    singles = []
    for k in range(min(NMAX, 4)):
        info = run_connect(*seqs[k], thres, R, return_device=True)
        singles.append((int(info.n_traj), int(info.n_points), int(info.solver_iterations)))
    reps = 5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for k in range(min(NMAX, 4)):
            info = run_connect(*seqs[k], thres, R, return_device=True)
    torch.cuda.synchronize(); ms1 = 1e3 * (time.perf_counter() - t0) / (reps * min(NMAX, 4))
    pts1 = float(np.mean([s[1] for s in singles]))
    print("%-52s one psfm_connect per sequence: %7.3f ms per sequence, %.3e points/s" % (label, ms1, pts1 / (ms1 * 1e-3)), flush=True)
    out.append({"shape": label, "batch": 0, "ms_per_sequence": ms1, "points_per_s": pts1 / (ms1 * 1e-3)})
    for B in Bs:
        ctxs, infos = run_connect_batch(seqs[:B], thres, R)
        got = [(int(i.n_traj), int(i.n_points), int(i.solver_iterations)) for i in infos]
        same = all(got[k] == singles[k] for k in range(min(B, len(singles))))
        ctxs, infos = run_connect_batch(seqs[:B], thres, R)
        reps = 6
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            ctxs, infos = run_connect_batch(seqs[:B], thres, R)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        pts = sum(int(i.n_points) for i in infos)
        ctxs[0].set_profiling(1)
        ctxs, infos = run_connect_batch(seqs[:B], thres, R)
        torch.cuda.synchronize()
        pr = ctxs[0].profile()
        ctxs[0].set_profiling(0)
        kind = "solver" if opt else "chain_step"
        us = 1e3 * pr[kind]["total_ms"] / max(pr[kind]["launches"], 1)
        row = {"shape": label, "batch": B, "ms_per_sequence": 1e3 * dt / B, "ms_per_batch": 1e3 * dt, "points_per_s": pts / dt,
               "speedup_vs_single": ms1 / (1e3 * dt / B), "frame_launch_us": us, "frame_launches": int(pr[kind]["launches"]),
               "flow_check_ms": pr["flow_check"]["total_ms"], "finalize_ms": pr["finalize"]["total_ms"],
               "modes": sorted(set(int(i.chain_mode) for i in infos)), "counts_equal_single": bool(same)}
        out.append(row)
        print("%-52s batch of %2d: %7.3f ms per sequence (%.2fx), %.3e points/s; frame launch %6.1f us x %d, flow_check %.3f ms, "
              "finalize %.3f ms, modes %s, counts equal %s" % (label, B, row["ms_per_sequence"], row["speedup_vs_single"], row["points_per_s"],
                                                              us, row["frame_launches"], row["flow_check_ms"], row["finalize_ms"], row["modes"], same), flush=True)
    del data, seqs
    torch.cuda.empty_cache()
if len(sys.argv) > 1 and sys.argv[1]:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
