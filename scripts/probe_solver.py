#!/usr/bin/env python3
"""track_optimize end to end (psfm_connect with the stride-2 stacks) under the solver modes of psfm_ctx_set_solver.

    python scripts/probe_solver.py [H W T r] ; PSFM_FUSED_WAVES=3|4 selects the fused kernel's register target
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect

H, W, T, r = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (1080, 1920, 101, 2)
# PSFM_PROBE_HARD=1: SURVEY 8(d)'s second distribution (sigma 0.3, 5 % occluder area) -- every solve goes through the launch chain
dist = psfm_synth.HARD if os.environ.get("PSFM_PROBE_HARD") else dict(sigma=0.05, n_occluders=2)
d = psfm_synth.synth_sequence_torch(T, H, W, seed=6 if os.environ.get("PSFM_PROBE_HARD") else 5, stride2=True, device="cuda", **dist)
ctx = _hip.context()
out = {"shape": [H, W, T, r], "fused_waves": os.environ.get("PSFM_FUSED_WAVES", "3"), "seq_waves": os.environ.get("PSFM_SEQ_WAVES", "default"),
       "pc_blocks": os.environ.get("PSFM_PC_BLOCKS", "default"), "flows": dict(dist)}
MODES = {"chain": (1, 0), "fused": (2, 0), "adaptive": (0, 0), "fused_k4": (2, 4), "fused_k2": (2, 2), "fused_k5": (2, 5)}
names = os.environ.get("PSFM_PROBE_MODES", "chain,fused,adaptive,fused_k4").split(",")
for name, mode, k in [(n, *MODES[n]) for n in names]:
    ctx.set_solver(mode, k)
    for _ in range(3):
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], 1.0, r, return_device=True)
    n = 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], 1.0, r, return_device=True)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    ctx.set_profiling(1)
    for _ in range(n):
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], 1.0, r, return_device=True)
    torch.cuda.synchronize()
    pr = ctx.profile()
    ctx.set_profiling(0)
    import numpy as np
    from point_trajectory.trajectory import _result_to_host
    Rh = _result_to_host(ctx, info)
    nf = T - 1
    birth = Rh.birth.astype(np.int64); last = birth + Rh.length - 1
    tb = np.bincount(birth, minlength=nf + 3).cumsum(); tl = np.bincount(last, minlength=nf + 3).cumsum()
    its = [s_["iterations"] for s_ in Rh.solve_stats]
    n3 = [float(tb[f - 1] - tl[f]) for f in range(1, nf)]
    scale = {"tracks_per_solve": float(np.mean(n3)), "iterations_per_solve": float(np.mean(its)) if its else 0.0,
             "track_iterations_per_solve": float(np.mean([k * n for k, n in zip(its, n3)])) if len(its) == len(n3) else None}
    del Rh
    out[name] = {"ms_per_sequence": ms, "points": int(info.n_points), "iters": int(info.solver_iterations), **scale,
                 "solves": int(info.n_solves), "counters": ctx.solver_counters(),
                 "solver_ms_per_seq": pr["solver"]["total_ms"] / n, "solver_launches_per_seq": pr["solver"]["launches"] / n,
                 "chain_ms_per_seq": pr["chain_step"]["total_ms"] / n}
ctx.set_solver(0, 0)
print(json.dumps(out))
