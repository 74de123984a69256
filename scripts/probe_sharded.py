#!/usr/bin/env python3
"""Where the time of the track-sharded mode goes at world = 1 (no collectives): Stage A, chain steps, solves, finish."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import torch
import psfm_dist, psfm_synth
from point_trajectory.shard import HipShardEngine, flow_check_slice
H, W, T, r = 1080, 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 101, 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=1, sigma=0.05, n_occluders=2, stride2=True, device="cuda")
comm = psfm_dist.TorchComm()
eng = HipShardEngine()
for rep in range(2):
    t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    occ = psfm_dist.flow_check_sharded(d["flows_f"], d["flows_b"], 1.0, flow_check_slice, comm=comm)
    occ2 = psfm_dist.flow_check_sharded(d["flows_f2"], d["flows_b2"], 1.0, flow_check_slice, comm=comm)
    torch.cuda.synchronize(); t["stage_a"] = time.perf_counter() - t0
    GW, GH = (W + r - 1) // r, (H + r - 1) // r
    eng.begin(T - 1, H, W, r, 0, GW * GH, True)
    reduce = psfm_dist.make_reduce(comm=comm)
    ts = tv = 0.0
    for f in range(T - 1):
        a = time.perf_counter()
        x = eng.step(f, d["flows_f"][f], occ[f]); comm.all_reduce_max_(x); eng.after_exchange(f, x)
        b = time.perf_counter()
        if f >= 1:
            eng.solve(f, d["flows_f"][f - 1], d["flows_f"][f], d["flows_f2"][f - 1], occ2[f - 1], reduce)
        c = time.perf_counter()
        ts += b - a; tv += c - b
    torch.cuda.synchronize(); t["steps_host"] = ts; t["solves_host"] = tv
    a = time.perf_counter()
    info, keys = eng.finish_device(r, W)
    ids, n = psfm_dist.global_ids_device(keys, comm)
    torch.cuda.synchronize(); t["finish"] = time.perf_counter() - a
    t["total"] = time.perf_counter() - t0
    t["counters"] = dict(eng.counters); t["k"] = eng.k; t["n_traj"] = n
    print(json.dumps(t))
