#!/usr/bin/env python3
"""ONE hard 1080p sequence over TWO ranks that share the box's single GPU (two processes, gloo for the host-side collectives): the
cross-rank resident solve (psfm_shard_solve_peer: granule areas over IPC, each process's launch on half of the block slots) against
the exchange form (PSFM_SHARD_PEER=0) and against ONE psfm_connect call on the same tensors.  Prints one JSON line.

    python scripts/probe_peer_two_ranks.py [frames=101] [dist=hard|realistic|clean]

What this measures and what it does not: both ranks' kernels run on ONE device (they share its CUs and its memory system), so the
figure is an upper bound on what the hand-off protocol costs, not a two-GPU speed-up; no xGMI link is crossed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import numpy as np

H, W, RATIO, THRES = 1080, 1920, 2, 1.0


def worker(rank, world, port, frames, dist_name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth
        from point_trajectory.shard import HipShardEngine, flow_check_slice
        from point_trajectory.trajectory import run_connect
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        kw = {"hard": psfm_synth.HARD, "realistic": dict(psfm_synth.REALISTIC, realistic=True), "clean": dict(sigma=0.05, n_occluders=2)}[dist_name]
        d = psfm_synth.synth_sequence_torch(frames, H, W, seed=6, stride2=True, device=dev, **kw)
        out = {}
        for peer in ("1", "0"):
            os.environ["PSFM_SHARD_PEER"] = peer
            eng = HipShardEngine()
            ms = []
            for rep in range(3):
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                part = psfm_dist.connect_sharded(eng, d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO, flow_check_slice,
                                                 keep_on_device=True)
                torch.cuda.synchronize(); dist.barrier()
                ms.append(1e3 * (time.perf_counter() - t0))
            out["peer" if peer == "1" else "exchange"] = {"ms": ms, "counters": dict(eng.counters), "n_traj": int(part["n_traj"]),
                                                           "iterations": int(part["solver_iterations"]), "solves": int(part["n_solves"])}
        if rank == 0:      # the one-GPU call on the same tensors, rank 1 idle
            ms = []
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO, return_device=True)
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - t0))
            out["psfm_connect"] = {"ms": ms, "n_traj": int(info.n_traj), "iterations": int(info.solver_iterations)}
        dist.barrier()
        ret[rank] = out
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 101
    dist_name = sys.argv[2] if len(sys.argv) > 2 else "hard"
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(2, 36500 + os.getpid() % 2000, frames, dist_name, ret), nprocs=2, join=True)
    r0 = ret[0]
    best = lambda k: min(r0[k]["ms"])
    print(json.dumps({"frames": frames, "flows": dist_name, "world": 2, "one_gpu": True,
                      "peer_ms": best("peer"), "exchange_ms": best("exchange"), "psfm_connect_ms": best("psfm_connect"),
                      "peer_over_psfm_connect": best("peer") / best("psfm_connect"), "exchange_over_psfm_connect": best("exchange") / best("psfm_connect"),
                      "counters_peer": r0["peer"]["counters"], "counters_exchange": r0["exchange"]["counters"],
                      "same_counts": r0["peer"]["n_traj"] == r0["psfm_connect"]["n_traj"] == r0["exchange"]["n_traj"] and
                                     r0["peer"]["iterations"] == r0["psfm_connect"]["iterations"], "all_ms": {k: r0[k]["ms"] for k in r0}}))
