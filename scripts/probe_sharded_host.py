#!/usr/bin/env python3
"""Where does ONE sequence through the sharded engine at world size 1 spend its time -- on the host (Python enqueueing) or on the
device?  Times the enqueue loop against the final synchronisation and prints the host profile of one run.

    python scripts/probe_sharded_host.py [frames=401]
"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import torch
import bench_common as bench
import bench_extras
import psfm_dist
import psfm_synth
from point_trajectory.shard import HipShardEngine, flow_check_slice

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 401
dev = torch.device("cuda", 0)
d = psfm_synth.synth_sequence_torch(frames, bench.H, bench.W, seed=1, sigma=0.05, n_occluders=2, stride2=True, device=dev)
comm, eng = psfm_dist.TorchComm(), HipShardEngine()


def once():
    return psfm_dist.connect_sharded(eng, d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], bench.THRES, bench.RATIO,
                                     flow_check_slice, comm=comm, keep_on_device=True)


once(); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); once(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host returned after %.2f ms, device done after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
pr = cProfile.Profile(); pr.enable(); once(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
