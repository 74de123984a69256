#!/bin/bash
# round 5, call k: adaptive launch coverage (ScanNet-sized, 1000 frames: the trimmed launches were outgrown around frame 640 and the batch ran twice),
# batch tests incl. disk to disk, where a small sequence's disk-to-disk time goes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > gpurun_out/r05_k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_k_tests.log; tail -4 gpurun_out/r05_k_tests.log
timeout 900 python - > gpurun_out/r05_k_scannet1000.txt 2>&1 <<'PY'
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'particle-sfm_amd')
import torch, psfm_synth
from point_trajectory.trajectory import run_connect_batch, run_connect
from point_trajectory import _hip
H, W, T, R, thres, B = 480, 640, 1000, 1, 3.0, 4
data = [psfm_synth.synth_sequence_torch(T, H, W, seed=4 + k, sigma=0.05, n_occluders=2, stride2=True) for k in range(B)]
seqs = [(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"]) for d in data]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctxs, infos = run_connect_batch(seqs, thres, R)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("batch of %d x %d frames: %.1f ms per batch, %.2f ms per sequence; lanes peak %s of capacity %s" % (B, T, 1e3 * dt, 1e3 * dt / B,
          [int(i.n_lanes_peak) for i in infos], [int(i.lane_capacity) for i in infos]), flush=True)
ctxs[0].set_profiling(1)
ctxs, infos = run_connect_batch(seqs, thres, R); torch.cuda.synchronize()
pr = ctxs[0].profile(); ctxs[0].set_profiling(0)
print("frame launches %d, avg %.1f us; finalize %.2f ms" % (pr["solver"]["launches"], 1e3 * pr["solver"]["total_ms"] / pr["solver"]["launches"], pr["finalize"]["total_ms"]))
run_connect(*seqs[0], thres, R, return_device=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
info = run_connect(*seqs[0], thres, R, return_device=True)
torch.cuda.synchronize(); print("one psfm_connect: %.2f ms; lanes peak %d of %d (grid %d)" % (1e3 * (time.perf_counter() - t0), info.n_lanes_peak, info.lane_capacity, H * W))
PY
cat gpurun_out/r05_k_scannet1000.txt | grep -v amdgpu
timeout 600 python scripts/probe_e2e_small.py > gpurun_out/r05_k_e2e_small.txt 2>&1; grep -v amdgpu gpurun_out/r05_k_e2e_small.txt
