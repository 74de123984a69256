#!/bin/bash
# round 5, call y8: track_optimize against the oracle on random sequences incl. two-flow ones, lengths just behind a window, realistic flows
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 21 22; do
  timeout 1200 python scripts/stress_optimize.py 400 $seed 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|\"cases\"" | tee -a gpurun_out/r05_y8_stress_optimize.txt
done
