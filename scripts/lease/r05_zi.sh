#!/bin/bash
# round 5, call zi: the stress runs at DAVIS / Sintel-sized frames (PSFM_STRESS_BIG=1) and stress_optimize big
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PSFM_STRESS_BIG=1 timeout 600 python scripts/stress_batch.py 16 41 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500 | tee gpurun_out/r05_zi_stress_big.txt
PSFM_STRESS_BIG=1 timeout 600 python scripts/stress_sharded.py 24 41 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500 | tee -a gpurun_out/r05_zi_stress_big.txt
timeout 900 python scripts/stress_optimize.py 16 6 big 2>&1 | grep -v amdgpu.ids | grep "MISMATCH\|\"cases\"" | tee -a gpurun_out/r05_zi_stress_big.txt
