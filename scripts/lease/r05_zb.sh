#!/bin/bash
# round 5, call zb: more rounds of stress_threads; stress_persist on the final sources
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 2 3; do
  timeout 900 python scripts/stress_threads.py 100 $seed 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-500 | tee -a gpurun_out/r05_zb_stress_threads.txt
done
timeout 900 python scripts/stress_persist.py 300 11 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee gpurun_out/r05_zb_stress_persist.txt
