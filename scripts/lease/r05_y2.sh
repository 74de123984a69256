#!/bin/bash
# round 5, call y2: randomised stress of psfm_connect_batch against one psfm_connect per sequence
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python scripts/stress_batch.py 60 1 > gpurun_out/r05_y2_stress_batch.txt 2>&1
echo "rc=$?" >> gpurun_out/r05_y2_stress_batch.txt
tail -12 gpurun_out/r05_y2_stress_batch.txt
