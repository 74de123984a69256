#!/bin/bash
# round 5, call zm: the stress batch of seed 54 in which psfm_connect reported "no progress in 8 windows"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PSFM_TRACE=1 timeout 600 python scripts/stress_batch.py 150 54 > gpurun_out/r05_zm_seed54.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_zm_seed54.txt | tail -30 | cut -c1-400
grep "^\[psfm" gpurun_out/r05_zm_seed54.txt | tail -5 | cut -c1-300
