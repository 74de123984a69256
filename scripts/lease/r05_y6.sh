#!/bin/bash
# round 5, call y6: the solver forms of psfm_connect against the batch's bits (seed 5 batch 8)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PSFM_STRESS_FORMS=1 timeout 600 python scripts/stress_batch.py 9 5 8 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/r05_y6_repro.txt
