#!/bin/bash
# round 5, call y4: the batch of the stress (seed 3, batch 86) whose two-flow sequence differs between psfm_connect_batch and psfm_connect, beside the oracle
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/stress_batch.py 120 3 86 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_y4_repro.txt | tail -12
