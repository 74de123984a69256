#!/bin/bash
# round 5, call zj: randomised stress of the native .flo ingest
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/stress_ingest.py 300 1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 | tee gpurun_out/r05_zj_stress_ingest.txt
