#!/bin/bash
# round 5, call zc: batches of 64 sequences (the entry point's maximum) in the stress; the sharded tests after the abort-path refactor; stress_persist big
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PSFM_STRESS_B=64 timeout 900 python scripts/stress_batch.py 12 31 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600 | tee gpurun_out/r05_zc_stress_b64.txt
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -W ignore 2>&1 | tail -2 | tee gpurun_out/r05_zc_tests.txt
timeout 900 python scripts/stress_persist.py 24 5 big 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r05_zc_stress_persist_big.txt
