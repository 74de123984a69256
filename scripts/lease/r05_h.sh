#!/bin/bash
# round 5, call h: after the resident-header refactor and the concurrent redo of sequences that leave a batch: solver + batch tests, realistic batch probe, hard timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_batch.py -x -q > gpurun_out/r05_h_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_h_tests.log; tail -5 gpurun_out/r05_h_tests.log
timeout 900 python scripts/probe_batch.py gpurun_out/r05_h_probe_batch_real.json sintel_real davis_real > gpurun_out/r05_h_probe_batch_real.txt 2>&1
tail -12 gpurun_out/r05_h_probe_batch_real.txt
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 600 python scripts/probe_solver.py 2>/dev/null | tail -3 > gpurun_out/r05_h_probe_hard.txt; cat gpurun_out/r05_h_probe_hard.txt | cut -c1-600
