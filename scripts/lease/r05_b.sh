#!/bin/bash
# round 5, call b: psfm_connect_batch -- parity tests, then the throughput probe on the small BASELINE shapes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > gpurun_out/r05_b_tests.log 2>&1
echo "batch tests rc=$?" >> gpurun_out/r05_b_tests.log
tail -25 gpurun_out/r05_b_tests.log
timeout 300 python -m pytest tests/test_gpu_solver.py -x -q -k "redo_of" > gpurun_out/r05_b_tests2.log 2>&1
echo "redo tests rc=$?" >> gpurun_out/r05_b_tests2.log
tail -5 gpurun_out/r05_b_tests2.log
timeout 900 python scripts/probe_batch.py gpurun_out/r05_b_probe_batch.json > gpurun_out/r05_b_probe_batch.txt 2>&1
tail -30 gpurun_out/r05_b_probe_batch.txt
