#!/bin/bash
# round 5, call y5: the fix for the solve that outlasts its window's launches: its test, the solver / batch test files, the stress with seeds 3-6
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_batch.py -m gpu -x -q -W ignore > gpurun_out/r05_y5_tests.log 2>&1
echo "tests rc=$? in $SECONDS s" >> gpurun_out/r05_y5_tests.log; tail -8 gpurun_out/r05_y5_tests.log
for seed in 3 4 5 6; do
  timeout 900 python scripts/stress_batch.py 150 $seed 2>&1 | tail -2 | tee -a gpurun_out/r05_y5_stress_batch.txt
done
