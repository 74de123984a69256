#!/bin/bash
# round 5, call f: resident solves of several host threads side by side (psfm_ctx_set_resident_budget)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -k "side_by_side or exclusive_sequence or giving_up or gives_up" > gpurun_out/r05_f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_f_tests.log; tail -5 gpurun_out/r05_f_tests.log
timeout 1200 python scripts/probe_resident_budget.py gpurun_out/r05_f_resident_budget.json > gpurun_out/r05_f_resident_budget.txt 2>&1
cat gpurun_out/r05_f_resident_budget.txt | tail -30
