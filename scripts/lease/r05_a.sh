#!/bin/bash
# round 5, call a: the resident give-up fix + how much host-thread concurrency buys on the small BASELINE shapes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -k "gives_up or giving_up or redo_of or persistent_launch_is_the_same" > gpurun_out/r05_a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_a_tests.log
tail -5 gpurun_out/r05_a_tests.log
timeout 600 python scripts/probe_concurrent_small.py gpurun_out/r05_a_concurrent_small.json > gpurun_out/r05_a_concurrent_small.txt 2>&1
cat gpurun_out/r05_a_concurrent_small.txt | tail -20
