#!/bin/bash
# round 6, call b: the GPU suite with this round's new tests (autograd pin, hand-derived solves, stress scripts, capacity retry, budget, cap)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 > gpurun_out/r06_b_gpu_suite.txt
echo "gpu suite: $SECONDS s"; tail -45 gpurun_out/r06_b_gpu_suite.txt
