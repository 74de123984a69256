#!/bin/bash
# round 6, call h: the whole GPU suite with durations
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -q --durations=40 2>&1 | tail -60 > gpurun_out/r06_h_gpu_suite.txt
echo "gpu suite: $SECONDS s"; tail -60 gpurun_out/r06_h_gpu_suite.txt
