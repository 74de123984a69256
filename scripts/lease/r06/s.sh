#!/bin/bash
# round 6, call s: the GPU suite twice over (flakiness check), the way the driver runs it (-x -q)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do SECONDS=0; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; echo "run $i: $SECONDS s"; done | tee gpurun_out/r06_s_suite_twice.txt
