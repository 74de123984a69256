#!/bin/bash
# round 3: the whole GPU suite on the final binary
O=$GRAFT_REPO_ROOT/gpurun_out/r03_x; mkdir -p $O
(time timeout 560 python -m pytest tests -q -m gpu) > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
