#!/bin/bash
# round 4: the GPU tests touched since r04_e (gate yield, mid-solve give-up, 8 thread-ranks), then solver + sharded suites
O=$GRAFT_REPO_ROOT/gpurun_out/r04_j; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -12 $O/tests.log
