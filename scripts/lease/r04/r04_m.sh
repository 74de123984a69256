#!/bin/bash
# round 4: the full -m gpu suite, then the round's profile collection (scripts/profile_round4.sh)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_m; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "tests rc $?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
bash scripts/profile_round4.sh r04_m1 > $O/profile.log 2>&1; tail -25 $O/profile.log
