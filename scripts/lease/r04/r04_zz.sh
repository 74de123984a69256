#!/bin/bash
# round 4, final sources: the whole -m gpu suite, then the profile set of scripts/profile_round4.sh (kernel stats, PMC passes, bench line,
# the files bench.py replays); the suite's log goes to its own directory (the summariser removes gpurun_out/<tag>)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zz_tests; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc $?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
bash scripts/profile_round4.sh r04_zz > $O/profile.log 2>&1; tail -3 $O/profile.log
