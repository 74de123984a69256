#!/bin/bash
# round 4, final binary: the whole -m gpu suite, the profile set of scripts/profile_round4.sh (kernel stats, PMC passes, bench line, the
# files bench.py replays), the shape table of the chain policy, and whether parallel pwrite()s to tmpfs scale on the box (track.npy writer)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_x; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc $?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
bash scripts/profile_round4.sh r04_x > $O/profile.log 2>&1; tail -3 $O/profile.log
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/probe_shapes.py > $O/probe_shapes.txt 2>&1; tail -30 $O/probe_shapes.txt
timeout 120 python scripts/micro/pwrite_scale.py > $O/pwrite_scale.txt 2>&1; cat $O/pwrite_scale.txt; nproc >> $O/pwrite_scale.txt
