#!/bin/bash
# round 4: kernel-level accounting of the hard path (rocprofv3 stats), solver tests
O=$GRAFT_REPO_ROOT/gpurun_out/r04_e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/solver_tests.log 2>&1; echo "tests rc $?" >> $O/solver_tests.log; tail -3 $O/solver_tests.log
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard.json 2> $O/hard.err; cat $O/hard.json
export TMPDIR=/tmp; cd /tmp
timeout 300 env PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive rocprofv3 --kernel-trace --stats -f csv -d $O/hard_stats -o r04_e_hard -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/hard_under_rocprof.log 2>&1 < /dev/null
find $O/hard_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_e_hard_kernel_stats.csv
rm -rf $O/hard_stats
head -14 $O/r04_e_hard_kernel_stats.csv | cut -c1-200
