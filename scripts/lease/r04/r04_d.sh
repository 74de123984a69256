#!/bin/bash
# round 4: iteration 0 inside the resident launch: the solver's GPU tests, timing of the hard 1080p sequence (+ the separate-init form)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_d; mkdir -p $O
cd $GRAFT_REPO_ROOT
scripts/micro/dpp_check.bin > $O/dpp.log 2>&1; tail -1 $O/dpp.log
timeout 900 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/solver_tests.log 2>&1; echo "tests rc $?" >> $O/solver_tests.log; tail -5 $O/solver_tests.log
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard.json 2> $O/hard.err; cat $O/hard.json
PSFM_PC_INIT_INSIDE=0 PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_sep.json 2> $O/hard_sep.err; cat $O/hard_sep.json
