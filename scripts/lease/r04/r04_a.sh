#!/bin/bash
# round 4, first run of the resident solve: DPP exchange tree, the solver's GPU tests, the hard 1080p sequence
O=$GRAFT_REPO_ROOT/gpurun_out/r04_a; mkdir -p $O
cd $GRAFT_REPO_ROOT
scripts/micro/dpp_check.bin > $O/dpp.log 2>&1; echo "dpp rc $?" >> $O/dpp.log; cat $O/dpp.log
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu > $O/solver_tests.log 2>&1; echo "tests rc $?" >> $O/solver_tests.log; tail -15 $O/solver_tests.log
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard.json 2> $O/hard.err; tail -2 $O/hard.err; cat $O/hard.json
PSFM_PC_PERSIST=0 PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_launches.json 2> $O/hard_launches.err; cat $O/hard_launches.json
