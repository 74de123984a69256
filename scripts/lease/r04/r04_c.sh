#!/bin/bash
# round 4: transposed block sums + control tweaks: exchange-tree check, solver tests (persistent == launches bit for bit), timing, timeline
O=$GRAFT_REPO_ROOT/gpurun_out/r04_c; mkdir -p $O
cd $GRAFT_REPO_ROOT
scripts/micro/dpp_check.bin > $O/dpp.log 2>&1; echo "dpp rc $?" >> $O/dpp.log; cat $O/dpp.log
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "persistent or vs_oracle or optimize_location" > $O/solver_tests.log 2>&1; echo "tests rc $?" >> $O/solver_tests.log; tail -5 $O/solver_tests.log
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard.json 2> $O/hard.err; cat $O/hard.json
PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_tl.so timeout 300 python scripts/timeline_resident.py > $O/timeline.txt 2> $O/timeline.err; sed -n 4,5p $O/timeline.txt; tail -8 $O/timeline.txt
