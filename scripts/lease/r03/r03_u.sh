#!/bin/bash
# round 3: per-block phase timeline of the launch chain's iteration kernel on the hard 1080p flows (where do its ~17 us go?)
O=$GRAFT_REPO_ROOT/gpurun_out/r03_u; mkdir -p $O
PSFM_PROBE_HARD=1 PSFM_PC_PERSIST=0 PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_tl.so timeout 200 python scripts/timeline_solver.py > $O/timeline.txt 2> $O/timeline.err; cat $O/timeline.txt; tail -3 $O/timeline.err
