#!/bin/bash
# round 4: do the waves of a SIMD lose time by being in the same phase of the frame kernel at the same time?  The waves in odd wave slots
# start k x 3.4 us late (variants stg1..3) against the product library; clean flows, 1080p
O=$GRAFT_REPO_ROOT/gpurun_out/r04_w; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
P=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/libpsfm_hip.so
for lib in $P $V/libpsfm_hip_stg1.so $V/libpsfm_hip_stg2.so $V/libpsfm_hip_stg3.so $P $V/libpsfm_hip_stg1.so $V/libpsfm_hip_stg2.so $V/libpsfm_hip_stg3.so; do
  PSFM_HIP_LIB=$lib PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=d['adaptive']; print('lib=$(basename "$lib")', 'ms/seq %.3f' % a['ms_per_sequence'], 'solver ms/seq %.3f' % a['solver_ms_per_seq'], 'launches', a['solver_launches_per_seq'], a['counters'])" | tee -a $O/ab.txt
done
