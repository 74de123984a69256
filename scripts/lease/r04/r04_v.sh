#!/bin/bash
# round 4: the fused solve's tap neighbourhood in LDS + the waves' sums through the DPP row tree: parity + clean-flow timing A/B
#   (variants: old = park + gathers, nbo = DPP row tree + gathers, product = DPP row tree + neighbourhood)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_v; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
P=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/libpsfm_hip.so
for lib in $P $V/libpsfm_hip_old.so $V/libpsfm_hip_nbo.so $P $V/libpsfm_hip_old.so $V/libpsfm_hip_nbo.so; do
  PSFM_HIP_LIB=$lib PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=d['adaptive']; print('lib=$(basename "$lib")', 'ms/seq %.3f' % a['ms_per_sequence'], 'solver ms/seq %.3f' % a['solver_ms_per_seq'], 'launches', a['solver_launches_per_seq'], a['counters'])" | tee -a $O/ab.txt
done
