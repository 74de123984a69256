#!/bin/bash
# round 4: the control step in every lane of wave 0 (uniform branches) vs under a one-lane exec mask
O=$GRAFT_REPO_ROOT/gpurun_out/r04_h; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "persistent or gives_up" > $O/solver_tests.log 2>&1; echo "tests rc $?" >> $O/solver_tests.log; tail -3 $O/solver_tests.log
for v in default lane0 default lane0; do
  if [ $v = default ]; then unset PSFM_HIP_LIB; else export PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_$v.so; fi
  PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['adaptive']['ms_per_sequence'])" | tee -a $O/ab.txt
done
PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_tl.so timeout 300 python scripts/timeline_resident.py 2>/dev/null | sed -n 4,5p
