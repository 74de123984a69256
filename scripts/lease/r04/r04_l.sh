#!/bin/bash
# round 4: the fused solve's grid totals as two hops of tagged granules (default) vs the two-level tickets (variant "tickets")
O=$GRAFT_REPO_ROOT/gpurun_out/r04_l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
for v in default tickets default tickets; do
  if [ $v = default ]; then unset PSFM_HIP_LIB; else export PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_$v.so; fi
  for shape in "1080 1920 101 2" "436 1024 50 2" "480 640 300 1"; do
    PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py $shape 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=d['adaptive']; print('$v', d['shape'], 'ms/seq %.3f' % a['ms_per_sequence'], 'solver launch us %.2f' % (1e3*a['solver_ms_per_seq']/max(a['solver_launches_per_seq'],1)), a['counters'])" | tee -a $O/ab.txt
  done
done
