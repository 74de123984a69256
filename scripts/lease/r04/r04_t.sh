#!/bin/bash
# round 4: spatially sorted participant lists (PSFM_PC_SORT=0: lane order): parity + hard-sequence timing A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04_t; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
for v in 1 0 1 0; do
  PSFM_PC_SORT=$v PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=d['adaptive']; print('PSFM_PC_SORT=$v', 'ms/seq %.3f' % a['ms_per_sequence'])" | tee -a $O/ab.txt
done
