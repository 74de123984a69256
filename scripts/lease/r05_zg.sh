#!/bin/bash
# round 5, call zg: what a speculated reject branch could hide -- timing builds of the resident solve with a second track phase beside the
# all-reduce (PC_RES_WHATIF=1: on the three waves that do not run the hand-off; 2: on all four in front of it); hard flows, 1080p x 101
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in whatif0 whatif1 whatif2 whatif0 whatif1; do
  PSFM_HIP_LIB=$PWD/particle-sfm_amd/lib/variants/libpsfm_hip_$v.so PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2>&1 | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); a=b['adaptive']; print('$v', round(a['ms_per_sequence'],2), 'ms per sequence; solver', round(a['solver_ms_per_seq'],2), 'ms;', a['iters'], 'iterations', a['counters'])" | tee -a gpurun_out/r05_zg_whatif.txt
done
