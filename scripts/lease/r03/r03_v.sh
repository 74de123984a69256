#!/bin/bash
# round 3: the persistent solve's hand-off with block 0 as the fixed reducer and data-tagged packets (four hops instead of seven)
O=$GRAFT_REPO_ROOT/gpurun_out/r03_v; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu) > $O/t1.log 2>&1; tail -4 $O/t1.log | head -2
export PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive
for i in 1 2; do timeout 200 python scripts/probe_solver.py > $O/hard_$i.json 2> $O/hard_$i.err; python -c "
import json; d=json.load(open('$O/hard_$i.json'))['adaptive']; print(round(d['ms_per_sequence'],2), round(d['solver_ms_per_seq'],2), d['iters'], d['counters'])"; done
