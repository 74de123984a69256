#!/bin/bash
# round 5, call w: sharded tests (the one-rank solver forms); bench single_sequence figures with the frozen collector
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -W ignore > gpurun_out/r05_w_tests.log 2>&1
echo "sharded tests rc=$? in $SECONDS s" >> gpurun_out/r05_w_tests.log; tail -25 gpurun_out/r05_w_tests.log
for a in "101 hard" "401"; do
  PSFM_PROBE_GC=freeze timeout 300 python scripts/probe_single_sequence.py $a 2>&1 | tail -1 | tee -a gpurun_out/r05_w_single_sequence.txt
done
