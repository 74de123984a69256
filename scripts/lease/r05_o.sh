#!/bin/bash
# round 5, call o: sharded tests again; the clean single sequence with and without the local forms on ONE box (call n: 51 ms where 37.6 were expected)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -W ignore > gpurun_out/r05_o_tests.log 2>&1
echo "sharded tests rc=$? in $SECONDS s" >> gpurun_out/r05_o_tests.log; tail -15 gpurun_out/r05_o_tests.log
for l in 1 0 1 0; do
  PSFM_SHARD_LOCAL=$l timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | sed "s/^/LOCAL=$l /" | tee -a gpurun_out/r05_o_single_sequence.txt
done
