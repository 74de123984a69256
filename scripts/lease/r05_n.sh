#!/bin/bash
# round 5, call n: the sharded engine at world size 1 with the one-GPU call's solver forms for rejecting solves (psfm_shard_solve_local):
# its tests, then single_sequence on clean / hard / realistic-like flows beside the exchange form
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/r05_n_tests.log 2>&1
echo "sharded tests rc=$? in $SECONDS s" >> gpurun_out/r05_n_tests.log; tail -15 gpurun_out/r05_n_tests.log
for a in "101 hard" "401" "101"; do
  timeout 300 python scripts/probe_single_sequence.py $a 2>&1 | tail -1 | tee -a gpurun_out/r05_n_single_sequence.txt
done
