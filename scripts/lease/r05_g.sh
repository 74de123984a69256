#!/bin/bash
# round 5, call g: the sharded engine's redo with rounds enqueued ahead -- thread-rank tests, then hard / clean timing A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/r05_g_tests.log 2>&1
echo "sharded tests rc=$?" >> gpurun_out/r05_g_tests.log; tail -4 gpurun_out/r05_g_tests.log
for ra in 1 8; do
  PSFM_SHARD_ROUNDS_AHEAD=$ra timeout 600 python scripts/probe_single_sequence.py 101 hard 2>/dev/null | tail -1 >> gpurun_out/r05_g_single_sequence.txt
done
timeout 600 python scripts/probe_single_sequence.py 401 2>/dev/null | tail -1 >> gpurun_out/r05_g_single_sequence.txt
cat gpurun_out/r05_g_single_sequence.txt
