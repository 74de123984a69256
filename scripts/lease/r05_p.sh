#!/bin/bash
# round 5, call p: trace of the one-rank engine's checkpoints on the clean 401-frame sequence (bimodal 37.6 / 49.4 ms in call o)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3; do
  PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | tail -40 >> gpurun_out/r05_p_trace.txt
  echo "----" >> gpurun_out/r05_p_trace.txt
done
tail -60 gpurun_out/r05_p_trace.txt
