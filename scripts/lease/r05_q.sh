#!/bin/bash
# round 5, call q: sharded tests; is the one-time 40 ms hiccup in the sequence after the first resident launch the runtime's scratch reclaim?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -W ignore > gpurun_out/r05_q_tests.log 2>&1
echo "sharded tests rc=$? in $SECONDS s" >> gpurun_out/r05_q_tests.log; tail -15 gpurun_out/r05_q_tests.log
for v in 1 0 1 0 1 0; do
  HSA_NO_SCRATCH_RECLAIM=$v PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | tail -40 | sed "s/^/NORECLAIM=$v /" >> gpurun_out/r05_q_trace.txt
done
grep ms_per_sequence gpurun_out/r05_q_trace.txt | cut -c1-120
