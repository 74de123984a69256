#!/bin/bash
# round 5, call s: the hiccup again -- local engine but the fused redo through the exchange form (diagnosis)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { l=$1; shift
  env "$@" PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | tail -40 | sed "s/^/$l /" >> gpurun_out/r05_s_trace.txt
}
for i in 1 2 3; do
  run redo_exchange PSFM_SHARD_DIAG=redo_exchange
  run default X=1
done
grep ms_per_sequence gpurun_out/r05_s_trace.txt | cut -c1-120
