#!/bin/bash
# round 5, call t: the hiccup -- local engine without its budget / exchange engine with a budget (diagnosis)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { l=$1; shift
  env "$@" PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | tail -40 | sed "s/^/$l /" >> gpurun_out/r05_t_trace.txt
}
for i in 1 2 3; do
  run no_budget PSFM_SHARD_DIAG=no_budget
  run budget_only PSFM_SHARD_DIAG=budget_only PSFM_SHARD_LOCAL=0
  run local0 PSFM_SHARD_LOCAL=0
done
grep ms_per_sequence gpurun_out/r05_t_trace.txt | cut -c1-120
