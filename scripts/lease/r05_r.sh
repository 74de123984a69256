#!/bin/bash
# round 5, call r: what makes the sequence after the first local redo 40-50 ms slower (window 128..255)?  redo by launches / one slot per thread / default
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { # label, env...
  l=$1; shift
  env "$@" PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | tail -40 | sed "s/^/$l /" >> gpurun_out/r05_r_trace.txt
}
for i in 1 2; do
  run persist0 PSFM_PC_PERSIST=0
  run slots1 PSFM_PC_SLOTS=1
  run default X=1
  run margin1 PSFM_SHARD_K_MARGIN=1
done
grep ms_per_sequence gpurun_out/r05_r_trace.txt | cut -c1-200
