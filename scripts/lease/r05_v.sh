#!/bin/bash
# round 5, call v: is the one-time +40 ms a full collection of Python's cyclic GC landing in the timed region?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3; do
  PSFM_PROBE_GC=log PSFM_SHARD_TRACE=1 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | grep -v Warning | grep "gc\]\|128..255 checked\|ms_per_seq" | cut -c1-110 | sed "s/^/log /" | tee -a gpurun_out/r05_v.txt
  PSFM_PROBE_GC=freeze timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | cut -c1-100 | sed "s/^/freeze /" | tee -a gpurun_out/r05_v.txt
done
