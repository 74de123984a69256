#!/bin/bash
# round 5, call u: the committed tree (before the local solver forms) beside the working tree on ONE box: is the intermittent +13 ms of the
# clean 401-frame single sequence older than the change?
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2 3 4; do
  (cd _old && timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | cut -c1-100 | sed "s/^/old /") | tee -a gpurun_out/r05_u.txt
  PSFM_SHARD_LOCAL=0 timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | cut -c1-100 | sed "s/^/new-local0 /" | tee -a gpurun_out/r05_u.txt
  timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | cut -c1-100 | sed "s/^/new /" | tee -a gpurun_out/r05_u.txt
done
