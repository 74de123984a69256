#!/bin/bash
# round 5, call x2: the one-rank engine with Stage A in chunks on a side stream (LazyMaps) now that its frames are one launch each
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in 0 8 16 32 0 8 16; do
  PSFM_SHARD_STAGE_A_CHUNK=$c timeout 300 python scripts/probe_single_sequence.py 401 2>&1 | tail -1 | cut -c1-330 | sed "s/^/chunk=$c /" | tee -a gpurun_out/r05_x2_stage_a.txt
done
PSFM_SHARD_STAGE_A_CHUNK=8 timeout 300 python scripts/probe_single_sequence.py 101 hard 2>&1 | tail -1 | cut -c1-330 | sed "s/^/hard chunk=8 /" | tee -a gpurun_out/r05_x2_stage_a.txt
