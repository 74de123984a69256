#!/bin/bash
# round 5, call zf: where a round of the exchange form goes at world size 1: host time enqueuing vs wall
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PSFM_SHARD_LOCAL=0 PSFM_SHARD_TRACE=1 PSFM_PROBE_GC=freeze timeout 300 python scripts/probe_single_sequence.py 21 hard 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/r05_zf_rounds.txt
