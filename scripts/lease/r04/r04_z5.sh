#!/bin/bash
# round 4: the sharded engine at world size 1 -- Stage A in chunks of 8 / 32 / 100 pairs under the recurrence vs up front, one checkpoint per 256 frames
O=$GRAFT_REPO_ROOT/gpurun_out/r04_z5; mkdir -p $O
cd $GRAFT_REPO_ROOT
export PSFM_SHARD_CHECK_EVERY=256
for m in "0 8" "1 8" "1 32" "1 100" "0 8" "1 32" "1 100"; do
  set -- $m
  PSFM_SHARD_LAZY_CHECK=$1 PSFM_SHARD_CHECK_CHUNK=$2 timeout 300 python scripts/probe_single_sequence.py 401 2> /dev/null | tail -1 | cut -c1-120 | sed "s/^/lazy=$1 chunk=$2 /" | tee -a $O/ab.txt
done
PSFM_SHARD_LAZY_CHECK=1 PSFM_SHARD_CHECK_CHUNK=32 timeout 300 python scripts/probe_sharded_host.py 401 2>&1 | head -14 | cut -c1-150 | tee -a $O/host.txt
