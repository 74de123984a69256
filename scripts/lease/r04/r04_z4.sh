#!/bin/bash
# round 4: the sharded engine at world size 1 -- frames between two host synchronisations (a stall is seen early through pinned memory anyway)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_z4; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 16 64 400 16 64 400; do
  PSFM_SHARD_LAZY_CHECK=0 PSFM_SHARD_CHECK_EVERY=$m timeout 300 python scripts/probe_single_sequence.py 401 2> /dev/null | tail -1 | sed "s/^/check_every=$m /" | tee -a $O/ab.txt
done
