#!/bin/bash
# round 4: ONE sequence through the sharded engine at world size 1 -- an iteration of margin on the speculated K against the redone solves
O=$GRAFT_REPO_ROOT/gpurun_out/r04_z; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do
  PSFM_SHARD_K_MARGIN=$m timeout 300 python scripts/probe_single_sequence.py 401 2> /dev/null | tail -1 | sed "s/^/margin=$m /" | tee -a $O/ab.txt
done
