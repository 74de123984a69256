#!/bin/bash
# round 4: host or device?  the sharded engine's enqueue loop at world size 1 under a profiler
O=$GRAFT_REPO_ROOT/gpurun_out/r04_z3; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 1 0; do echo "PSFM_SHARD_LAZY_CHECK=$m" >> $O/host.txt; PSFM_SHARD_LAZY_CHECK=$m timeout 300 python scripts/probe_sharded_host.py 401 >> $O/host.txt 2>&1; done
cat $O/host.txt | cut -c1-180
