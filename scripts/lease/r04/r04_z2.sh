#!/bin/bash
# round 4: ONE sequence through the sharded engine at world size 1 -- Stage A in chunks under the recurrence (PSFM_SHARD_LAZY_CHECK=1) vs up front
O=$GRAFT_REPO_ROOT/gpurun_out/r04_z2; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
for m in 1 0 1 0; do
  PSFM_SHARD_LAZY_CHECK=$m timeout 300 python scripts/probe_single_sequence.py 401 2> /dev/null | tail -1 | sed "s/^/lazy=$m /" | tee -a $O/ab.txt
done
