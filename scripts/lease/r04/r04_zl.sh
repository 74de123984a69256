#!/bin/bash
# round 4: the sharded engine's stall flag written to pinned memory by the control kernel itself (product) vs a 4-byte copy behind every
# control launch (variant memcpy): sharded tests + ONE sequence of configs[3] at world size 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zl; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
P=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/libpsfm_hip.so
for lib in $P $V/libpsfm_hip_memcpy.so $P $V/libpsfm_hip_memcpy.so; do
  PSFM_HIP_LIB=$lib timeout 300 python scripts/probe_single_sequence.py 401 2> /dev/null | tail -1 | cut -c1-140 | sed "s/^/$(basename $lib) /" | tee -a $O/ab.txt
done
