#!/bin/bash
# round 4: the fused loop's flow_check in units of one pixel per lane, the next unit's flow prefetched (product) vs whole slices (fcu0) vs
# units without the prefetch (fcnp): exactness on every shape + us per step of the fused launch
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zd; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
P=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/libpsfm_hip.so
for lib in $P $V/libpsfm_hip_fcu0.so $V/libpsfm_hip_fcnp.so $P $V/libpsfm_hip_fcu0.so $V/libpsfm_hip_fcnp.so; do
  PSFM_HIP_LIB=$lib timeout 300 python scripts/probe_persist_variant.py 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing_1080p']
print(d['lib'], 'exact' if d['ok'] else 'NOT EXACT', 'track us/step %.2f' % t['track_chain_us_per_step'], 'connect us/step %.2f' % t['connect_us_per_step'], 'connect ms %.3f' % t['connect_ms'], 'modes', t['track_mode'], t['connect_mode'])" | tee -a $O/ab.txt
done
