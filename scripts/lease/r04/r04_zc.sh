#!/bin/bash
# round 4: where do the fused loop's flow_check slices cost?  what-if builds (timing only, maps incomplete): 16 = no mandatory slices in
# front of the arrival, 32 = no slices in the barrier wait (all of them in front of the arrival), 48 = no slices at all behind the prologue
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zc; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
P=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/libpsfm_hip.so
for lib in $P $V/libpsfm_hip_w16.so $V/libpsfm_hip_w32.so $V/libpsfm_hip_w48.so $P $V/libpsfm_hip_w16.so; do
  PSFM_HIP_LIB=$lib timeout 300 python scripts/probe_persist_variant.py 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timing_1080p']
print(d['lib'], 'exact' if d['ok'] else 'NOT EXACT', 'track us/step %.2f' % t['track_chain_us_per_step'], 'connect us/step %.2f' % t['connect_us_per_step'], 'connect ms %.3f' % t['connect_ms'], 'modes', t['track_mode'], t['connect_mode'])" | tee -a $O/ab.txt
done
