#!/bin/bash
# round 4: what a frame of the persistent loop would cost without one of the links of its chain (timing experiments: wrong results)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_k; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in default whatif1 whatif2 whatif3 whatif4 whatif9; do
  if [ $v = default ]; then unset PSFM_HIP_LIB; else export PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_$v.so; fi
  timeout 200 python scripts/probe_persist_variant.py 2> $O/$v.err | python -c "import json,sys; d=json.load(sys.stdin); t=d['timing_1080p']; print('$v', 'exact' if d['ok'] else 'NOT exact', 'track us/step %.2f' % t['track_chain_us_per_step'], 'connect us/step %.2f' % t['connect_us_per_step'], 'modes', t['track_mode'], t['connect_mode'])" | tee -a $O/whatif.txt
done
