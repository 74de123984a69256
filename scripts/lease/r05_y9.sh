#!/bin/bash
# round 5, call y9: randomised stress of connect_sharded (one rank: local forms; 2 / 3 thread-ranks: exchange form) against psfm_connect
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 1 2; do
  timeout 1200 python scripts/stress_sharded.py 120 $seed > gpurun_out/r05_y9_tmp.txt 2>&1; echo "rc=$?" >> gpurun_out/r05_y9_tmp.txt
  grep -v amdgpu.ids gpurun_out/r05_y9_tmp.txt | tail -4 | cut -c1-420 | tee -a gpurun_out/r05_y9_stress_sharded.txt
done
