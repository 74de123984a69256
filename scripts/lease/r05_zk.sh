#!/bin/bash
# round 5, call zk: randomised stress of traj_to_matches on the device against the host tables
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python scripts/stress_consumers.py 150 1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 | tee gpurun_out/r05_zk_stress_consumers.txt
