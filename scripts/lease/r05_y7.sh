#!/bin/bash
# round 5, call y7: the batch stress, seeds 5-10
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 5 6 7 8 9 10; do
  timeout 900 python scripts/stress_batch.py 150 $seed 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/r05_y7_stress_batch.txt
done
