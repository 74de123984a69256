#!/bin/bash
# round 5, call y3: the batch stress with two more seeds
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for seed in 2 3; do
  timeout 900 python scripts/stress_batch.py 120 $seed 2>&1 | tail -2 | tee -a gpurun_out/r05_y3_stress_batch.txt
done
