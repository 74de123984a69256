#!/bin/bash
# round 6, call o: soak at DAVIS / Sintel-sized frames (the cross-rank solve on launches of several hundred blocks), fresh seeds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
SECONDS=0; PSFM_STRESS_BIG=1 timeout 1200 python scripts/stress_sharded.py 40 631 2>&1 | tail -3; echo "stress_sharded big: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_optimize.py 12 632 big 2>&1 | tail -2; echo "stress_optimize big: $SECONDS s"
SECONDS=0; PSFM_STRESS_BIG=1 timeout 900 python scripts/stress_batch.py 10 633 2>&1 | tail -2; echo "stress_batch big: $SECONDS s"
} | tee gpurun_out/r06_o_soak_big.txt
