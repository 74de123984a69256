#!/bin/bash
# round 6, call q: long soak on fresh seeds (about 20 minutes)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
SECONDS=0; timeout 900 python scripts/stress_sharded.py 1500 641 2>&1 | tail -2; echo "stress_sharded: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_batch.py 300 642 2>&1 | tail -2; echo "stress_batch: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_optimize.py 600 643 2>&1 | tail -1; echo "stress_optimize: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_threads.py 60 644 2>&1 | tail -1; echo "stress_threads: $SECONDS s"
SECONDS=0; PSFM_STRESS_BIG=1 timeout 900 python scripts/stress_sharded.py 120 645 2>&1 | tail -1; echo "stress_sharded big: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_consumers.py 300 646 2>&1 | tail -1; echo "stress_consumers: $SECONDS s"
} | tee gpurun_out/r06_q_soak_long.txt
