#!/bin/bash
# round 6, call zq: long soak on fresh seeds on the final sources (own record sort, group plan)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
SECONDS=0; timeout 900 python scripts/stress_sharded.py 1500 661 2>&1 | tail -2; echo "stress_sharded: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_batch.py 300 662 2>&1 | tail -2; echo "stress_batch: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_optimize.py 600 663 2>&1 | tail -1; echo "stress_optimize: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_persist.py 1500 664 2>&1 | tail -1; echo "stress_persist: $SECONDS s"
SECONDS=0; timeout 900 python scripts/stress_threads.py 60 665 2>&1 | tail -1; echo "stress_threads: $SECONDS s"
SECONDS=0; PSFM_STRESS_BIG=1 timeout 900 python scripts/stress_sharded.py 120 666 2>&1 | tail -1; echo "stress_sharded big: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_consumers.py 300 667 2>&1 | tail -1; echo "stress_consumers: $SECONDS s"
} | tee gpurun_out/r06_zq_soak_long.txt
