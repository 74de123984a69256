#!/bin/bash
# round 6, call n: soak of the randomised stress scripts on fresh seeds (the cross-rank solve runs inside stress_sharded's 2 / 3 thread-rank cases)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
SECONDS=0; timeout 900 python scripts/stress_sharded.py 150 611 2>&1 | tail -4; echo "stress_sharded: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_batch.py 80 612 2>&1 | tail -3; echo "stress_batch: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_optimize.py 80 613 2>&1 | tail -2; echo "stress_optimize: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_threads.py 12 614 2>&1 | tail -3; echo "stress_threads: $SECONDS s"
SECONDS=0; timeout 400 python scripts/stress_persist.py 150 615 2>&1 | tail -2; echo "stress_persist: $SECONDS s"
SECONDS=0; timeout 400 python scripts/stress_ingest.py 200 616 2>&1 | tail -2; echo "stress_ingest: $SECONDS s"
} | tee gpurun_out/r06_n_soak.txt
