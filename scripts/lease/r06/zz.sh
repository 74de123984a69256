#!/bin/bash
# round 6, last call: the whole GPU suite on the final sources, then a soak on fresh seeds through the new finalize (own sort, group plan)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r06_zz_suite.txt
{
SECONDS=0; timeout 600 python scripts/stress_persist.py 400 651 2>&1 | tail -1; echo "stress_persist: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_optimize.py 400 652 2>&1 | tail -1; echo "stress_optimize: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_batch.py 200 653 2>&1 | tail -2; echo "stress_batch: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_sharded.py 600 654 2>&1 | tail -2; echo "stress_sharded: $SECONDS s"
SECONDS=0; timeout 600 python scripts/stress_consumers.py 200 655 2>&1 | tail -1; echo "stress_consumers: $SECONDS s"
} | tee gpurun_out/r06_zz_soak.txt
