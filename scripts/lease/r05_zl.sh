#!/bin/bash
# round 5, call zl: a soak of every stress script with fresh seeds on the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
o=gpurun_out/r05_zl_soak.txt
for seed in 51 52 53 54; do timeout 600 python scripts/stress_batch.py 150 $seed 2>&1 | tail -1 | cut -c1-300 >> $o; done
for seed in 51 52; do timeout 600 python scripts/stress_threads.py 100 $seed 2>&1 | tail -1 | cut -c1-300 >> $o; done
for seed in 51 52 53; do timeout 600 python scripts/stress_sharded.py 150 $seed 2>&1 | tail -1 | cut -c1-300 >> $o; done
for seed in 51 52; do timeout 900 python scripts/stress_optimize.py 300 $seed 2>&1 | grep "MISMATCH\|\"cases\"" | cut -c1-300 >> $o; done
timeout 600 python scripts/stress_consumers.py 300 51 2>&1 | tail -1 | cut -c1-300 >> $o
timeout 600 python scripts/stress_ingest.py 400 51 2>&1 | tail -1 | cut -c1-300 >> $o
timeout 600 python scripts/stress_persist.py 400 51 2>&1 | tail -1 | cut -c1-300 >> $o
cat $o
