#!/bin/bash
# round 5, call zp: a last mini-soak on the final sources (7a0c03b8493de096)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
o=gpurun_out/r05_zp_soak.txt
timeout 200 python scripts/stress_threads.py 60 71 2>&1 | tail -1 | cut -c1-300 >> $o
timeout 200 python scripts/stress_sharded.py 150 71 2>&1 | tail -1 | cut -c1-300 >> $o
timeout 200 python scripts/stress_optimize.py 200 71 2>&1 | grep "MISMATCH\|\"cases\"" | cut -c1-300 >> $o
cat $o
