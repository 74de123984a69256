#!/bin/bash
# round 4, final sources: random track_optimize sequences on the device against the oracle (small maps: 600 cases; big maps up to
# 500 k tracks per solve: 24 cases -- the resident solve with 1-3 tracks per thread and its streamed tail)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zf; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/stress_optimize.py 600 4 2>&1 | tail -3 | tee $O/stress_small.txt
timeout 900 python scripts/stress_optimize.py 24 5 big 2>&1 | tail -3 | tee $O/stress_big.txt
