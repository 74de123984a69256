#!/bin/bash
# round 5, call zn0: the lane-table-full fix: its test and the stress seed that found it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_solver.py -m gpu -x -q -W ignore -k "lane_table_full or outlasts" 2>&1 | tail -15
timeout 300 python scripts/stress_batch.py 150 54 2>&1 | tail -2 | cut -c1-400
