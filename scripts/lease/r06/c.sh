#!/bin/bash
# round 6, call c: the cross-rank resident solve (thread ranks, give-up, two processes over IPC) + the tests that failed / are new
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x -k "cross_rank or grows_tables" -s 2>&1 | tail -40 > gpurun_out/r06_c_peer.txt
echo "peer tests: $SECONDS s"; tail -40 gpurun_out/r06_c_peer.txt
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_sharded.py 2>&1 | tail -8
echo "rest of the suite (without test_gpu_sharded.py): $SECONDS s"
