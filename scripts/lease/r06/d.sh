#!/bin/bash
# round 6, call d: sharded tests (capacity retry, cross-rank resident solve) + the host write-path probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -q -k "cross_rank or grows_tables" -s 2>&1 | tail -60 > gpurun_out/r06_d_peer.txt
echo "peer tests: $SECONDS s"; tail -60 gpurun_out/r06_d_peer.txt
timeout 300 python scripts/micro/write_paths.py 0.9 2>&1 | tee gpurun_out/r06_d_write_paths.txt
