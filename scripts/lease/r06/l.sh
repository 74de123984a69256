#!/bin/bash
# round 6, call l: the cross-rank solve with its granule areas in fine-grained memory (default) and in plain hipMalloc memory
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -k "cross_rank" 2>&1 | tail -3
for c in 0 1; do PSFM_PEER_COARSE=$c timeout 300 python scripts/probe_peer_thread_ranks.py 2 101 hard 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('coarse=$c', {k: d[k] for k in ('peer_ms', 'exchange_ms', 'psfm_connect_ms', 'peer_over_psfm_connect', 'counters_peer')})"; done | tee gpurun_out/r06_l_peer_finegrained.txt
