#!/bin/bash
# round 6, call m: the cross-rank solve with the ranks' rows polled by all four waves
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -k "cross_rank or hip_engine" 2>&1 | tail -3
for w in 2 4; do GPU_MAX_HW_QUEUES=16 timeout 400 python scripts/probe_peer_thread_ranks.py $w 101 hard 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('world=$w', {k: d[k] for k in ('peer_ms', 'exchange_ms', 'psfm_connect_ms', 'peer_over_psfm_connect', 'counters_peer')})"; done | tee gpurun_out/r06_m_peer_waves.txt
