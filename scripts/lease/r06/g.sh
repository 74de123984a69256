#!/bin/bash
# round 6, call g: kernel stats of the two-thread-rank hard sequence (cross-rank resident solve), durations of the stress tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_g
export TMPDIR=/tmp
cat > /tmp/peer_once.py <<'P'
import sys, os, json
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "scripts"))
import probe_peer_thread_ranks as p
print(json.dumps(p.run(2, 101, "hard", forms=("peer",), reps=2)))
P
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_g/prof -- python /tmp/peer_once.py > $GRAFT_REPO_ROOT/gpurun_out/r06_g/peer_once.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_g/peer_once.err)
tail -1 gpurun_out/r06_g/peer_once.json | cut -c1-600
f=$(find gpurun_out/r06_g/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" gpurun_out/r06_g/peer_kernel_stats.csv
rm -rf gpurun_out/r06_g/prof
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_stress.py -m gpu -q --durations=12 2>&1 | tail -20
echo "stress tests: $SECONDS s"
