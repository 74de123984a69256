#!/bin/bash
# round 6, call r: the lane-tiled finalize of the persistent loop -- parity, then the headline with it and without it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python scripts/stress_persist.py 200 651 2>&1 | tail -1
for v in 1 0; do
  PSFM_FIN_LANES=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lanes=$v', d['value'], d['ms_per_step'], d['kernels']['finalize_avg_us'], d['parity'])"
done | tee gpurun_out/r06_r_finalize_lanes.txt
