#!/bin/bash
# round 6, call j: the sharded tests again, the validator test, the ingest probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_validate_flow_dir.py -m gpu -q --durations=6 2>&1 | tail -30
echo "sharded + validator: $SECONDS s"
bash scripts/lease/r06/i.sh
