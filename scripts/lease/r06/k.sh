#!/bin/bash
# round 6, call k: solver tests + the hard / realistic probes after the block-sum change
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_whole_sequence.py -m gpu -q -x -k "not configs3 and not configs4" 2>&1 | tail -4
for i in 1 2; do PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py 2>&1 | tail -1 | cut -c1-400; done
