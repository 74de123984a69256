#!/bin/bash
# round 6, last call: the evidence for profiles/ on the final sources (z.sh), then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/lease/r06/z.sh
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r06_zz_suite.txt
