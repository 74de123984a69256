#!/bin/bash
# round 5, call zh: the whole -m gpu suite on the final tree (Python-side changes since call z: connect_sharded's abort path, stress scripts)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q -W ignore > gpurun_out/r05_zh_tests.log 2>&1
echo "gpu suite rc=$? in $SECONDS s" >> gpurun_out/r05_zh_tests.log; tail -6 gpurun_out/r05_zh_tests.log
