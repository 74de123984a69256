#!/bin/bash
# round 5, call j: the whole -m gpu suite on the sources of the moment (incl. the slow whole-sequence cases with their report)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/r05_j_whole_sequence_parity.jsonl
SECONDS=0
PSFM_WHOLE_SEQ_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_j_whole_sequence_parity.jsonl timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_j_tests.log 2>&1
echo "gpu suite rc=$? in $SECONDS s" >> gpurun_out/r05_j_tests.log; tail -6 gpurun_out/r05_j_tests.log
