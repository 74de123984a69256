#!/bin/bash
# round 5, call zn: the lane-table-full fix: (its test and the stress seed that found it: call zn0) the whole -m gpu suite and the profile set on the final sources
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/r05_zn_whole_sequence_parity.jsonl
SECONDS=0
PSFM_WHOLE_SEQ_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_zn_whole_sequence_parity.jsonl timeout 1200 python -m pytest tests -m gpu -x -q -W ignore > gpurun_out/r05_zn_tests.log 2>&1
echo "gpu suite rc=$? in $SECONDS s" >> gpurun_out/r05_zn_tests.log; tail -4 gpurun_out/r05_zn_tests.log
SECONDS=0
bash scripts/profile_round5.sh r05_zn > gpurun_out/r05_zn_profile.log 2>&1
echo "profile_round5 rc=$? in $SECONDS s"; tail -2 gpurun_out/r05_zn_profile.log
