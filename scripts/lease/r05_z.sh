#!/bin/bash
# round 5, call z: the whole -m gpu suite on the final sources, then the profile set (scripts/profile_round5.sh r05_z) and the bench line with its files in place
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/r05_z_whole_sequence_parity.jsonl
SECONDS=0
PSFM_WHOLE_SEQ_REPORT=$GRAFT_REPO_ROOT/gpurun_out/r05_z_whole_sequence_parity.jsonl timeout 1500 python -m pytest tests -m gpu -x -q -W ignore > gpurun_out/r05_z_tests.log 2>&1
echo "gpu suite rc=$? in $SECONDS s" >> gpurun_out/r05_z_tests.log; tail -6 gpurun_out/r05_z_tests.log
SECONDS=0
bash scripts/profile_round5.sh r05_z > gpurun_out/r05_z_profile.log 2>&1
echo "profile_round5 rc=$? in $SECONDS s"; tail -3 gpurun_out/r05_z_profile.log
