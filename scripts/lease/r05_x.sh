#!/bin/bash
# round 5, call x: the profile set of the final sources (scripts/profile_round5.sh) -> gpurun_out/r05_x_summary/
cd "$GRAFT_REPO_ROOT"
SECONDS=0
bash scripts/profile_round5.sh r05_x > gpurun_out/r05_x_profile.log 2>&1
echo "profile_round5 rc=$? in $SECONDS s"; tail -25 gpurun_out/r05_x_profile.log
