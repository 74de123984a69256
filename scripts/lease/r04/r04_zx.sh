#!/bin/bash
# round 4, final sources: the profile set once more (sources with the leaner control launch of the sharded engine) with the frame kernel's fabric-side bytes on configs[2]'s and configs[4]'s shapes too
# (solver_valu.json: hbm_bytes_per_launch_by_shape), then the bench line that replays it
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zx_log; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash scripts/profile_round4.sh r04_zx > $O/profile.log 2>&1; tail -3 $O/profile.log
cp $GRAFT_REPO_ROOT/gpurun_out/r04_zx_summary/solver_valu.json $GRAFT_REPO_ROOT/gpurun_out/r04_zx_summary/traffic_chain_*.json $GRAFT_REPO_ROOT/profiles/
cd $GRAFT_REPO_ROOT; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
