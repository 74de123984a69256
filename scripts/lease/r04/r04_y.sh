#!/bin/bash
# round 4, final binary: the shape table of the chain policy (per-frame launches vs the persistent loop), whether parallel pwrite()s to
# tmpfs scale on the box (track.npy writer), and the bench line once more -- now replaying PMC files measured on these very sources
O=$GRAFT_REPO_ROOT/gpurun_out/r04_y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/probe_shapes.py > $O/probe_shapes.txt 2>&1; tail -30 $O/probe_shapes.txt
timeout 120 python scripts/micro/pwrite_scale.py > $O/pwrite_scale.txt 2>&1; nproc >> $O/pwrite_scale.txt; cat $O/pwrite_scale.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
