#!/bin/bash
# round 3: the sharded engine with chain step + fused export as ONE launch per frame (psfm_shard_frame)
O=$GRAFT_REPO_ROOT/gpurun_out/r03_aa; mkdir -p $O
(time timeout 150 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; tail -5 $O/t1.log | head -3
for m in 1 0; do PSFM_SHARD_MERGED=$m timeout 100 python scripts/probe_single_sequence.py 201 > $O/single_m$m.json 2> $O/single_m$m.err; cat $O/single_m$m.json; tail -2 $O/single_m$m.err | grep -v amdgpu.ids; done
