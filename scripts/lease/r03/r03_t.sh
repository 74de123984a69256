#!/bin/bash
# round 3: what the launch chain's iteration loop gains from requesting loads earlier (PC_ITER_PIPE 0..3) and from more
# resident waves (PC_PERSIST_WAVES 4, larger grids): the hard 1080p sequence under each build of the library.
O=$GRAFT_REPO_ROOT/gpurun_out/r03_t; mkdir -p $O
V=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants
export PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive
run() { tag=$1; shift; env "$@" timeout 200 python scripts/probe_solver.py > $O/$tag.json 2> $O/$tag.err; python -c "
import json,sys; d=json.load(open('$O/$tag.json'))['adaptive']; print('$tag', round(d['ms_per_sequence'],2), round(d['solver_ms_per_seq'],2), d['iters'], d['counters'])"; }
run p0w3 PSFM_HIP_LIB=$V/libpsfm_hip_p0w3.so
run p1w3 PSFM_HIP_LIB=$V/libpsfm_hip_p1w3.so
run p2w3 PSFM_HIP_LIB=$V/libpsfm_hip_p2w3.so
run p1w4_512 PSFM_HIP_LIB=$V/libpsfm_hip_p1w4.so
run p1w4_768 PSFM_HIP_LIB=$V/libpsfm_hip_p1w4.so PSFM_PC_BLOCKS=768 PSFM_PC_PERSIST_FULL=1
run p1w4_1024 PSFM_HIP_LIB=$V/libpsfm_hip_p1w4.so PSFM_PC_BLOCKS=1024 PSFM_PC_PERSIST_FULL=1
run p1w3_launches PSFM_HIP_LIB=$V/libpsfm_hip_p1w3.so PSFM_PC_PERSIST=0
