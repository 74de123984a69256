#!/bin/bash
# round 3: the software-pipelined iteration of the launch chain / persistent solve (state of track j+1 and its taps at x
# requested under track j's arithmetic): parity tests, then the hard 1080p sequence -- persistent solve at 3 waves per SIMD
# (168 VGPRs, 8 spilled), the 2-wave build (184 VGPRs, grid = residency), and the per-iteration launches.
O=$GRAFT_REPO_ROOT/gpurun_out/r03_s; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu) > $O/t1.log 2>&1; tail -4 $O/t1.log | head -2
export PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive
timeout 200 python scripts/probe_solver.py > $O/hard_w3.json 2> $O/hard_w3.err; cut -c1-400 $O/hard_w3.json
PSFM_HIP_LIB=$GRAFT_REPO_ROOT/particle-sfm_amd/lib/variants/libpsfm_hip_pw2.so timeout 200 python scripts/probe_solver.py > $O/hard_w2.json 2> $O/hard_w2.err; cut -c1-400 $O/hard_w2.json
PSFM_PC_PERSIST=0 timeout 200 python scripts/probe_solver.py > $O/hard_launches.json 2> $O/hard_launches.err; cut -c1-400 $O/hard_launches.json
(time timeout 600 python -m pytest tests/test_gpu_whole_sequence.py -x -q -m gpu -k "hard") > $O/t2.log 2>&1; tail -4 $O/t2.log | head -2
