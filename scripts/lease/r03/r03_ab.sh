#!/bin/bash
# round 3: psfm_frame_kernel / psfm_pc_fused_kernel at 4 waves per SIMD without spills (the kernels state the range of K)
O=$GRAFT_REPO_ROOT/gpurun_out/r03_ab; mkdir -p $O
(timeout 60 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; tail -2 $O/t1.log | head -1
timeout 60 python scripts/probe_single_sequence.py 201 2>/dev/null | cut -c1-200
(timeout 100 python -m pytest tests/test_gpu_solver.py -x -q -m gpu) > $O/t2.log 2>&1; tail -2 $O/t2.log | head -1
