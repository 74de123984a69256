#!/bin/bash
# round 5, call ze: the control step of the exchange form folded into the next launch (psfm_shard_solve_export kinds 3 / 4): sharded tests,
# stress_sharded, the hard single sequence through the exchange form with and without the fold
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -W ignore > gpurun_out/r05_ze_tests.log 2>&1
echo "sharded tests rc=$? in $SECONDS s" >> gpurun_out/r05_ze_tests.log; tail -12 gpurun_out/r05_ze_tests.log
timeout 900 python scripts/stress_sharded.py 120 3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 | tee gpurun_out/r05_ze_stress_sharded.txt
for f in 1 0 1 0; do
  PSFM_SHARD_FOLD_CONTROL=$f PSFM_PROBE_GC=freeze timeout 300 python scripts/probe_single_sequence.py 101 hard 2>&1 | tail -1 | cut -c1-420 | sed "s/^/fold=$f /" | tee -a gpurun_out/r05_ze_fold.txt
done
