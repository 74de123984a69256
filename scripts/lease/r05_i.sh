#!/bin/bash
# round 5, call i: trimmed batch grids (+ detection), early leave of rejecting sequences: tests, then A/B probes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > gpurun_out/r05_i_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_i_tests.log; tail -5 gpurun_out/r05_i_tests.log
for tr in 1 0; do
  echo "== PSFM_BATCH_GRID_TRIM=$tr" >> gpurun_out/r05_i_probe_trim.txt
  PSFM_BATCH_GRID_TRIM=$tr timeout 600 python scripts/probe_batch.py "" davis sintel scannet 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_i_probe_trim.txt
done
cat gpurun_out/r05_i_probe_trim.txt
timeout 600 python scripts/probe_batch.py gpurun_out/r05_i_probe_batch_real.json sintel_real davis_real > gpurun_out/r05_i_probe_batch_real.txt 2>&1
tail -9 gpurun_out/r05_i_probe_batch_real.txt
