#!/bin/bash
# round 5, call c: batch with flow_check on the side stream -- parity tests, throughput probe, A/B against up-front flow_check
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q > gpurun_out/r05_c_tests.log 2>&1
echo "batch tests rc=$?" >> gpurun_out/r05_c_tests.log
tail -8 gpurun_out/r05_c_tests.log
timeout 300 python -m pytest tests/test_gpu_solver.py -x -q -k "redo_of" > gpurun_out/r05_c_tests2.log 2>&1
echo "redo tests rc=$?" >> gpurun_out/r05_c_tests2.log
tail -3 gpurun_out/r05_c_tests2.log
timeout 900 python scripts/probe_batch.py gpurun_out/r05_c_probe_batch.json > gpurun_out/r05_c_probe_batch.txt 2>&1
tail -30 gpurun_out/r05_c_probe_batch.txt
for ch in 0 4 16; do
  echo "== PSFM_BATCH_FC_CHUNK=$ch" >> gpurun_out/r05_c_probe_ab.txt
  PSFM_BATCH_FC_CHUNK=$ch timeout 600 python scripts/probe_batch.py "" davis sintel 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c_probe_ab.txt
done
cat gpurun_out/r05_c_probe_ab.txt
