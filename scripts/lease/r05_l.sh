#!/bin/bash
# round 5, call l: the native .flo stack loader (psfm_load_flo_stack): ingest tests, small / 1080p disk-to-disk against the Python pipeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -k "ingest or end_to_end or connect_sequences" > gpurun_out/r05_l_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_l_tests.log; tail -4 gpurun_out/r05_l_tests.log
for nat in 1 0; do
  echo "== PSFM_FLO_NATIVE=$nat" >> gpurun_out/r05_l_e2e.txt
  PSFM_FLO_NATIVE=$nat timeout 600 python scripts/probe_e2e_small.py 2>&1 | grep -v amdgpu >> gpurun_out/r05_l_e2e.txt
  PSFM_FLO_NATIVE=$nat timeout 600 python scripts/end_to_end.py 101 /dev/shm/psfm_e2e_l 2>&1 | grep -v amdgpu | cut -c1-330 >> gpurun_out/r05_l_e2e.txt
done
cat gpurun_out/r05_l_e2e.txt
