#!/bin/bash
O=gpurun_out/r03_h; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; grep -E "passed|failed" $O/t1.log
for p in 1 0; do PSFM_PC_PERSIST=$p PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_persist$p.json 2>$O/hard_persist$p.err; cat $O/hard_persist$p.json; done
PSFM_PROBE_MODES=adaptive,chain timeout 300 python scripts/probe_solver.py > $O/easy.json 2>$O/easy.err; cat $O/easy.json
