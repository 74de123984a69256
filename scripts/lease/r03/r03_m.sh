#!/bin/bash
O=gpurun_out/r03_m; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; grep -E "passed|failed|Error" $O/t1.log | head -5
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1]); s=d['single_sequence']; print({k: s[k] for k in ('ms_per_sequence','one_gpu_psfm_connect_ms_per_sequence','counts_equal_one_gpu','solver_counters','trust_region_iterations') if k in s})"
