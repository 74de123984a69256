#!/bin/bash
# finalize: group plan instead of decode + scan, segment prefix on the device -- parity, then A/B on the headline step
O=gpurun_out/r06_u; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_plan.json 2> $O/bench_plan.err
PSFM_FIN_PLAN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu > $O/bench_scan.json 2> $O/bench_scan.err
python - <<'P'
import json
for f in ("plan","scan"):
    try:
        l=json.loads(open("gpurun_out/r06_u/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, l["ms_per_step"], l["kernels"]["finalize_avg_us"], l.get("parity"))
    except Exception as e: print(f, "ERR", e)
P
tail -3 $O/bench_plan.err
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/suite.txt; cat $O/suite.txt
PSFM_FIN_PLAN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 > $O/parity_scan.txt; cat $O/parity_scan.txt
