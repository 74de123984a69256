#!/bin/bash
# own record sort + group plan: the sort on its own, the headline step A/B (no profiler), then the whole GPU suite
O=gpurun_out/r06_x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sort.py -q -m gpu -x 2>&1 | tail -5 | tee $O/sort_tests.txt
for cfg in 11 01 00; do
PSFM_FIN_SORT=${cfg:0:1} PSFM_FIN_PLAN=${cfg:1:1} timeout 300 python bench.py --steps 20 --warmup 5 --no-extras $([ $cfg = 11 ] || echo --no-cpu) > $O/bench$cfg.json 2> $O/bench$cfg.err
done
python - <<'P' | tee gpurun_out/r06_x/ab.txt
import json
for cfg in ("11", "01", "00"):
    try:
        l = json.loads(open("gpurun_out/r06_x/bench%s.json" % cfg).read().strip().splitlines()[-1])
        print("own sort %s, group plan %s: ms/step %.4f  finalize %.1f us  value %.4g  parity %s" % (cfg[0], cfg[1], l["ms_per_step"], l["kernels"]["finalize_avg_us"], l["value"], l.get("parity")))
    except Exception as e:
        print(cfg, "ERR", e)
P
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/suite.txt
PSFM_FIN_SORT=0 PSFM_FIN_PLAN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3 | tee $O/parity_old_forms.txt
