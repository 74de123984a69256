#!/bin/bash
# round 3, run c: XCD-aware chunk mapping of flow_check (stand-alone, background and fused-in-the-loop forms): parity, A/B timing
O=gpurun_out/r03_c; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solver.py -x -q -m gpu) > $O/t1.log 2>&1; tail -4 $O/t1.log
for x in 0 1; do
  PSFM_FC_XCD=$x timeout 200 python scripts/probe_flow_check.py > $O/fc_xcd$x.json 2> $O/fc_xcd$x.err; cat $O/fc_xcd$x.json
  PSFM_FC_XCD=$x timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $O/bench_xcd$x.json 2> $O/bench_xcd$x.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_xcd$x.json').read().strip().split('\n')[-1]); print('xcd$x', 'ms_per_step', d['ms_per_step'], 'loop_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  PSFM_FC_XCD=$x PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/easy_xcd$x.json 2>$O/easy_xcd$x.err; cat $O/easy_xcd$x.json
done
