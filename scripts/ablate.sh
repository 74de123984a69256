#!/bin/bash
# perf ablation of the fused chain kernel (results are WRONG with PSFM_DEBUG_SKIP != 0; timing only)
for d in 0 1 2 3 4 7; do
  echo "== PSFM_DEBUG_SKIP=$d"; PSFM_DEBUG_SKIP=$d python bench.py --steps 3 --warmup 1 --no-cpu --frames 41 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.readline()); print('ms/step %.3f chain_us %.2f fc_us %.1f fin_us %.1f'%(j['ms_per_step'], j['roofline']['avg_launch_us'], j['kernels']['flow_check']['avg_launch_us'], j['kernels']['finalize_avg_us']))"
done
