#!/bin/bash
# round 5, call e: the whole bench line with the new extras (how long does it take?)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/r05_e_bench.json 2> gpurun_out/r05_e_bench.err
echo "bench rc=$? in $SECONDS s"; tail -3 gpurun_out/r05_e_bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_e_bench.json').read().strip().splitlines()[-1])
print('value %.3e ms %.3f' % (b['value'], b['ms_per_step']))
for k in ('secondary','secondary_1080p','secondary_hard','secondary_realistic','secondary_davis','secondary_scannet','secondary_davis_batch','secondary_batch','secondary_scannet_batch','single_sequence','single_sequence_hard','end_to_end','end_to_end_batch','concurrent'):
    v=b.get(k)
    if not isinstance(v,dict): print(k, v); continue
    if 'error' in v: print(k,'ERROR',v['error']); continue
    print(k, {q: v[q] for q in ('ms_per_sequence','ms_per_batch','speedup_vs_one_psfm_connect_per_sequence','modes','rejected_steps','iterations_per_solve','occluded_fraction','solver_counters','one_gpu_psfm_connect_ms_per_sequence','ratio_to_one_gpu_call','total_s','runs','parity','frame_launch') if q in v})
print('cpu_baseline', {k:v for k,v in b['cpu_baseline'].items() if k in ('value','cores','port_8_threads','reference_python_this_box')})
PY
