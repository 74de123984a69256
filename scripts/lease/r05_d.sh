#!/bin/bash
# round 5, call d: batch + realistic tests, smoke, the whole bench line with the new extras
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_solver.py tests/test_gpu_parity.py -x -q -k "batch or realistic or redo_of or track_golden" > gpurun_out/r05_d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_d_tests.log
tail -6 gpurun_out/r05_d_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05_d_smoke.log 2>&1; tail -2 gpurun_out/r05_d_smoke.log
/usr/bin/time -v timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/r05_d_bench.json 2> gpurun_out/r05_d_bench.err
echo "bench rc=$?"; grep -E "Elapsed|Maximum resident" gpurun_out/r05_d_bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05_d_bench.json').read().strip().splitlines()[-1])
print('value %.3e ms %.3f' % (b['value'], b['ms_per_step']))
for k in ('secondary','secondary_1080p','secondary_hard','secondary_realistic','secondary_davis','secondary_scannet','secondary_davis_batch','secondary_batch','secondary_scannet_batch','single_sequence','single_sequence_hard','end_to_end','end_to_end_batch','concurrent'):
    v=b.get(k)
    if not isinstance(v,dict): print(k, v); continue
    if 'error' in v: print(k,'ERROR',v['error']); continue
    print(k, {q: v[q] for q in ('ms_per_sequence','ms_per_batch','speedup_vs_one_psfm_connect_per_sequence','modes','rejected_steps','iterations_per_solve','occluded_fraction','solver_counters','one_gpu_psfm_connect_ms_per_sequence','ratio_to_one_gpu_call','total_s','runs','parity','frame_launch') if q in v})
print('cpu_baseline', {k:v for k,v in b['cpu_baseline'].items() if k in ('value','cores','port_8_threads','reference_python_this_box')})
PY
