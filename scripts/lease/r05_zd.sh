#!/bin/bash
# round 5, call zd: the driver's own sequence at round end -- smoke(), then `python bench.py` with no flags -- timed
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "smoke: $SECONDS s"
SECONDS=0
timeout 900 python bench.py > gpurun_out/r05_zd_bench.json 2> gpurun_out/r05_zd_bench.err
echo "bench rc=$? in $SECONDS s"; python -c "
import json
b=json.loads(open('gpurun_out/r05_zd_bench.json').read().strip().splitlines()[-1])
print(b['metric'], b['value'], b['ms_per_step'], b['roofline']['frac'], b['cpu_baseline']['value'], sorted(k for k in b if isinstance(b[k],dict)))
print({k:(b[k].get('error') or b[k].get('ms_per_sequence')) for k in b if isinstance(b[k],dict) and ('ms_per_sequence' in b[k] or 'error' in b[k])})
"
tail -3 gpurun_out/r05_zd_bench.err
