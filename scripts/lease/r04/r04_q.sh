#!/bin/bash
# round 4: XCD-banded tiles in the per-frame chain step / the frame kernels (psfm_xcd_tile): parity, timing A/B (PSFM_XCD_TILES=0 = block order), fabric reads
O=$GRAFT_REPO_ROOT/gpurun_out/r04_q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
for v in 1 0 1 0; do
  for shape in "1080 1920 101 2" "436 1024 50 2" "480 640 300 1"; do
    PSFM_XCD_TILES=$v PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py $shape 2> /dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=d['adaptive']; print('PSFM_XCD_TILES=$v', d['shape'], 'ms/seq %.3f' % a['ms_per_sequence'], 'solver launch us %.2f' % (1e3*a['solver_ms_per_seq']/max(a['solver_launches_per_seq'],1)))" | tee -a $O/ab.txt
  done
done
export TMPDIR=/tmp; cd /tmp
for v in 1 0; do
  PSFM_XCD_TILES=$v PSFM_PROBE_MODES=adaptive timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum --kernel-include-regex "psfm_seq" -f csv -d $O/rd$v -o r -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/rd$v.log 2>&1 < /dev/null
done
python - <<'P'
import csv, glob, os, collections
csv.field_size_limit(1 << 30)
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_q"
for sub in ("rd1", "rd0"):
    f = glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f: print(sub, "no output"); continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "psfm_seq" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_EA0_RDREQ_DRAM_32B_sum": per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(per.values()); v = [x for x in v if x > 0.25 * v[-1]]
    print(sub, "psfm_seq_kernel reads per launch with work: %.1f MB over %d launches" % (32.0 * sum(v) / len(v) / 1e6, len(v)))
P
rm -rf $O/rd1 $O/rd0
