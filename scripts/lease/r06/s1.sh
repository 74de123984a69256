#!/bin/bash
# the slot gather: parity on the small shapes first (short timeouts: a kernel that never ends must not cost the lease), then the headline step
O=gpurun_out/r06_s1; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -4 | tee $O/parity.txt
grep -q "passed" $O/parity.txt || exit 1
grep -q "failed\|error" $O/parity.txt && exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for slots in 1 0; do
PSFM_FIN_SLOTS=$slots timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof$slots -- python $R/bench.py --steps 10 --warmup 2 --no-extras > $R/$O/bench$slots.json 2> $R/$O/bench$slots.err
done
cd $R
python - <<'P' | tee gpurun_out/r06_s1/kernels.txt
import csv, glob, json
for slots in (1, 0):
    f = glob.glob("gpurun_out/r06_s1/prof%d/*/*kernel_stats.csv" % slots)[0]
    for r in csv.DictReader(open(f)):
        if ("psfm" in r["Name"]) and not "sort" in r["Name"]:
            print("slots %d  %-60s calls %5s avg %10.1f ns" % (slots, r["Name"][:60], r["Calls"], float(r["AverageNs"])))
    try:
        l = json.loads(open("gpurun_out/r06_s1/bench%d.json" % slots).read().strip().splitlines()[-1])
        print("slots", slots, "ms/step", l["ms_per_step"], "finalize us", l["kernels"]["finalize_avg_us"], l.get("parity"))
    except Exception as e:
        print("ERR", e, open("gpurun_out/r06_s1/bench%d.err" % slots).read()[-1500:])
P
rm -rf $O/prof1 $O/prof0
