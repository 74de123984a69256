#!/bin/bash
# finalize A/B: own sort (PSFM_FIN_SORT) x group plan (PSFM_FIN_PLAN): kernel stats of the headline step, then parity
O=gpurun_out/r06_v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in 11; do
PSFM_FIN_SORT=${cfg:0:1} PSFM_FIN_PLAN=${cfg:1:1} rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof$cfg -- python $R/bench.py --steps 10 --warmup 2 --no-extras > $R/$O/bench$cfg.json 2> $R/$O/bench$cfg.err
done
cd $R
python - <<'P'
import csv, glob, json
for cfg in ("11",):
    f = glob.glob("gpurun_out/r06_v/prof%s/*/*kernel_stats.csv" % cfg)[0]
    rows = [r for r in csv.DictReader(open(f)) if ("psfm" in r["Name"] or "rocprim" in r["Name"] or "fillBuffer" in r["Name"])]
    out = open("gpurun_out/r06_v/kernels_%s.txt" % cfg, "w")
    for r in rows:
        line = "%-70s calls %5s avg %10.1f ns" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]))
        print(line); out.write(line + "\n")
    t = glob.glob("gpurun_out/r06_v/prof%s/*/*kernel_trace.csv" % cfg)[0]
    per = {}
    for r in csv.DictReader(open(t)):
        nm = r["Kernel_Name"]
        if "psfm_sort" in nm:
            per.setdefault(nm[:24], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for nm, v in per.items():
        print(nm, "by pass:", [round(sum(v[p::4]) / len(v[p::4]) / 1e3, 1) for p in range(4)])
    try:
        l = json.loads(open("gpurun_out/r06_v/bench%s.json" % cfg).read().strip().splitlines()[-1])
        print(cfg, "ms/step", l["ms_per_step"], "finalize us", l["kernels"]["finalize_avg_us"], l.get("parity"))
    except Exception as e:
        print("ERR", e, open("gpurun_out/r06_v/bench%s.err" % cfg).read()[-1500:])
    print()
P
rm -rf $O/prof11 $O/prof01
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sort.py -q -m gpu -x 2>&1 | tail -3
