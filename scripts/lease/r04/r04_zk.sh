#!/bin/bash
# round 4: what the sharded engine's launches cost at world size 1 (kernel trace of scripts/probe_single_sequence.py 201): per-kernel stats
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zk; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/scripts/probe_single_sequence.py 201 > $O/run.log 2>&1
python - <<PY
import csv, glob
fn = glob.glob("$O/trace/**/t_kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn)) if "psfm_" in r["Name"] or "rocprim" in r["Name"]]
with open("$O/kernel_stats.txt", "w") as f:
    for r in rows[:14]:
        line = "%-60s calls %6s avg %9.1f ns total %8.2f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6)
        print(line); f.write(line + "\n")
PY
tail -1 $O/run.log | cut -c1-200
rm -rf $O/trace
