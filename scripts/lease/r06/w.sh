#!/bin/bash
# where do the fillBufferAligned launches come from: count at two step counts
O=gpurun_out/r06_w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in 4 14; do
rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof$st -- python $R/bench.py --steps $st --warmup 2 --no-cpu --no-extras > $R/$O/bench$st.json 2> $R/$O/bench$st.err
done
cd $R
python - <<'P'
import csv, glob
for st in (4, 14):
    f = glob.glob("gpurun_out/r06_w/prof%d/*/*kernel_stats.csv" % st)[0]
    for r in csv.DictReader(open(f)):
        if "fillBuffer" in r["Name"] or "persist_kernel" in r["Name"]:
            print(st, r["Name"][:50], r["Calls"], r["AverageNs"])
P
rm -rf $O/prof4 $O/prof14
