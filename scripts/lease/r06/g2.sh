#!/bin/bash
# round 6, call g2: idle time between the kernels of one headline step (kernel trace timestamps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_g2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu > $O/bench.json 2> $O/bench.err
python - $O <<'P' | tee $O/gaps.txt
import csv, glob, sys
o = sys.argv[1]
t = glob.glob(o + "/prof/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(t))]
rows.sort()
# steps: from one psfm_persist_init_kernel to the next
idx = [i for i, r in enumerate(rows) if "psfm_persist_init" in r[2]]
steps = [rows[a:b] for a, b in zip(idx, idx[1:])]
steps = [s for s in steps if any("gather_delta" in r[2] for r in s)][-8:]
import collections
acc = collections.OrderedDict()
for s in steps:
    s = [r for r in s if "psfm" in r[2] or "rocprim" in r[2] or "fillBuffer" in r[2]]
    for k, (a, b) in enumerate(zip(s, s[1:])):
        key = "%02d %s -> %s" % (k, a[2][:26], b[2][:26])
        acc.setdefault(key, []).append((b[0] - a[1]) / 1e3)
    acc.setdefault("step span (init start -> gather end)", []).append((s[-1][1] - s[0][0]) / 1e3)
    acc.setdefault("sum of kernels", []).append(sum(r[1] - r[0] for r in s) / 1e3)
for k, v in acc.items():
    print("%-70s %8.1f us" % (k, sum(v) / len(v)))
P
rm -rf $O/prof
