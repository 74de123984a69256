#!/bin/bash
# round 4: exact HBM byte counters (TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B: 32-byte units, a 128-byte request counts 4)
# validated on the stand-alone flow_check launch (known read volume), then read off the chain kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r04_o; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
run() { d=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" > $O/$d.log 2>&1 < /dev/null; }
PSFM_BENCH_TWO_CALLS=1 run two_rd --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum -f csv -d $O/two_rd -o r -- $B2
PSFM_BENCH_TWO_CALLS=1 run two_wr --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum -f csv -d $O/two_wr -o w -- $B2
run fused_rd --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum -f csv -d $O/fused_rd -o r -- $B2
run fused_wr --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum -f csv -d $O/fused_wr -o w -- $B2
python - <<'P'
import csv, glob, json, os, collections
csv.field_size_limit(1 << 30)
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_o"
out = {}
for sub in ("two_rd", "two_wr", "fused_rd", "fused_wr"):
    f = glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f: out[sub] = open(O + "/" + sub + ".log").read()[-400:]; continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "psfm_" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").strip(), r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for (_, k, c), v in per.items(): acc[k][c].append(v)
    out[sub] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items() if "persist_kernel" in k or "flow_check" in k or "gather" in k}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
P
for t in two_rd two_wr fused_rd fused_wr; do rm -rf $O/$t; done
