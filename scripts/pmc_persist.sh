#!/bin/bash
# PMC passes over the persistent loop on ready maps (psfm_track, chain mode 2, 1080p x 101): instruction mix and wait cycles.
# usage (on the GPU box): bash scripts/pmc_persist.sh <tag> [track|connect]
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=$1; WHAT=${2:-track}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/skew
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/p$i -o p -- python $R/scripts/run_track_once.py $WHAT 4 > $OUT/p$i.log 2>&1 < /dev/null
  tail -2 $OUT/p$i.log
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "persist_kernel" not in k or "init" in k: continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]].append(float(row["Counter_Value"]))
out = {}
for c, d in acc.items():
    vals = [sum(v) for v in d.values()]
    out[c] = sum(vals) / len(vals)
w = out.get("SQ_WAVES", 8192.0); fr = 100.0
print(json.dumps(out, indent=1))
for c in sorted(out):
    print("%-24s per wave and frame: %10.1f" % (c, out[c] / w / fr))
json.dump(out, open("$R/gpurun_out/skew/pmc_${TAG}_${WHAT}.json", "w"), indent=1)
PY
