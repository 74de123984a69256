#!/bin/bash
# round 4: PMC (SQ counters only -- TA / TCC passes hang kernels with a device-wide hand-off) on the resident solve, hard 1080p sequence
O=$GRAFT_REPO_ROOT/gpurun_out/r04_g; mkdir -p $O
export TMPDIR=/tmp PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive
cd /tmp
pass() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "psfm_pc_resident" -f csv -d $O/$tag -o s -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/$tag.log 2>&1 < /dev/null; }
pass sq SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
pass sq2 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
python - <<'P' > $O/summary.json 2> $O/summary.err
import csv, glob, json, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_g"
out = {}
for tag in ("sq", "sq2"):
    f = glob.glob(O + "/" + tag + "/**/*counter_collection.csv", recursive=True)
    if not f: out[tag] = "no output"; continue
    agg = collections.defaultdict(float); calls = set()
    for row in csv.DictReader(open(f[0])):
        if "psfm_pc_resident" not in row["Kernel_Name"]: continue
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); calls.add(row["Dispatch_Id"])
    out[tag] = {"dispatches": len(calls), **{k: v / max(len(calls), 1) for k, v in agg.items()}}
print(json.dumps(out, indent=1))
P
cat $O/summary.json; tail -3 $O/summary.err
for t in sq sq2; do rm -rf $O/$t; done
