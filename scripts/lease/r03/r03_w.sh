#!/bin/bash
# round 3: what bounds the iteration loop of the persistent solve?  PMC passes on the hard 1080p sequence, product kernels only.
O=$GRAFT_REPO_ROOT/gpurun_out/r03_w; mkdir -p $O
export TMPDIR=/tmp PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive
cd /tmp
pass() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "psfm_pc_persist" -f csv -d $O/$tag -o s -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/$tag.log 2>&1 < /dev/null; }
pass sq SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum
pass sq2 SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
python - <<'P' > $O/summary.json 2> $O/summary.err
import csv, glob, json, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_w"
out = {}
for tag in ("sq", "ta", "tcc", "sq2"):
    f = glob.glob(O + "/" + tag + "/**/*counter_collection.csv", recursive=True)
    if not f: out[tag] = "no output"; continue
    agg = collections.defaultdict(float); calls = set()
    for row in csv.DictReader(open(f[0])):
        if "psfm_pc_persist" not in row["Kernel_Name"]: continue
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); calls.add(row["Dispatch_Id"])
    out[tag] = {"dispatches": len(calls), **{k: v / max(len(calls), 1) for k, v in agg.items()}}
print(json.dumps(out, indent=1))
P
cat $O/summary.json; cat $O/summary.err | tail -3
for t in sq ta tcc sq2; do rm -rf $O/$t; grep -c "Segmentation\|Aborted" $O/$t.log; done
