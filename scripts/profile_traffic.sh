#!/bin/bash
# HBM traffic of the chain kernels from the PMC counters (separate --pmc passes), for profiles/traffic_chain_*.json:
#   two-call step (stand-alone flow_check, whose read volume calibrates FETCH_SIZE, + the persistent loop on its maps)
#   and the default fused step.  Usage (GPU box): bash scripts/profile_traffic.sh r02_t
TAG=${1:-r02_t}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
PSFM_BENCH_TWO_CALLS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/two_fetch -o f -- $B > $OUT/two_fetch.log 2>&1 < /dev/null
PSFM_BENCH_TWO_CALLS=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/two_write -o w -- $B > $OUT/two_write.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fused_fetch -o f -- $B > $OUT/fused_fetch.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/fused_write -o w -- $B > $OUT/fused_write.log 2>&1 < /dev/null
PSFM_CHAIN_MODE_BENCH=1 true
python - <<PY
import csv, json, os, collections, shutil
csv.field_size_limit(1 << 30)
out = "$OUT"
def avg(sub, pre):
    fn = os.path.join(out, sub, pre + "_counter_collection.csv")
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "psfm_" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        per[(r["Dispatch_Id"], name, r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (_, k, c), v in per.items():
        acc[(k, c)].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
H, W, NF = 1080, 1920, 100
tf, tw = avg("two_fetch", "f"), avg("two_write", "w")
ff, fw = avg("fused_fetch", "f"), avg("fused_write", "w")
fc = [k for k in tf if "flow_check" in k[0]][0]
cal = (16.0 * H * W * NF / 1024.0) / tf[fc]
res = {"flow_check_kernel": fc[0], "fetch_calibration": cal,
       "calibration_note": "FETCH_SIZE (KB) of the stand-alone flow_check launch against its exactly known read volume 16*H*W*100 bytes "
                           "(MI355X_MICROARCH.md: the counter is uncalibrated on gfx950 and depends on the access width); WRITE_SIZE used as is",
       "two_calls": {k[0] + ":" + k[1]: v for k, v in list(tf.items()) + list(tw.items())},
       "fused": {k[0] + ":" + k[1]: v for k, v in list(ff.items()) + list(fw.items())}}
pk = [k for k in tf if "chain_persist" in k[0]][0][0]
res["traffic_chain_persist"] = {"kernel": pk, "FETCH_SIZE_KB_per_launch": tf[(pk, "FETCH_SIZE")], "WRITE_SIZE_KB_per_launch": tw[(pk, "WRITE_SIZE")],
                                "fetch_calibration": cal, "hbm_bytes_per_launch": (tf[(pk, "FETCH_SIZE")] * cal + tw[(pk, "WRITE_SIZE")]) * 1024.0}
res["traffic_chain_fused"] = {"kernel": pk + " (flow_check fused in)", "FETCH_SIZE_KB_per_launch": ff[(pk, "FETCH_SIZE")],
                              "WRITE_SIZE_KB_per_launch": fw[(pk, "WRITE_SIZE")], "fetch_calibration": cal,
                              "hbm_bytes_per_launch": (ff[(pk, "FETCH_SIZE")] * cal + fw[(pk, "WRITE_SIZE")]) * 1024.0}
json.dump(res, open(os.path.join(os.path.dirname(out), "$TAG" + "_traffic.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("fetch_calibration", "traffic_chain_persist", "traffic_chain_fused")}, indent=1))
shutil.rmtree(out, ignore_errors=True)
PY
