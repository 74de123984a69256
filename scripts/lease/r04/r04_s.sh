#!/bin/bash
# round 4: gather locality of the launch chain on hard flows: lines per wave-load and cache hit rates of psfm_pc_iter_kernel
# (one launch per trust-region iteration: PSFM_PC_PERSIST=0 -- TA / TCP / TCC passes hang a kernel with a device-wide hand-off), lists banded / not
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s; mkdir -p $O
export TMPDIR=/tmp PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive PSFM_PC_PERSIST=0
cd /tmp
for v in 1 0; do
  PSFM_PC_BAND=$v timeout 300 rocprofv3 --kernel-trace --pmc TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-include-regex "psfm_pc_iter" -f csv -d $O/tcp$v -o t -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/tcp$v.log 2>&1 < /dev/null
  PSFM_PC_BAND=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "psfm_pc_iter" -f csv -d $O/tcc$v -o t -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/tcc$v.log 2>&1 < /dev/null
done
python - <<'P' | tee $O/locality.txt
import csv, glob, os, collections
csv.field_size_limit(1 << 30)
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_s"
for sub in ("tcp1", "tcc1", "tcp0", "tcc0"):
    f = glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not f: print(sub, "no output:", open(O + "/" + sub + ".log").read()[-300:]); continue
    agg = collections.defaultdict(float); n = set()
    for r in csv.DictReader(open(f[0])):
        if "psfm_pc_iter" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    d = {k: v / max(len(n), 1) for k, v in agg.items()}
    extra = ""
    if "TA_FLAT_READ_WAVEFRONTS_sum" in d and d["TA_FLAT_READ_WAVEFRONTS_sum"] > 0:
        extra = " | cache-line accesses per wave-load %.1f, L1 miss requests per wave-load %.1f, avg L1->L2 read latency %.0f cycles" % (
            d["TCP_TOTAL_CACHE_ACCESSES_sum"] / d["TA_FLAT_READ_WAVEFRONTS_sum"], d["TCP_TCC_READ_REQ_sum"] / d["TA_FLAT_READ_WAVEFRONTS_sum"],
            d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(d["TCP_TCC_READ_REQ_sum"], 1))
    if "TCC_REQ_sum" in d:
        extra = " | L2 hit rate %.3f" % (d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1))
    print(sub, "(PSFM_PC_BAND=%s)" % sub[-1], "launches", len(n), {k: round(v) for k, v in d.items()}, extra)
P
rm -rf $O/tcp1 $O/tcc1 $O/tcp0 $O/tcc0
