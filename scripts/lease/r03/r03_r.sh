#!/bin/bash
# round 3, final validation: the whole GPU suite, a PMC pass on the hard-flow path restricted to the product kernels
# (rocprofv3 crashed inside torch's elementwise kernels of the data generator when it instrumented everything), the
# sharded engine's fused export at 3 / 4 waves per SIMD, and the default bench line.
O=$GRAFT_REPO_ROOT/gpurun_out/r03_r; mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -q -m gpu -x) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
for w in 3 4; do PSFM_FUSED_WAVES=$w timeout 200 python scripts/probe_single_sequence.py 201 > $O/single_w$w.json 2> $O/single_w$w.err; cat $O/single_w$w.json; done
cd /tmp
PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM \
   --kernel-include-regex "psfm_pc_" -f csv -d $O/hard_sq -o s -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $O/hard_sq.log 2>&1 < /dev/null
python - <<'P' > $O/hard_sq_summary.json 2> $O/hard_sq_summary.err
import csv, glob, json, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_r"
f = glob.glob(O + "/hard_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
print(json.dumps({k: {"dispatches": len(calls[k]), **{c: v for c, v in agg[k].items()}} for k in agg}, indent=1))
P
cat $O/hard_sq_summary.json | head -60; tail -3 $O/hard_sq.log
rm -rf $O/hard_sq
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
