#!/bin/bash
# round 4: XCD-banded tiles in finalize's gather kernels (PSFM_GATHER_BAND=0: block order): finalize span, checksum, parity tests, and the
# gather's fabric-side bytes either way
O=$GRAFT_REPO_ROOT/gpurun_out/r04_za; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do PSFM_GATHER_BAND=$v timeout 300 python scripts/probe_finalize.py 2> /dev/null | tail -1 | sed "s/^/band=$v /" | tee -a $O/ab.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  PSFM_GATHER_BAND=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum --kernel-include-regex "psfm_gather" -f csv -d $O/rd$v -o r -- python $GRAFT_REPO_ROOT/scripts/probe_finalize.py > $O/rd$v.log 2>&1
  python - <<PY | tee -a $O/ab.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("$O/rd$v/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), x in per.items(): acc[c].append(x)
print("band=$v", {c: round(sum(x) / len(x) * (32 if "32B" in c else 1) / 1e6, 1) for c, x in acc.items()}, "(MB read per launch / M requests)")
PY
  rm -rf $O/rd$v
done
