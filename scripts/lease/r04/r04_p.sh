#!/bin/bash
# round 4: XCD-banded lane ownership of the persistent loop (psfm_vblock): exactness, us per step, fabric-side read bytes; PSFM_PP_XCD=0 = id order
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  PSFM_PP_XCD=$v timeout 200 python scripts/probe_persist_variant.py 2> $O/v$v.err | python -c "import json,sys; d=json.load(sys.stdin); t=d['timing_1080p']; print('PSFM_PP_XCD=$v', 'exact' if d['ok'] else 'NOT exact', 'track us/step %.2f' % t['track_chain_us_per_step'], 'connect us/step %.2f' % t['connect_us_per_step'], 'connect ms %.3f' % t['connect_ms'], 'track ms %.3f' % t['track_ms'])" | tee -a $O/ab.txt
done
export TMPDIR=/tmp; cd /tmp
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
for v in 1 0; do
  PSFM_PP_XCD=$v PSFM_BENCH_TWO_CALLS=1 timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum -f csv -d $O/two_rd$v -o r -- $B2 > $O/two_rd$v.log 2>&1 < /dev/null
  PSFM_PP_XCD=$v timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum -f csv -d $O/fused_rd$v -o r -- $B2 > $O/fused_rd$v.log 2>&1 < /dev/null
done
python - <<'P'
import csv, glob, os, collections
csv.field_size_limit(1 << 30)
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_p"
for sub in ("two_rd1", "two_rd0", "fused_rd1", "fused_rd0"):
    f = glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "chain_persist" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_EA0_RDREQ_DRAM_32B_sum": per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(per.values())
    print(sub, "chain_persist reads per launch: %.3f GB" % (32.0 * sum(v) / len(v) / 1e9))
P
for d in two_rd1 two_rd0 fused_rd1 fused_rd0; do rm -rf $O/$d; done
