"""Turn the scratch output of scripts/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/:
  <tag>_kernel_stats.csv   product kernels of the rocprofv3 --kernel-trace --stats run
  <tag>_pmc_summary.json   per-kernel averages of the PMC passes (per launch)
  <tag>_bench.json         the default bench line
  traffic_chain_{step,persist}.json  HBM bytes per launch of the chain kernels (FETCH_SIZE calibrated on flow_check)
Usage: python scripts/summarize_profiles.py r01_d [--fused]   (--fused: the bench step ran psfm_connect with flow_check inside the loop)
"""
import csv, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
csv.field_size_limit(1 << 30)

def short(name):
    n = name.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n

# ---- kernel stats ----
rows = list(csv.DictReader(open(os.path.join(src, "stats", tag + "_kernel_stats.csv"))))
keep = [r for r in rows if "psfm_" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]]
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -f csv -- python bench.py --steps 10 --warmup 2 --no-cpu --no-extras\n")
    f.write("# product kernels only (torch kernels of the synthetic-data generator omitted); durations in ns.\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for r in keep:
        f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (short(r["Name"])[:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                  r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))
# ---- PMC ----
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, pre in (("pmc_fetch", "f"), ("pmc_write", "w"), ("pmc_sq", "s")):
    fn = os.path.join(src, sub, pre + "_counter_collection.csv")
    if not os.path.exists(fn):
        continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "psfm_" not in r["Kernel_Name"]:
            continue
        per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, cn), v in per.items():
        acc[k][cn].append(v)
summary = {k: {cn: sum(v) / len(v) for cn, v in d.items()} for k, d in acc.items()}
for k in summary:
    summary[k]["launches_sampled"] = max(len(v) for v in acc[k].values())
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
# ---- traffic of the chain kernels ----
H, W, NF = 1080, 1920, 100
fc = summary.get("psfm_flow_check_x4_kernel")
cal, cal_src = None, ""
if fc and "FETCH_SIZE" in fc:
    cal = (16.0 * H * W * NF / 1024.0) / fc["FETCH_SIZE"]      # known read volume / counter (KB)
    cal_src = "calibrated in the same run on psfm_flow_check_x4_kernel"
else:   # the stand-alone flow_check kernel did not run (fused step): the calibration of the previous collection
    prev = os.path.join(dst, "traffic_chain_persist.json")
    if os.path.exists(prev):
        cal = json.load(open(prev))["fetch_calibration"]
        cal_src = "calibration factor taken from profiles/traffic_chain_persist.json (same device type, flow_check kernel of that run)"
if cal:
    for kern, out in (("psfm_chain_step_kernel<2>", "traffic_chain_step.json"), ("psfm_chain_persist_kernel<2>", "traffic_chain_fused.json" if "--fused" in sys.argv else "traffic_chain_persist.json")):
        k = summary.get(kern)
        if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
            continue
        hbm = (k["FETCH_SIZE"] * cal + k["WRITE_SIZE"]) * 1024.0
        json.dump({"kernel": kern,
                   "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 --no-cpu --no-extras)" % tag,
                   "FETCH_SIZE_KB_per_launch": k["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": k["WRITE_SIZE"],
                   "fetch_calibration": cal,
                   "calibration_note": "MI355X_MICROARCH.md: FETCH_SIZE is uncalibrated on gfx950 for accesses other than 16 B/lane -> " + cal_src + ", whose read volume is known exactly (16*H*W*100 bytes per launch, 8-byte/lane loads like the chain kernels' taps). WRITE_SIZE is used as is.",
                   "hbm_bytes_per_launch": hbm}, open(os.path.join(dst, out), "w"), indent=1)
        print(kern, "HBM bytes per launch %.4g (fetch cal %.3f)" % (hbm, cal))
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    open(os.path.join(dst, tag + "_bench.json"), "w").write(open(bj).read())
print(json.dumps({k: {c: round(v, 1) for c, v in d.items()} for k, d in summary.items() if "chain" in k or "flow_check" in k}, indent=1))
