#!/usr/bin/env python3
"""On the GPU box, right after scripts/profile_round3.sh: boil gpurun_out/<tag>/ down to what gets committed under
profiles/ (the raw per-dispatch CSVs are far beyond what travels back):
    <tag>_kernel_stats.csv / <tag>_opt_kernel_stats.csv   product kernels of the two rocprofv3 --kernel-trace --stats runs
    <tag>_pmc_summary.json / <tag>_opt_pmc_summary.json    per-kernel, per-launch averages of the PMC passes
    <tag>_bench.json                                         the default bench line of the same binary
Usage: python scripts/summarize_profiles3.py r03_a   ->   gpurun_out/<tag>_summary/
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "gpurun_out", tag + "_summary")
os.makedirs(dst, exist_ok=True)
csv.field_size_limit(1 << 30)


def short(name):
    n = name.split("(")[0].strip()
    return n[5:] if n.startswith("void ") else n


def stats(sub, stem, out, cmd):
    fn = os.path.join(src, sub, stem + "_kernel_stats.csv")
    if not os.path.exists(fn):
        return
    rows = list(csv.DictReader(open(fn)))
    keep = [r for r in rows if "psfm_" in r["Name"] or "rocprim" in r["Name"]]
    with open(os.path.join(dst, out), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -f csv -- %s\n" % cmd)
        f.write("# product kernels only (torch kernels of the synthetic-data generator omitted); durations in ns.\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in keep:
            f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (short(r["Name"])[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))


def pmc(prefix, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, pre in ((prefix + "pmc_fetch", "f"), (prefix + "pmc_write", "w"), (prefix + "pmc_sq", "s")):
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
    summary = {}
    for k, d in acc.items():
        summary[k] = {cn: sum(v) / len(v) for cn, v in d.items()}
        summary[k]["launches_sampled"] = max(len(v) for v in d.values())
        # launches that did real work only (no-op launches of stalled frames would dilute the averages)
        if "SQ_INSTS_VALU" in d:
            big = [i for i, v in enumerate(d["SQ_INSTS_VALU"]) if v > 0.25 * max(d["SQ_INSTS_VALU"])]
            summary[k]["launches_with_work"] = len(big)
    json.dump(summary, open(os.path.join(dst, out), "w"), indent=1, sort_keys=True)
    return summary


stats("stats", tag, tag + "_kernel_stats.csv", "python bench.py --steps 10 --warmup 2 --no-cpu --no-extras")
stats("hard_stats", tag + "_hard", tag + "_hard_kernel_stats.csv", "PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive python scripts/probe_solver.py  (1080p x 101 frames, sigma 0.3 + 5 % occluders: every solve rejects steps)")
stats("opt_stats", tag + "_opt", tag + "_opt_kernel_stats.csv", "PSFM_PROBE_MODES=fused python scripts/probe_solver.py  (1080p x 101 frames, flow_check x2 + track_optimize, 13 sequences)")
a = pmc("", tag + "_pmc_summary.json")
b = pmc("opt_", tag + "_opt_pmc_summary.json")
for name in ("bench.json", "bench.err"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))
for lg, outn in (("opt_under_rocprof.log", "_opt_probe.json"), ("hard_under_rocprof.log", "_hard_probe.json")):
    p = os.path.join(src, lg)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, tag + outn), "w").write(lines[-1])
shutil.rmtree(src, ignore_errors=True)
print(json.dumps({k: {c: round(v, 1) for c, v in d.items()} for k, d in (b or {}).items() if "seq" in k or "fused" in k or "chain_step" in k}, indent=1))
