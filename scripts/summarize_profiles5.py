#!/usr/bin/env python3
"""On the GPU box, inside scripts/profile_round5.sh: boil gpurun_out/<tag>/ down to what gets committed under profiles/
(the raw per-dispatch CSVs are far beyond what travels back) -- round 4's set plus the batched launches (batch_pmc.json, *_batch_*_kernel_stats.csv):
    <tag>_kernel_stats.csv / _opt_kernel_stats.csv / _hard_kernel_stats.csv   product kernels of the rocprofv3 --kernel-trace --stats runs
    <tag>_pmc_summary.json                                                       per-kernel, per-launch averages of every PMC pass
    traffic_chain_{fused,persist,step}.json, solver_valu.json                    what bench.py replays, stamped with the source hash
    <tag>_bench.json, <tag>_opt_probe.json, <tag>_hard_probe.json                the bench line / probe lines of the same binary
Usage: python scripts/summarize_profiles5.py r05_x   ->   gpurun_out/<tag>_summary/
"""
import collections
import csv
import importlib.util
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
spec = importlib.util.spec_from_file_location("psfm_build", os.path.join(ROOT, "particle-sfm_amd", "build.py"))
_b = importlib.util.module_from_spec(spec); spec.loader.exec_module(_b)
SHA = _b.source_hash()


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
        f.write("# rocprofv3 --kernel-trace --stats -f csv -- %s   (sources %s)\n" % (cmd, SHA))
        f.write("# product kernels only (torch kernels of the synthetic-data generator omitted); durations in ns.\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in keep:
            f.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (short(r["Name"])[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))


def counters(sub, pre):
    """{kernel: {counter: per-launch average over the launches that did real work}}"""
    fn = os.path.join(src, sub, pre + "_counter_collection.csv")
    if not os.path.exists(fn):
        return {}
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "psfm_" not in r["Kernel_Name"]:
            continue
        per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for (_, k, cn), v in per.items():
        acc[k][cn].append(v)
    out = {}
    for k, d in acc.items():
        n = max(len(v) for v in d.values())
        # launches that did real work only (no-op launches behind a stall flag / past the end of a sequence would dilute the averages)
        ref = d.get("SQ_INSTS_VALU") or next(iter(d.values()))
        big = [i for i, v in enumerate(ref) if v > 0.25 * max(ref)] if max(ref) > 0 else list(range(len(ref)))
        out[k] = {cn: sum(v[i] for i in big if i < len(v)) / max(len(big), 1) for cn, v in d.items()}
        out[k]["launches_sampled"] = n
        out[k]["launches_with_work"] = len(big)
    return out


def probe_line(log):
    p = os.path.join(src, log)
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


stats("stats", tag, tag + "_kernel_stats.csv", "python bench.py --steps 10 --warmup 2 --no-cpu --no-extras")
stats("opt_stats", tag + "_opt", tag + "_opt_kernel_stats.csv", "PSFM_PROBE_MODES=adaptive python scripts/probe_solver.py  (1080p x 101 frames, flow_check x2 + track_optimize, clean flows)")
stats("hard_stats", tag + "_hard", tag + "_hard_kernel_stats.csv", "PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive python scripts/probe_solver.py  (1080p x 101 frames, sigma 0.3 + 5 % occluders: every solve rejects steps)")
stats("batch_davis_stats", tag + "_bd", tag + "_batch_davis_kernel_stats.csv", "python scripts/run_batch_once.py davis 16 5  (psfm_connect_batch: 16 x 480x854x50 frames, sample_ratio 4, flow_check + track)")
stats("batch_sintel_stats", tag + "_bs", tag + "_batch_sintel_kernel_stats.csv", "python scripts/run_batch_once.py sintel 16 5  (psfm_connect_batch: 16 x 436x1024x50 frames, sample_ratio 2, flow_check x2 + track_optimize)")
stats("batch_scannet_stats", tag + "_bn", tag + "_batch_scannet_kernel_stats.csv", "python scripts/run_batch_once.py scannet 4 3  (psfm_connect_batch: 4 x 480x640x200 frames, sample_ratio 1, thres 3.0, flow_check x2 + track_optimize)")
summary = {"source_sha16": SHA, "round": tag}
for sub, pre in (("batch_davis_fetch", "f"), ("batch_davis_write", "w"), ("batch_sintel_pmc_sq", "s"), ("batch_scannet_pmc_sq", "s"),
                 ("opt_pmc_sq_436x1024x50x2", "s"), ("opt_pmc_sq_480x640x200x1", "s"),
                 ("fused_fetch", "f"), ("fused_write", "w"), ("two_fetch", "f"), ("two_write", "w"), ("step_fetch", "f"), ("step_write", "w"),
                 ("pmc_sq", "s"), ("opt_pmc_sq", "s"), ("opt_fetch", "f"), ("opt_write", "w"), ("hard_pmc_sq", "s"),
                 ("opt_fetch_436x1024x50x2", "f"), ("opt_write_436x1024x50x2", "w"), ("opt_fetch_480x640x200x1", "f"), ("opt_write_480x640x200x1", "w")):
    summary[sub] = counters(sub, pre)
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)

# ---- HBM-side traffic of the chain kernels: the L2s' fabric request counters in 32-byte units, checked against the stand-alone
#      flow_check launch, whose read and write volumes are known exactly ----
H, W, NF = 1080, 1920, 100
try:
    tf, tw = summary["two_fetch"], summary["two_write"]
    fc = [k for k in tf if "flow_check" in k][0]
    rd_ok = tf[fc]["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32.0 / (16.0 * H * W * NF)
    wr_ok = tw[fc]["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] * 32.0 / (1.0 * H * W * NF)
    note = ("bytes = 32 x (TCC_EA0_RDREQ_DRAM_32B + TCC_EA0_WRREQ_WRITE_DRAM_32B + TCC_EA0_WRREQ_ATOMIC_DRAM_32B), the L2s' fabric-side "
            "requests in 32-byte units (a 128-byte request counts 4; Infinity-Cache hits are included).  Checked on the stand-alone "
            "flow_check launch of the same run: counted / known = %.4f for its reads (16*H*W*100 bytes), %.4f for its writes (H*W*100).  "
            "FETCH_SIZE (= 64 bytes x TCC_EA0_RDREQ on this chip, whose requests are 128 bytes) would report half of the reads" % (rd_ok, wr_ok))

    def traffic(fetch, write, pick, label, how):
        k = [x for x in fetch if pick in x][0]
        rd = fetch[k]["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32.0
        wr = (write[k]["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] + write[k].get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0)) * 32.0
        return {"kernel": k + label, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                "read_requests_per_launch": fetch[k].get("TCC_EA0_RDREQ_sum"),
                "FETCH_SIZE_KB_per_launch_equivalent": fetch[k].get("TCC_EA0_RDREQ_sum", 0.0) * 64.0 / 1024.0,     # (what FETCH_SIZE reports here)
                "counter_check_on_flow_check": {"reads_counted_over_known": rd_ok, "writes_counted_over_known": wr_ok},
                "source": "scripts/profile_round5.sh %s: rocprofv3 --pmc, reads and writes in separate passes, %s" % (tag, how),
                "note": note, "flow_check_kernel": fc, "round": tag, "source_sha16": SHA}
    base = "bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
    json.dump(traffic(summary["fused_fetch"], summary["fused_write"], "chain_persist", " (flow_check fused in)", base),
              open(os.path.join(dst, "traffic_chain_fused.json"), "w"), indent=1)
    json.dump(traffic(tf, tw, "chain_persist", "", "PSFM_BENCH_TWO_CALLS=1 " + base), open(os.path.join(dst, "traffic_chain_persist.json"), "w"), indent=1)
    json.dump(traffic(summary["step_fetch"], summary["step_write"], "chain_step", "", "PSFM_BENCH_CHAIN_MODE=1 " + base),
              open(os.path.join(dst, "traffic_chain_step.json"), "w"), indent=1)
except Exception as e:      # noqa: BLE001
    print("traffic files not written:", type(e).__name__, e)

# ---- VALU wave-instructions of the solver launches (what bench.py scales by its own run's tracks x iterations) ----
try:
    opt, hard = probe_line("opt_stats.log"), probe_line("hard_stats.log")
    sv = {"source_sha16": SHA, "round": tag}
    ko = [k for k in summary["opt_pmc_sq"] if "psfm_seq_kernel" in k]
    if ko and opt:
        a = opt["adaptive"]
        wi = summary["opt_pmc_sq"][ko[0]]["SQ_INSTS_VALU"]
        waves = a["tracks_per_solve"] / 64.0
        per_it = 320        # static census of the iteration loop (scripts/isa_count.py)
        hbm, hsrc = None, None
        try:
            fo = [v for k, v in summary["opt_fetch"].items() if "psfm_seq_kernel" in k][0]
            wo = [v for k, v in summary["opt_write"].items() if "psfm_seq_kernel" in k][0]
            rd = fo["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32.0
            wr = (wo["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] + wo.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0)) * 32.0
            hbm = rd + wr
            hsrc = ("%s_pmc_summary.json opt_fetch / opt_write (separate rocprofv3 --pmc passes of scripts/probe_solver.py, 1080p x 101 frames, "
                    "clean flows): 32 x TCC_EA0_RDREQ_DRAM_32B = %.1f MB read + 32 x (TCC_EA0_WRREQ_WRITE_DRAM_32B + ..ATOMIC_DRAM_32B) = "
                    "%.1f MB written per launch with work; sources %s" % (tag, rd / 1e6, wr / 1e6, SHA))
        except Exception:       # noqa: BLE001
            pass
        by_shape = {}
        if hbm:
            by_shape["1080x1920x2"] = hbm
        for shp, key in (("436x1024x2", "436x1024x50x2"), ("480x640x1", "480x640x200x1")):
            try:
                fo = [v for k, v in summary["opt_fetch_" + key].items() if "psfm_seq_kernel" in k][0]
                wo = [v for k, v in summary["opt_write_" + key].items() if "psfm_seq_kernel" in k][0]
                by_shape[shp] = fo["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32.0 + (wo["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] +
                                                                           wo.get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0)) * 32.0
            except Exception:       # noqa: BLE001
                pass
        sv.update({"kernel": ko[0], "valu_per_wave_per_iteration": per_it,
                   "valu_per_wave_fixed": wi / waves - per_it * a["iterations_per_solve"],
                   "hbm_bytes_per_launch": hbm, "traffic_source": hsrc, "hbm_bytes_measured_at": "1080p, sample_ratio 2",
                   "hbm_bytes_per_launch_by_shape": by_shape,      # "HxWxsample_ratio": same counters, scripts/probe_solver.py on that shape
                   "source": "%s_pmc_summary.json opt_pmc_sq: SQ_INSTS_VALU %.2f M per launch with work, %.0f tracks (%.0f waves) and %.2f iterations "
                             "per solve (probe of the same run); %d per wave and iteration from the static census, the rest fixed"
                             % (tag, wi / 1e6, a["tracks_per_solve"], waves, a["iterations_per_solve"], per_it)})
    kh = [k for k in summary["hard_pmc_sq"] if "psfm_pc_resident" in k]
    if kh and hard:
        a = hard["adaptive"]
        c = summary["hard_pmc_sq"][kh[0]]
        wi, fixed, nwaves = c["SQ_INSTS_VALU"], 190, 2048
        rounds = a["iterations_per_solve"] + 1.0        # + iteration 0
        sv["chain"] = {"kernel": kh[0] + " (the launch chain's trust-region loop as ONE launch, the tracks' state on chip)",
                       "valu_per_track_iteration": (wi - fixed * nwaves * rounds) / (a["track_iterations_per_solve"] / 64.0),
                       "valu_per_wave_per_iteration_fixed": fixed, "waves": nwaves,
                       # (two waves per SIMD: the SIMD's time is half the summed wave time)
                       "valu_issue_busy_frac": c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0) / 2.0, 1.0),
                       "wait_frac": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                       "source": "%s_pmc_summary.json hard_pmc_sq: SQ_INSTS_VALU %.2f M wave-instructions per solve (%.1f iterations, %.2f M "
                                 "track-iterations per solve: probe of the same run; rocprofv3 --pmc with --kernel-include-regex psfm_pc_) = %d per "
                                 "wave and round fixed (static census: block sums, hand-off, control; 512 blocks x 4 waves) + the rest per 64 "
                                 "track-iterations" % (tag, wi / 1e6, a["iterations_per_solve"], a["track_iterations_per_solve"] / 1e6, fixed),
                       "round": tag}
    json.dump(sv, open(os.path.join(dst, "solver_valu.json"), "w"), indent=1)
    for line, name in ((opt, "_opt_probe.json"), (hard, "_hard_probe.json")):
        if line:
            json.dump(line, open(os.path.join(dst, tag + name), "w"))
except Exception as e:      # noqa: BLE001
    print("solver_valu.json not written:", type(e).__name__, e)

# ---- the batched launches: measured HBM bytes of the DAVIS x 16 chain step, measured VALU wave-instructions of the Sintel x 16 / ScanNet x 4
#      frame launches, and their average durations from the kernel-trace runs of the same binary ----
try:
    def avg_ns(sub, stem, pick):
        fn = os.path.join(src, sub, stem + "_kernel_stats.csv")
        for r in csv.DictReader(open(fn)):
            if pick in r["Name"]:
                return float(r["AverageNs"]), int(r["Calls"])
        return None, 0
    bp = {"source_sha16": SHA, "round": tag}
    fd, wd = summary.get("batch_davis_fetch", {}), summary.get("batch_davis_write", {})
    kd = [k for k in fd if "chain_step_batch" in k]
    if kd:
        rd = fd[kd[0]]["TCC_EA0_RDREQ_DRAM_32B_sum"] * 32.0
        wr = (wd[kd[0]]["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] + wd[kd[0]].get("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum", 0.0)) * 32.0
        ns, calls = avg_ns("batch_davis_stats", tag + "_bd", "chain_step_batch")
        bp["davis_b16"] = {"kernel": kd[0], "batch": 16, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                           "avg_launch_ns_kernel_trace": ns, "launches_kernel_trace": calls,
                           "frac_physical": ((rd + wr) / (ns * 1e-9) / 8e12) if ns else None,
                           "source": "scripts/profile_round5.sh %s: rocprofv3 --pmc (reads and writes in separate passes, launches with work only) and "
                                     "--kernel-trace --stats of python scripts/run_batch_once.py davis 16; the side stream's flow_check runs beside "
                                     "these launches" % tag}
    for key, sub, stem, shape, B in (("sintel_b16", "batch_sintel", tag + "_bs", "sintel", 16), ("scannet_b4", "batch_scannet", tag + "_bn", "scannet", 4)):
        c = summary.get(sub + "_pmc_sq", {})
        kk = [k for k in c if "seq_batch" in k]
        if not kk:
            continue
        ns, calls = avg_ns(sub + "_stats", stem, "seq_batch")
        wi = c[kk[0]]["SQ_INSTS_VALU"]
        bp[key] = {"kernel": kk[0], "batch": B, "valu_wave_instructions_per_launch": wi, "launches_with_work": c[kk[0]].get("launches_with_work"),
                   "avg_launch_ns_kernel_trace": ns, "launches_kernel_trace": calls,
                   "valu_issue_frac": (wi / (ns * 1e-9) / 614.4e9) if ns else None,
                   "valu_busy_frac": c[kk[0]].get("SQ_ACTIVE_INST_VALU", 0.0) / max(c[kk[0]].get("SQ_WAVE_CYCLES", 1.0) / 4.0, 1.0),
                   "wait_frac": c[kk[0]].get("SQ_WAIT_ANY", 0.0) / max(c[kk[0]].get("SQ_WAVE_CYCLES", 1.0), 1.0),
                   "source": "scripts/profile_round5.sh %s: SQ_INSTS_VALU per launch with work (rocprofv3 --pmc, --kernel-include-regex psfm_seq_batch) and the "
                             "average launch duration of the --kernel-trace --stats run (all launches, the spare ones of every window included) of "
                             "python scripts/run_batch_once.py %s %d; 614.4 G wave-instructions/s = 1024 SIMDs x 2.4 GHz / 4" % (tag, shape, B)}
    # the single-sequence frame kernel on the same shapes, same counters: what the batch is compared with
    for key, sub in (("sintel_single", "opt_pmc_sq_436x1024x50x2"), ("scannet_single", "opt_pmc_sq_480x640x200x1")):
        c = summary.get(sub, {})
        kk = [k for k in c if "psfm_seq_kernel" in k]
        if kk:
            bp[key] = {"kernel": kk[0], "valu_wave_instructions_per_launch": c[kk[0]]["SQ_INSTS_VALU"],
                       "valu_busy_frac": c[kk[0]].get("SQ_ACTIVE_INST_VALU", 0.0) / max(c[kk[0]].get("SQ_WAVE_CYCLES", 1.0) / 4.0, 1.0),
                       "wait_frac": c[kk[0]].get("SQ_WAIT_ANY", 0.0) / max(c[kk[0]].get("SQ_WAVE_CYCLES", 1.0), 1.0)}
    for log, name in (("batch_davis_stats.log", "davis_b16"), ("batch_sintel_stats.log", "sintel_b16"), ("batch_scannet_stats.log", "scannet_b4")):
        line = probe_line(log)
        if line and name in bp:
            bp[name]["run"] = line
    json.dump(bp, open(os.path.join(dst, "batch_pmc.json"), "w"), indent=1)
except Exception as e:      # noqa: BLE001
    print("batch_pmc.json not written:", type(e).__name__, e)

for name in ("bench.json", "bench.err"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))
shutil.rmtree(src, ignore_errors=True)
print(json.dumps({k: v for k, v in summary.items() if k in ("source_sha16",)}))
