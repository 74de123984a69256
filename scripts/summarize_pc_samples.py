"""Aggregate rocprofv3 PC-sampling CSVs (host_trap) of one kernel by source line and by instruction (runs on the GPU box:
the raw CSVs are large).  usage: summarize_pc_samples.py <dir> <kernel substring> <out.json>"""
import collections, csv, glob, json, os, sys
d, kern, out = sys.argv[1], sys.argv[2], sys.argv[3]
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)]
print("csv files:", [os.path.basename(f) for f in files])
disp = {}
for f in files:
    if f.endswith("kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            disp[row.get("Dispatch_Id")] = row.get("Kernel_Name", "")
by_line, by_inst, total, kept = collections.Counter(), collections.Counter(), 0, 0
for f in files:
    if "pc_sampling" not in os.path.basename(f):
        continue
    rd = csv.DictReader(open(f))
    print("columns:", rd.fieldnames)
    for row in rd:
        total += 1
        name = disp.get(row.get("Dispatch_Id"), "")
        if disp and kern not in name:
            continue
        kept += 1
        by_line[row.get("Instruction_Comment", "")] += 1
        by_inst[(row.get("Instruction", "") or "").split(" ")[0]] += 1
print("samples: %d total, %d in kernels matching %r" % (total, kept, kern))
json.dump({"total": total, "kept": kept, "by_line": by_line.most_common(400), "by_opcode": by_inst.most_common(80)}, open(out, "w"))
for k, v in by_line.most_common(70):
    print("%6d %5.1f%%  %s" % (v, 100.0 * v / max(kept, 1), k[-90:]))
for k, v in by_inst.most_common(30):
    print("%6d %5.1f%%  %s" % (v, 100.0 * v / max(kept, 1), k))
