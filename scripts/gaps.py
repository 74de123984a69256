import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "psfm" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last step: from last flow_check to end
fc = [k for k, r in enumerate(rows) if "flow_check" in r["Kernel_Name"]]
k0 = fc[-1]
seg = rows[k0:]
t0 = int(seg[0]["Start_Timestamp"])
busy = 0; prev_end = None; gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev_end is not None: gaps.append((s - prev_end, r["Kernel_Name"].split("(")[0][:40]))
    prev_end = e
total = prev_end - t0
print("last step: kernels %d  wall %.1f us  busy %.1f us  idle %.1f us" % (len(seg), total / 1e3, busy / 1e3, (total - busy) / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    agg[n][0] += g; agg[n][1] += 1
for n, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  gap before %-42s total %8.1f us  n=%4d  avg %6.2f us" % (n, g / 1e3, c, g / 1e3 / c))
dur = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].split("(")[0][:40]
    dur[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[n][1] += 1
for n, (g, c) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    print("  kernel %-46s total %8.1f us  n=%4d  avg %7.2f us" % (n, g / 1e3, c, g / 1e3 / c))
