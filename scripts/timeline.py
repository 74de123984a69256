import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
idx = np.nonzero(raw[:, 0] > 0)[0]
a = raw[idx]
t0 = a[:, 0].min()
a = (a - t0) / 100.0      # s_memrealtime: 100 MHz constant clock -> microseconds
print("blocks", len(a), "span(us) %.2f" % a[:, 6].max())
names = ["start", "early loads back", "gathers issued", "barrier1", "barrier2(atomics)", "gathers back", "end"]
for k in range(7):
    v = a[:, k]
    print("%-20s min %6.2f p10 %6.2f med %6.2f p90 %6.2f max %6.2f" % (names[k], v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
d = np.diff(a[:, :7], axis=1)
for k in range(6):
    print("phase %d->%d  med %5.2f  p90 %5.2f" % (k, k + 1, np.median(d[:, k]), np.percentile(d[:, k], 90)))
print("start by block index (every 128th):", np.round(a[::128, 0], 2).tolist())
print("end   by block index (every 128th):", np.round(a[::128, 6], 2).tolist())
