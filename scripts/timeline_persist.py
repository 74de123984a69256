"""Debug: per-block phase timeline of ONE frame of the persistent loop (build with PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track

frame = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, H, W, r = 101, int(sys.argv[2]) if len(sys.argv) > 4 else 1080, int(sys.argv[3]) if len(sys.argv) > 4 else 1920, int(sys.argv[4]) if len(sys.argv) > 4 else 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
lib = _hip.lib()
fn = lib.psfm_debug_persist_timeline
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
_hip.context().set_chain_mode(2)
for it in range(3):
    assert fn(frame if it == 2 else -1, None, 0) == 0
    info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    torch.cuda.synchronize()
NB = int(info.lane_capacity) // (256 + 32)
buf2 = np.zeros((2, 4096, 8), np.uint64)
assert fn(0, buf2.ctypes.data, NB) == 0
buf = buf2[0, :NB]
t = buf.astype(np.int64)
t_next = buf2[1, :NB].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["frame start", "barrier passed", "C done (sync1)", "issue+atomics", "sync2", "results done", "stores acked", "sync F"]
for k, n in enumerate(names):
    a = us[:, k]
    print("%-16s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f" % (n, a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max()))
for k in range(1, 8):
    dd = us[:, k] - us[:, k - 1]
    print("phase -> %-16s median %6.2f p90 %6.2f max %6.2f" % (names[k], np.median(dd), np.percentile(dd, 90), dd.max()))
dur = us[:, 7] - us[:, 1]
order = np.argsort(dur)
print("fastest blocks", order[:12], np.round(dur[order[:12]], 1))
print("slowest blocks", order[-12:], np.round(dur[order[-12:]], 1))
for lo in range(0, NB, 128):
    m = slice(lo, lo + 128)
    print("blocks %4d..%4d: post-barrier duration median %6.1f max %6.1f | results phase median %5.1f | ack median %5.1f" % (
        lo, lo + 127, np.median(dur[m]), dur[m].max(), np.median((us[:, 5] - us[:, 4])[m]), np.median((us[:, 6] - us[:, 5])[m])))
for x in range(8):
    m = (np.arange(NB) % 8) == x
    print("xcd-slot %d: median %6.1f" % (x, np.median(dur[m])))

st = np.zeros((NB, 8), np.int32)
assert fn(0, st.ctypes.data, -NB) == 0
print("per block at this frame: births mean %.2f max %d | adopted mean %.2f | free local lanes mean %.2f | live guests mean %.1f max %d | free guests mean %.1f" % (
    st[:, 0].mean(), st[:, 0].max(), st[:, 1].mean(), st[:, 2].mean(), st[:, 3].mean(), st[:, 3].max(), st[:, 4].mean()))
print("blocks that pop lanes: %d (lanes %d) | blocks that push lanes: %d (lanes %d)" % (
    (st[:, 5] > 0).sum(), st[:, 5].clip(0).sum(), (st[:, 6] > 0).sum(), st[:, 6].clip(0).sum()))
for name, m in (("pop", st[:, 5] > 0), ("push only", (st[:, 6] > 0) & (st[:, 5] <= 0)), ("neither", (st[:, 6] <= 0) & (st[:, 5] <= 0))):
    if m.any():
        print("  %-10s blocks: post-barrier duration median %.1f p90 %.1f max %.1f" % (name, np.median(dur[m]), np.percentile(dur[m], 90), dur[m].max()))
print("slowest 16 blocks: idx, post-barrier us, [C, issue, sync2, results, ack, syncF], births, adopted, free lanes, live guests, need, push")
ph = np.diff(us[:, 1:8], axis=1)
for b in order[-16:]:
    print("  %4d %6.1f %s  nb=%d nad=%d npd=%d ngl=%d need=%d push=%d" % (b, dur[b], np.round(ph[b], 1), st[b, 0], st[b, 1], st[b, 2], st[b, 3], st[b, 5], st[b, 6]))
big = st[:, 0] >= 64
print("blocks with >= 64 births: %d, post-barrier median %.1f max %.1f; others median %.1f max %.1f" % (big.sum(), np.median(dur[big]) if big.any() else 0, dur[big].max() if big.any() else 0, np.median(dur[~big]), dur[~big].max()))

usn = (t_next - t0) / 100.0
print("frame t: last sync F %.2f | frame t+1: first barrier pass %.2f, median %.2f, last %.2f  => release latency after the last arrival %.2f us" % (
    us[:, 7].max(), usn[:, 1].min(), np.median(usn[:, 1]), usn[:, 1].max(), usn[:, 1].min() - us[:, 7].max()))
print("period (median barrier pass t+1 - t): %.2f us" % (np.median(usn[:, 1]) - np.median(us[:, 1])))
late = usn[:, 0] - us[:, 7]
print("arrive -> next frame start: median %.2f ; next frame start -> barrier pass: median %.2f min %.2f (blocks that were last: %s)" % (
    np.median(late), np.median(usn[:, 1] - usn[:, 0]), (usn[:, 1] - usn[:, 0]).min(), np.round(np.sort(usn[:, 1] - usn[:, 0])[:5], 2)))
