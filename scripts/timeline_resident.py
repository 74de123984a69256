"""Debug: per-block phase timeline of eight rounds of one resident solve (psfm_pc_resident_kernel; build the library with
PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE or scripts/build_variant.py tl psfm_solver.hip -DPSFM_TIMELINE and PSFM_HIP_LIB=...)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track
H, W, T, r = 1080, 1920, 41, 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=6, stride2=True, **psfm_synth.HARD)
_hip.context().set_solver(1, 0)      # launch chain (as one resident launch per solve)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
_, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], 1.0)
info = run_track(d["flows_f"], occ, d["flows_f2"], occ2, r, return_device=True)
torch.cuda.synchronize()
fn = _hip.lib().psfm_debug_solver_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
buf = np.zeros((64, 1024, 4), np.uint64); n = ctypes.c_int(0)
assert fn(buf.ctypes.data, ctypes.byref(n)) == 0
tl = buf.reshape(-1)[:8 * 512 * 16].reshape(8, 512, 16).astype(np.int64)
print("recorded rounds", n.value)
names = ["start", "taps issued", "tracks done", "block sums", "published", "leader done", "totals in", "control done", "released", "totals broadcast", "derived"]
for rnd in range(min(n.value, 8)):
    t = tl[rnd]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    row = []
    for k, nm in enumerate(names):
        col = us[:32, k] if k == 5 else us[:, k]
        row.append("%s %.2f/%.2f/%.2f" % (nm, np.median(col), np.percentile(col, 90), col.max()))
    nxt = (tl[rnd + 1][:, 0].min() - t0) / 100.0 if rnd + 1 < min(n.value, 8) else float("nan")
    print("round %d (median/p90/max us from the first block's start; next round starts at %.2f): " % (rnd, nxt) + " | ".join(row))
# who are the stragglers?  (block ids of the ten slowest "tracks done - start" per round, and how the phases split for them)
for rnd in (2, 5):
    t = tl[rnd]; t0 = t[:, 0].min(); us = (t - t0) / 100.0
    dur = us[:, 2] - us[:, 0]
    order = np.argsort(-dur)[:12]
    print("round %d slowest blocks (id: start, taps-start, tracks-taps, sums-tracks):" % rnd,
          "; ".join("%d: %.2f %.2f %.2f %.2f" % (b, us[b, 0], us[b, 1] - us[b, 0], us[b, 2] - us[b, 1], us[b, 3] - us[b, 2]) for b in order))
    order = np.argsort(dur)[:6]
    print("   fastest:", "; ".join("%d: %.2f %.2f %.2f %.2f" % (b, us[b, 0], us[b, 1] - us[b, 0], us[b, 2] - us[b, 1], us[b, 3] - us[b, 2]) for b in order))
    print("   by b %% 8 (median tracks-start):", " ".join("%.2f" % np.median(dur[np.arange(512) % 8 == x]) for x in range(8)),
          "| by b // 256:", " ".join("%.2f" % np.median(dur[np.arange(512) // 256 == x]) for x in range(2)))
