"""Debug: per-block phase timeline of ONE chain_step launch (needs a build with PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE).

    PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE python particle-sfm_amd/build.py && python scripts/timeline_chain.py [frame]
"""
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
T, H, W, r = 101, 1080, 1920, 2
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
lib = _hip.lib()
fn = lib.psfm_debug_timeline
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
for it in range(3):
    assert fn(frame if it == 2 else -1, None, 0) == 0
    info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
    torch.cuda.synchronize()
NB = 4050
buf = np.zeros((NB, 8), np.uint64)
assert fn(0, buf.ctypes.data, NB) == 0
t = buf[:, :7].astype(np.int64)
act = t[:, 6] > 0                      # blocks that ran to the end
t0 = t[act | (t[:, 0] > 0), 0].min()
us = (t - t0) / 100.0                   # 100 MHz
xcc = (buf[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
a = us[act]
print("blocks total %d, active %d" % (NB, act.sum()))
names = ["start", "RT1 done", "after B1", "t0 atomics done", "after B2", "finish done", "end"]
for k, n in enumerate(names):
    print("%-16s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (
        n, a[:, k].min(), np.percentile(a[:, k], 10), np.median(a[:, k]), np.percentile(a[:, k], 90), a[:, k].max()))
life = a[:, 6] - a[:, 0]
print("lifetime  median %.2f p90 %.2f max %.2f" % (np.median(life), np.percentile(life, 90), life.max()))
for k in range(1, 7):
    dd = a[:, k] - a[:, k - 1]
    print("phase %-16s median %5.2f p90 %5.2f" % (names[k], np.median(dd), np.percentile(dd, 90)))
idle = us[~act & (t[:, 0] > 0)]
if len(idle):
    print("idle blocks: %d, start min %.2f max %.2f" % (len(idle), idle[:, 0].min(), idle[:, 0].max()))
# start time vs block index
idx = np.nonzero(act)[0]
for lo in range(0, idx.max() + 1, 256):
    m = (idx >= lo) & (idx < lo + 256)
    if m.any():
        print("blocks %4d..%4d  start %5.2f..%5.2f  end %5.2f..%5.2f" % (lo, lo + 255, a[m, 0].min(), a[m, 0].max(), a[m, 6].min(), a[m, 6].max()))
print("xcc histogram of active blocks", np.bincount(xcc[act], minlength=8))
