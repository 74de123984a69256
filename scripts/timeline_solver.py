"""Debug: per-block phase timeline of pc_iter launches (build with PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track
H, W, T, r = 1080, 1920, 21, 2
# PSFM_PROBE_HARD=1: the noisy distribution, whose solves run the launch chain (run with PSFM_PC_PERSIST=0 to see the launches)
dist = psfm_synth.HARD if os.environ.get("PSFM_PROBE_HARD") else dict(sigma=0.05, n_occluders=2)
d = psfm_synth.synth_sequence_torch(T, H, W, seed=2, stride2=True, **dist)
_hip.context().set_solver(1, 0)      # launch chain
_, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
_, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], 1.0)
info = run_track(d["flows_f"], occ, d["flows_f2"], occ2, r, return_device=True)
torch.cuda.synchronize()
fn = _hip.lib().psfm_debug_solver_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
buf = np.zeros((64, 1024, 4), np.uint64); n = ctypes.c_int(0)
assert fn(buf.ctypes.data, ctypes.byref(n)) == 0
print("recorded launches", n.value)
for s in range(min(n.value, 64)):
    if s not in (5, 10, 15, 18): continue
    t = buf[s, :512].astype(np.int64)
    lastb = int(buf[s, 1023, 0])
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print("launch %d: start %.2f..%.2f | loop done median %.2f p90 %.2f max %.2f | after ticket median %.2f max %.2f | last block %d: control done %.2f" % (
        s, us[:, 0].min(), us[:, 0].max(), np.median(us[:, 1]), np.percentile(us[:, 1], 90), us[:, 1].max(), np.median(us[:, 2]), us[:, 2].max(), lastb, us[lastb, 3]))
