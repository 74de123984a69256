"""Which way of running the recurrence is faster for which shape?  psfm_connect, 100 frames, per-frame launches (mode 1)
vs the persistent loop (mode 2), end to end (flow_check + recurrence + finalize), median of 8 runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, run_track
from point_trajectory.utils import flow_check_device
ctx = _hip.context()
shapes = [(1080, 1920, 2), (1080, 1920, 4), (720, 1280, 2), (720, 1280, 1), (480, 640, 1), (480, 854, 4), (436, 1024, 2), (540, 960, 1), (2160, 3840, 4)]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 101
for (H, W, r) in shapes:
    d = psfm_synth.synth_sequence_torch(T if H < 2000 else 31, H, W, seed=1, sigma=0.05, n_occluders=2, stride2=False)
    nf = d["flows_f"].shape[0]
    out = {}
    for mode in (1, 2, 0):
        ctx.set_chain_mode(mode)
        try:
            ts = []
            for it in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                info = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            out[mode] = (1e3 * float(np.median(ts[2:])), info.n_points, int(info.chain_mode))
        except Exception as e:
            out[mode] = (float("nan"), 0, -1)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    tr = {}
    for mode in (1, 2):
        ctx.set_chain_mode(mode)
        try:
            ts = []
            for it in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                info = run_track(d["flows_f"], occ, None, None, r, return_device=True)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            tr[mode] = 1e3 * float(np.median(ts[2:]))
        except Exception as e:
            tr[mode] = float("nan")
    G = ((H + r - 1) // r) * ((W + r - 1) // r)
    print("%4dx%4d r=%d frames %3d  G=%7d P=%8d | per-frame %7.3f ms (%5.1f us/frame)  persistent %7.3f ms (%5.1f us/frame)  ratio %.2f" % (
        H, W, r, nf, G, H * W, out[1][0], 1e3 * out[1][0] / nf, out[2][0], 1e3 * out[2][0] / nf, out[2][0] / out[1][0]))
    print("        psfm_connect in the DEFAULT mode: %7.3f ms (ran the %s)" % (out[0][0], "persistent loop" if out[0][2] == 2 else "per-frame launches"))
    print("        psfm_track on ready maps: per-frame %7.3f ms  persistent %7.3f ms  ratio %.2f" % (tr[1], tr[2], tr[2] / tr[1]))
    del d
ctx.set_chain_mode(0)
