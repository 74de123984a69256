"""GPU probe: persistent frame loop (chain mode 2) vs per-frame launches (mode 1): identical results? timing?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_track

cases = [(7, 48, 64, 2, 3), (9, 45, 70, 1, 5), (12, 50, 66, 3, 7), (10, 52, 61, 4, 9), (21, 200, 300, 2, 11), (31, 270, 480, 1, 12)]
if len(sys.argv) > 1 and sys.argv[1] == "one":
    cases = cases[:1]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    cases = [(101, 1080, 1920, 2, 0)]
ctx = _hip.context()
ok_all = True
for (T, H, W, r, seed) in cases:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, sigma=0.3 if H < 1000 else 0.05, n_occluders=2, stride2=False)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    res = {}
    for mode in [int(x) for x in os.environ.get('PSFM_MODES', '1,2').split(',')]:
        ctx.set_chain_mode(mode)
        try:
            for it in range(2):
                torch.cuda.synchronize(); t0 = time.time()
                out = run_track(d["flows_f"], occ, None, None, r)
                torch.cuda.synchronize(); t1 = time.time()
            res[mode] = (out, (t1 - t0) * 1e3)
            print("   mode %d: %.3f ms, info %s" % (mode, (t1 - t0) * 1e3, out.info))
        except Exception as e:
            print("case", (T, H, W, r), "mode", mode, "FAILED:", e)
            res[mode] = None
    a, b = res.get(1), res.get(2)
    if a is None or b is None:
        ok_all = False
        continue
    A, B = a[0], b[0]
    same = (len(A) == len(B) and np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length)
            and np.array_equal(A.off, B.off) and np.array_equal(A.xy, B.xy))
    ok_all &= bool(same)
    print("case T=%d %dx%d r=%d: n_traj %d/%d points %d/%d  identical=%s   per-frame %.3f ms  persistent %.3f ms  modes %s/%s lanes %s/%s" % (
        T, H, W, r, len(A), len(B), A.n_points, B.n_points, same, a[1], b[1], A.info.get("chain_mode"), B.info.get("chain_mode"),
        A.info.get("n_lanes_peak"), B.info.get("n_lanes_peak")))
    if not same and len(A) == len(B):
        nb = int((A.birth != B.birth).sum()); nl = int((A.length != B.length).sum())
        print("   birth diffs %d, length diffs %d" % (nb, nl))
        if A.xy.shape == B.xy.shape:
            dd = np.abs(A.xy - B.xy).max(axis=1)
            print("   xy diffs: %d rows, max %.3g, first rows %s" % (int((dd > 0).sum()), dd.max(), np.nonzero(dd > 0)[0][:10]))
ctx.set_chain_mode(0)
print("ALL IDENTICAL" if ok_all else "MISMATCH")
if not ok_all and os.environ.get("PSFM_DEBUG"):
    T, H, W, r, seed = cases[0]
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, sigma=0.3, n_occluders=2, stride2=False)
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    ctx.set_chain_mode(1); A = run_track(d["flows_f"], occ, None, None, r)
    ctx.set_chain_mode(2); B = run_track(d["flows_f"], occ, None, None, r)
    from collections import Counter
    ca = Counter(zip(A.birth.tolist(), A.length.tolist())); cb = Counter(zip(B.birth.tolist(), B.length.tolist()))
    print("missing in persistent (birth,len):count", sorted((ca - cb).items())[:40])
    print("extra in persistent", sorted((cb - ca).items())[:40])
