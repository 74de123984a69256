"""GPU probe: psfm_connect with the fused persistent loop (flow_check inside) vs per-frame launches; occ output check."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.utils import flow_check_device
from point_trajectory.trajectory import run_connect, _result_to_host

cases = [(7, 48, 64, 2, 3), (9, 45, 70, 1, 5), (12, 50, 66, 3, 7), (10, 64, 128, 2, 9), (21, 200, 300, 2, 11), (31, 270, 480, 1, 12), (4, 33, 47, 2, 13), (2, 40, 60, 2, 14)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    cases = [(101, 1080, 1920, 2, 0)]
ctx = _hip.context()
L = _hip.lib()
ok_all = True
for (T, H, W, r, seed) in cases:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=seed, sigma=0.3 if H < 1000 else 0.05, n_occluders=2, stride2=False)
    res = {}
    for mode in (1, 2):
        ctx.set_chain_mode(mode)
        ts = []
        for it in range(14):
            torch.cuda.synchronize(); t0 = time.time()
            info = run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r, return_device=True)
            torch.cuda.synchronize(); t1 = time.time()
            ts.append((t1 - t0) * 1e3)
        res[mode] = (_result_to_host(ctx, info), float(np.median(ts[2:])), info.chain_mode)
    A, B = res[1][0], res[2][0]
    same = (len(A) == len(B) and np.array_equal(A.birth, B.birth) and np.array_equal(A.length, B.length) and np.array_equal(A.xy, B.xy))
    # occ output through the caller's buffer (only fused when H*W % 128 == 0)
    _, occ_ref = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    occ_out = torch.full((T - 1, H, W), 7, dtype=torch.uint8, device="cuda")
    info = _hip.TrackInfo()
    _hip.check(L.psfm_connect(ctx.handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]), None, None, T - 1, H, W, 1.0, r,
                              _hip.ptr(occ_out), None, ctypes.byref(info), _hip.current_stream_ptr()))
    occ_same = bool(torch.equal(occ_out, occ_ref))
    ok_all &= bool(same) and occ_same
    print("case T=%d %dx%d r=%d: n_traj %d/%d identical=%s  occ out equal=%s (mode %d)  per-frame %.3f ms (mode %d)  fused %.3f ms (mode %d)" % (
        T, H, W, r, len(A), len(B), same, occ_same, info.chain_mode, res[1][1], res[1][2], res[2][1], res[2][2]))
ctx.set_chain_mode(0)
print("ALL IDENTICAL" if ok_all else "MISMATCH")
