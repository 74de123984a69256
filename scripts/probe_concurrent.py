"""Experiment: two independent sequences processed concurrently on ONE GPU (two contexts, two streams, two host
threads) vs back to back -- how much idle time inside the phase-serialised chain_step launches can be filled?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import ctypes, torch
import psfm_synth
from point_trajectory import _hip

H, W, T, r = 1080, 1920, 101, 2
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data = [psfm_synth.synth_sequence_torch(T, H, W, seed=k, sigma=0.05, n_occluders=2, stride2=False) for k in range(nseq)]
ctxs = [_hip.Context(0) for _ in range(nseq)]
streams = [torch.cuda.Stream() for _ in range(nseq)]
L = _hip.lib()

def run(k, reps):
    d, ctx, st = data[k], ctxs[k], streams[k]
    occ = torch.empty((T - 1, H, W), dtype=torch.uint8, device="cuda")
    info = _hip.TrackInfo()
    sp = ctypes.c_void_p(st.cuda_stream)
    for _ in range(reps):
        _hip.check(L.psfm_flow_check(ctx.handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]), T - 1, H, W, 1.0, _hip.ptr(occ), None, sp))
        _hip.check(L.psfm_track(ctx.handle, _hip.ptr(d["flows_f"]), _hip.ptr(occ), None, None, T - 1, H, W, r, ctypes.byref(info), sp))
    return info.n_points

for k in range(nseq): run(k, 2)
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
pts = sum(run(k, reps) for k in range(nseq))
torch.cuda.synchronize()
t_seq = time.perf_counter() - t0
ths = [threading.Thread(target=run, args=(k, reps)) for k in range(nseq)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
torch.cuda.synchronize()
t_con = time.perf_counter() - t0
print("sequences %d  back-to-back %.2f ms/seq   concurrent %.2f ms/seq  (%.2fx)" % (nseq, 1e3 * t_seq / (reps * nseq), 1e3 * t_con / (reps * nseq), t_seq / t_con))
