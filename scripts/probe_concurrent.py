"""Debug: several sequences in flight on one GPU with the persistent loop -- per-call times and which path ran."""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory import _hip
H, W, T, R = 1080, 1920, 101, 2
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = _hip.lib()
data = [psfm_synth.synth_sequence_torch(T, H, W, seed=100 + k, sigma=0.05, n_occluders=2, stride2=False) for k in range(n_seq)]
ctxs = [_hip.Context(0) for _ in range(n_seq)]
for c in ctxs: c.set_chain_mode(mode)
streams = [torch.cuda.Stream() for _ in range(n_seq)]
log = [[] for _ in range(n_seq)]
def worker(k, n):
    torch.cuda.set_device(0)
    sp = ctypes.c_void_p(streams[k].cuda_stream)
    info = _hip.TrackInfo()
    d = data[k]
    for _ in range(n):
        t0 = time.perf_counter()
        _hip.check(L.psfm_connect(ctxs[k].handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]), None, None, T - 1, H, W, 1.0, R, None, None, ctypes.byref(info), sp))
        log[k].append((round(1e3 * (time.perf_counter() - t0), 2), info.chain_mode))
for reps in (2, 4):
    for l in log: l.clear()
    ths = [threading.Thread(target=worker, args=(k, reps)) for k in range(n_seq)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("reps %d: %.2f ms per sequence" % (reps, 1e3 * dt / (reps * n_seq)))
    for k in range(n_seq): print("   thread", k, log[k])
