"""Several SMALL sequences in flight on one GPU through host threads + streams + contexts (what point_trajectory.batch does):
is a BASELINE configs[0] / [2] / [4]-shaped sequence launch-bound on the host, and how much does concurrency buy?
    python scripts/probe_concurrent_small.py [out.json]
Prints ms per sequence and points/s for 1 / 2 / 4 / 8 / 16 sequences in flight (the cheap measurement)."""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory import _hip

L = _hip.lib()
SHAPES = [("configs[0] DAVIS 480x854 r4 track", 480, 854, 50, 4, False, 1.0),
          ("configs[2] Sintel 436x1024 r2 optimize", 436, 1024, 50, 2, True, 1.0),
          ("configs[4] ScanNet 480x640 r1 optimize (200 frames)", 480, 640, 200, 1, True, 3.0)]
out = []
for label, H, W, T, R, opt, thres in SHAPES:
    NMAX = 16
    data = [psfm_synth.synth_sequence_torch(T, H, W, seed=100 + k, sigma=0.05, n_occluders=2, stride2=opt) for k in range(NMAX)]
    for n_seq in (1, 2, 4, 8, 16):
        ctxs = [_hip.Context(0) for _ in range(n_seq)]
        for c in ctxs:
            c.set_chain_mode(1 if n_seq > 1 else 0)      # (as batch.connect_sequences: launches only when several are in flight)
        streams = [torch.cuda.Stream() for _ in range(n_seq)]
        pts = [0] * n_seq

        def worker(k, n):
            torch.cuda.set_device(0)
            sp = ctypes.c_void_p(streams[k].cuda_stream)
            info = _hip.TrackInfo()
            d = data[k]
            for _ in range(n):
                _hip.check(L.psfm_connect(ctxs[k].handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]),
                                          _hip.ptr(d["flows_f2"]) if opt else None, _hip.ptr(d["flows_b2"]) if opt else None,
                                          T - 1, H, W, thres, R, None, None, ctypes.byref(info), sp))
            pts[k] = int(info.n_points)

        reps = 6
        for phase in (2, reps):
            ths = [threading.Thread(target=worker, args=(k, phase)) for k in range(n_seq)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row = {"shape": label, "in_flight": n_seq, "ms_per_sequence": 1e3 * dt / (reps * n_seq), "points_per_s": sum(pts) * reps / dt}
        out.append(row)
        print("%-52s in flight %2d: %7.3f ms per sequence, %.3e points/s" % (label, n_seq, row["ms_per_sequence"], row["points_per_s"]), flush=True)
        for c in ctxs:
            c.close()
    del data
    torch.cuda.empty_cache()
if len(sys.argv) > 1 and sys.argv[1]:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
