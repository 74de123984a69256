"""Sequences whose solves reject steps (psfm_synth.REALISTIC / HARD), several in flight on one GPU: resident solves side by side on
shares of the device's block slots (psfm_ctx_set_resident_budget) against the launch chain (what concurrent sequences used before)
and against one sequence at a time with the device to itself.
    python scripts/probe_resident_budget.py [out.json] [shape ...]"""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
import psfm_synth
from point_trajectory import _hip

L = _hip.lib()
SHAPES = {"sintel": ("configs[2] Sintel 436x1024 r2 optimize, realistic flows", 436, 1024, 50, 2, 1.0),
          "scannet": ("configs[4] ScanNet 480x640 r1 optimize (100 frames), realistic flows", 480, 640, 100, 1, 3.0),
          "davis": ("DAVIS 480x854 r4 optimize, realistic flows", 480, 854, 50, 4, 1.0)}
which = sys.argv[2:] or ["sintel", "scannet", "davis"]
out = []
for key in which:
    label, H, W, T, R, thres = SHAPES[key]
    NMAX = 8
    data = [psfm_synth.synth_realistic_torch(T, H, W, seed=300 + k, stride2=True, **psfm_synth.REALISTIC) for k in range(NMAX)]
    cap = _hip.context().resident_capacity()
    for n_thr, budget, mode in ((1, 0, 0), (2, 0, 1), (4, 0, 1), (2, cap // 2, 1), (4, cap // 4, 1), (8, cap // 8, 1), (4, cap // 2, 1)):
        ctxs = [_hip.Context(0) for _ in range(n_thr)]
        for c in ctxs:
            c.set_chain_mode(mode)
            c.set_resident_budget(budget)
        streams = [torch.cuda.Stream() for _ in range(n_thr)]
        res = [None] * n_thr

        def worker(k, n):
            torch.cuda.set_device(0)
            sp = ctypes.c_void_p(streams[k].cuda_stream)
            info = _hip.TrackInfo()
            for j in range(n):
                d = data[(k + j * n_thr) % NMAX]
                _hip.check(L.psfm_connect(ctxs[k].handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]), _hip.ptr(d["flows_f2"]), _hip.ptr(d["flows_b2"]),
                                          T - 1, H, W, thres, R, None, None, ctypes.byref(info), sp))
            res[k] = (int(info.n_points), int(info.solver_iterations), ctxs[k].solver_counters())

        reps = 3
        for phase in (1, reps):
            ths = [threading.Thread(target=worker, args=(k, phase)) for k in range(n_thr)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        cnt = res[0][2]
        row = {"shape": label, "threads": n_thr, "resident_budget": budget, "chain_mode": mode, "ms_per_sequence": 1e3 * dt / (reps * n_thr),
               "iterations_per_sequence": res[0][1], "counters_thread0": cnt}
        out.append(row)
        print("%-70s threads %d budget %3d: %8.3f ms per sequence (%d iterations; thread 0: %d resident launches, %d gave up, %d iteration launches, chain %d fused %d+%d)"
              % (label, n_thr, budget, row["ms_per_sequence"], res[0][1], cnt["resident_launches"], cnt["resident_giveups"], cnt["iteration_launches"],
                 cnt["chain"], cnt["fused"], cnt["fused_redone"]), flush=True)
        for c in ctxs:
            c.close()
    del data
    torch.cuda.empty_cache()
if len(sys.argv) > 1 and sys.argv[1]:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
