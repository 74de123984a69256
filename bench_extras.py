"""bench_extras.py -- everything bench.py measures OUTSIDE its timed region: the other BASELINE.json shapes, the path-consistency
solver on three flow distributions, B sequences per launch, one sequence over all ranks, disk to disk.  bench.py runs `run_all`
after the headline step unless --no-extras, writes the full record to bench_extras.json (beside bench.py and, when it exists,
under gpurun_out/) and carries `summary()` of it -- a few numbers per figure -- in its one JSON line."""
import json
import os
import sys
import time

from bench_common import (ROOT, H, W, N_FRAMES, RATIO, THRES, HBM_PEAK_GBS, VALU_PEAK_GWIPS, REFERENCE_SOLVER_THREADS,   # noqa: F401
                          quiet_gc, source_sha16, replayed, cpu_port_timed, cpu_threads_wide)


def concurrent_sequences(n_seq, n_frames, reps=4):
    """Throughput with several independent sequences in flight on ONE GPU (separate psfm contexts and HIP streams,
    one host thread each -- point_trajectory.batch): the single-sequence recurrence is latency-bound, concurrency
    fills its idle memory / SIMD time.  Outside the timed region; the headline `value` is one sequence at a time."""
    import ctypes
    import threading
    import torch
    import psfm_synth
    from point_trajectory import _hip
    L = _hip.lib()
    data = [psfm_synth.synth_sequence_torch(n_frames, H, W, seed=100 + k, sigma=0.05, n_occluders=2, stride2=False)
            for k in range(n_seq)]
    pts = [0] * n_seq
    ctxs = [_hip.Context(torch.cuda.current_device()) for _ in range(n_seq)]
    streams = [torch.cuda.Stream() for _ in range(n_seq)]

    def run(k, n):
        ctx = ctxs[k]
        sp = ctypes.c_void_p(streams[k].cuda_stream)
        info = _hip.TrackInfo()
        d = data[k]
        for _ in range(n):
            _hip.check(L.psfm_connect(ctx.handle, _hip.ptr(d["flows_f"]), _hip.ptr(d["flows_b"]), None, None, n_frames - 1,
                                      H, W, THRES, RATIO, None, None, ctypes.byref(info), sp))
        pts[k] = int(info.n_points)

    dev = torch.cuda.current_device()

    def worker(k, n):
        torch.cuda.set_device(dev)
        run(k, n)

    for phase_reps in (2, reps):      # first pass warms the per-context workspaces
        ths = [threading.Thread(target=worker, args=(k, phase_reps)) for k in range(n_seq)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    return {"sequences_in_flight": n_seq, "ms_per_sequence": 1e3 * dt / (reps * n_seq),
            "trajectory_points_per_s": sum(pts) * reps / dt}


def single_sequence_sharded(dev, rank, world, frames, reps=2, flows_dist=None, label="configs[3] shape"):
    """BASELINE.json configs[3]: ONE 1080p sequence with the full path-consistency optimize over all ranks, exactly
    (psfm_dist.connect_sharded: flow_check by frame pair + all-gather, tracks by birth row band, one all-reduce(max) of
    the blocked map per frame, solver sums all-gathered per launch; RCCL when world > 1).  Every rank synthesises the same
    seeded sequence.  Beside it, on rank 0, the same sequence through the one-GPU product call (psfm_connect)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import psfm_dist
    import psfm_synth
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    from point_trajectory.trajectory import run_connect
    torch.cuda.set_device(dev)
    d = psfm_synth.synth_sequence_torch(frames, H, W, seed=1, stride2=True, device=dev, **(flows_dist or dict(sigma=0.05, n_occluders=2)))
    comm = psfm_dist.TorchComm()
    eng = HipShardEngine()

    # The four stacks are OWNED by frame-pair slices (Stage A's shards): a rank keeps 1 / world of the sequence and
    # receives Stage B's frames by broadcast from their owners, two frames ahead (psfm_dist.FrameWindow).  (The generator makes
    # the whole sequence on every rank first -- synthetic data has no files to read a slice of; the rest is freed here.)
    n_total = frames - 1
    if world > 1:
        lo, hi = psfm_dist.shard_range(n_total, rank, world)
        lo2, hi2 = psfm_dist.shard_range(n_total - 1, rank, world)
        d = {"flows_f": d["flows_f"][lo:hi].clone(), "flows_b": d["flows_b"][lo:hi].clone(),
             "flows_f2": d["flows_f2"][lo2:hi2].clone(), "flows_b2": d["flows_b2"][lo2:hi2].clone()}
        torch.cuda.empty_cache()

    def once():
        return psfm_dist.connect_sharded(eng, d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO,
                                         flow_check_slice, comm=comm, keep_on_device=True,   # result left in HBM, like the headline step
                                         n_flows_total=n_total if world > 1 else None)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    part = once()
    sync()
    quiet_gc()
    t0 = time.perf_counter()
    for _ in range(reps):
        part = once()
    sync()
    dt = (time.perf_counter() - t0) / reps
    dt, pts = psfm_dist.reduce_totals(dt, float(part["n_points_local"]), device=dev)
    out = {"mode": "single-sequence", "world_size": dist.get_world_size() if world > 1 else 1,
           "backend": (dist.get_backend() if world > 1 else None),
           "workload": "%s: synthetic %dx(1920x1080) flow pairs + stride-2 stacks, sample_ratio=2, flow_check x2 + "
                       "track_optimize, ONE sequence over %d rank(s)" % (label, frames - 1, world), "flows": dict(flows_dist or dict(sigma=0.05, n_occluders=2)),
           "partition": "flow_check by frame pair (all-gather of bit-packed maps); tracks by birth row band; per frame one "
                        "all-reduce(max) of %d bytes; per fused solve one all-gather of k x 13 doubles" % (((W + RATIO - 1) // RATIO) * ((H + RATIO - 1) // RATIO) + 1),
           "ms_per_sequence": 1e3 * dt, "trajectory_points_per_s": pts / dt, "points": int(pts), "trajectories": int(part["n_traj"]),
           "solves": part["n_solves"], "trust_region_iterations": part["solver_iterations"],
           "solver_counters": dict(eng.counters), "local_trajectories_rank0": int(part["ids"].numel())}
    if world > 1:     # how the solves that reject steps ran across the ranks: cross-rank resident launches (psfm_shard_solve_peer) or the exchange form
        out["cross_rank_solve"] = {"launches": int(eng.counters.get("peer", 0)), "gave_up": int(eng.counters.get("peer_redone", 0)),
                                   "refused": getattr(eng, "peer_refused", None)}
    lc = eng.ctx.solver_counters()
    out["solver_launches"] = {k: lc[k] for k in ("resident_launches", "resident_giveups", "iteration_launches")}
    if world == 1:
        # one rank exchanges nothing: windows whose solves reject steps take the one-GPU call's forms (psfm_shard_solve_local: resident
        # solves).  What several ranks pay for the same solves -- export -> exchange -> control per trust-region iteration, rounds
        # enqueued ahead -- is this engine with PSFM_SHARD_LOCAL=0, timed beside it
        out["rejecting_solves"] = "one rank: psfm_shard_solve_local / _redo_local (the one-GPU call's resident solves)"
        if flows_dist is not None:
            os.environ["PSFM_SHARD_LOCAL"] = "0"
            try:
                once()
                sync()
                t0 = time.perf_counter()
                once()
                sync()
                out["ms_per_sequence_exchange_form"] = 1e3 * (time.perf_counter() - t0)
            finally:
                del os.environ["PSFM_SHARD_LOCAL"]
    out["flow_stacks_per_rank_GB"] = sum(int(v.numel()) * 4 for v in d.values()) / 1e9
    out["flow_ownership"] = "frame-pair slices + per-frame broadcast (psfm_dist.FrameWindow)" if world > 1 else "whole sequence (one rank)"
    if rank == 0 and world == 1:   # the one-GPU product call on the same tensors: time and counts
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO, return_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO, return_device=True)
        torch.cuda.synchronize()
        out["one_gpu_psfm_connect_ms_per_sequence"] = 1e3 * (time.perf_counter() - t0)
        out["ratio_to_one_gpu_call"] = out["ms_per_sequence"] / out["one_gpu_psfm_connect_ms_per_sequence"]
        out["counts_equal_one_gpu"] = bool(int(info.n_traj) == int(part["n_traj"]) and int(info.n_points) == int(pts))
    return out


def guarded(fn, timeout_s):
    """fn() on a helper thread with a deadline: (result, hung).  A collective that never completes (a rank that died)
    must not take the headline line with it."""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as e:     # noqa: BLE001
            box["r"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return {"error": "no result after %d s" % timeout_s}, True
    return box["r"], False


HARD_LABEL = "headline shape, hard flows (sigma 0.3, 5 % occluders)"


def sharded_children(rank, world, frames, timeout_s=None):
    """`single_sequence_sharded` twice (BASELINE configs[3]; the headline shape on hard flows) over all ranks, each rank's share in a CHILD
    process of that rank (this file run with --sharded-child; the children form a process group of their own on a fresh port).  The
    cross-GPU forms of the sharded engine -- RCCL exchanges, and granules written into other ranks' memory through IPC mappings -- have
    only run on one-GPU proxies; a device fault there ends the process it happens in, and that must be a child, never the process that
    holds the headline figure.  Returns (single, single_hard): rank 0's child's record, or {"error": ...}."""
    import socket
    import subprocess
    import torch.distributed as dist
    timeout_s = timeout_s or int(os.environ.get("PSFM_BENCH_CHILD_TIMEOUT", "240"))
    out = []
    for hard in (False, True):
        port = [None]
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port[0] = sk.getsockname()[1]
        dist.broadcast_object_list(port, src=0)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}    # the children's rank 0 hosts their store itself
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--frames", str(101 if hard else frames)] + (["--hard"] if hard else [])
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                res = {"error": "rank %d's child left with code %d: %s" % (rank, r.returncode, (r.stderr or r.stdout)[-300:])}
            elif rank == 0:
                res = json.loads(r.stdout.strip().splitlines()[-1])
                res["ran_in"] = "one child process per rank (a process group of their own)"
            else:
                res = {}
        except subprocess.TimeoutExpired:
            res = {"error": "rank %d's child: no result after %d s" % (rank, timeout_s)}
        except Exception as e:     # noqa: BLE001
            res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        every = [None] * world
        dist.all_gather_object(every, res.get("error"))
        bad = [e for e in every if e]
        if bad and "error" not in res:
            res = {"error": bad[0]}
        out.append(res)
        if bad:                     # the second run would meet the same end
            out.append({"error": "skipped: " + bad[0]})
            break
    return out[0], out[1]


def _sharded_child_main(argv):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--sharded-child", action="store_true")
    ap.add_argument("--frames", type=int, default=401)
    ap.add_argument("--hard", action="store_true")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ[k]) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"))
    dryrun = os.environ.get("PSFM_BENCH_DRYRUN_ONE_GPU", "0") == "1"      # as in bench.py: every rank on cuda:0, gloo
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dryrun:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=dev)
    if os.environ.get("PSFM_BENCH_CHILD_ABORT") == str(rank):     # tests: this rank's child dies the way a device fault ends a process
        os.abort()
    if a.hard:
        import psfm_synth
        rec = single_sequence_sharded(dev, rank, world, a.frames, reps=1, flows_dist=psfm_synth.HARD, label=HARD_LABEL)
    else:
        rec = single_sequence_sharded(dev, rank, world, a.frames)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    _, stuck = guarded(lambda: (dist.barrier(), torch.cuda.synchronize(), dist.destroy_process_group()), 60)
    sys.stdout.flush()
    os._exit(0)


def solver_roofline(R, prof, cnt, h, w, n_flows, ratio=RATIO):
    """SURVEY 8(d) algorithmic bytes of the track_optimize kernels / their HIP-event launch time / 8 TB/s.
    Per solve of frame f (N3 = tracks with three buffered points, k = trust-region iterations of that solve, P = H*W):
      pc_prepare = 2 min(8P, 32 N3) + min(P, 4 N3) + 16 N3 + 40 N3        (refs + scale: flow01, flow02, occ02 at p0)
      pc_solve   = k [ min(8P, 32 N3) + 72 N3 + 32 N3 ]                    (flow12 taps, state in, candidate out)
    The fused solve (psfm_pc_fused_kernel) is ONE launch per solve doing both, so its bytes are their sum; the chain
    step moves min(8P, 32A) + min(P, 4A) + 16A + 16A + A per frame (A = tracks alive at the step)."""
    import numpy as np
    P = float(h * w)
    birth = R.birth.astype(np.int64)
    last = birth + R.length - 1
    out = {}
    # N3 of the solve at loop index f (times f-1, f, f+1): born <= f-1, still there at f+1
    tb = np.bincount(birth, minlength=n_flows + 3).cumsum()           # tracks born <= t
    tl = np.bincount(last, minlength=n_flows + 3).cumsum()            # tracks whose last time <= t
    its = [s["iterations"] for s in R.solve_stats]
    frames = list(range(1, n_flows))
    alive_steps = float(R.n_points - int((last == n_flows).sum()))
    A = alive_steps / n_flows
    cb = min(8 * P, 32 * A) + min(P, 4 * A) + 16 * A + 16 * A + A
    ch = prof["chain_step"]
    merged = ch["launches"] * 2 < n_flows          # track_optimize ran the merged frame kernel: chain step inside the solve's launch
    if len(its) == len(frames) and prof["solver"]["launches"] > 0:
        tot = 0.0
        for f, k in zip(frames, its):
            n3 = float(tb[f - 1] - tl[f])          # born by f-1, last time >= f+1
            prep = 2 * min(8 * P, 32 * n3) + min(P, 4 * n3) + 16 * n3 + 40 * n3
            solve = k * (min(8 * P, 32 * n3) + 72 * n3 + 32 * n3)
            tot += prep + solve
        us = 1e3 * prof["solver"]["total_ms"] / prof["solver"]["launches"]
        per_solve = tot / len(frames) + (cb if merged else 0.0)
        fused = cnt["fused"] + cnt["fused_redone"] > cnt["chain"]
        # which kernels the non-fused windows ran: counted by the library (psfm_solver_launches), not assumed
        n_res, n_itl = int(cnt.get("resident_launches", 0)), int(cnt.get("iteration_launches", 0))
        resident = (not fused) and n_res > 0 and n_itl < n_res * 4
        name = ("psfm_seq_kernel = the frame kernel, device-paced (ONE launch per frame: chain step + fused solve)" if merged else
                "psfm_pc_fused_kernel (one launch per solve)") if fused else \
            ("psfm_pc_resident_kernel (one launch per solve: iteration 0, the trust-region loop with the tracks' state on chip, write-back)"
             if resident else "the launch chain: psfm_pc_init_kernel + one psfm_pc_iter_kernel launch per trust-region iteration "
                              "(%d iteration launches, %d resident launches of which %d gave up)" % (n_itl, n_res, int(cnt.get("resident_giveups", 0))))
        # What bounds these launches is f64 VALU issue, not bandwidth: wave-instructions per launch from the
        # PMC pass of the same kernel (SQ_INSTS_VALU, profiles/solver_valu.json: replayed, scaled by this run's track-iterations)
        # against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 f64 instruction; the SURVEY 8(d) byte MODEL and the PMC traffic ride along.
        entry = {
            "kernel": name, "bound": "valu-issue", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GWIPS,
            "avg_launch_us": us, "launches_timed": int(prof["solver"]["launches"]), "avg_iterations": float(np.mean(its)),
            "track_iterations_per_launch": float(np.mean([k * float(tb[f - 1] - tl[f]) for f, k in zip(frames, its)])),
            "hbm_model": {"bytes_per_launch": per_solve,
                          "bytes_breakdown": {"pc_prepare + pc_solve (SURVEY 8d)": tot / len(frames), "chain_step": cb if merged else 0.0},
                          "model_GBs": per_solve / (us * 1e-6) / 1e9, "model_frac_of_hbm_peak": per_solve / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "note": "SURVEY 8(d) prices one re-read of the state per trust-region iteration; the fused kernel keeps the "
                                  "iterate in registers, so this is NOT its HBM utilisation (see traffic)"},
            "note": "the timed launches include the few that overlap a flow_check chunk of the side stream and the retries"}
        vfile = os.path.join(ROOT, "profiles", "solver_valu.json")
        vall, vprov = replayed(vfile)
        if fused and vall:
            try:
                v = vall
                shape_key = "%dx%dx%d" % (h, w, ratio)
                shape_traffic = (v.get("hbm_bytes_per_launch_by_shape") or {}).get(shape_key)
                if shape_traffic is None and (h, w) == (1080, 1920):
                    shape_traffic = v.get("hbm_bytes_per_launch")
                wi = (v["valu_per_wave_per_iteration"] * entry["avg_iterations"] + v["valu_per_wave_fixed"]) * \
                     (entry["track_iterations_per_launch"] / max(entry["avg_iterations"], 1e-9)) / 64.0
                entry.update({"achieved": wi / (us * 1e-6) / 1e9, "frac": wi / (us * 1e-6) / 1e9 / VALU_PEAK_GWIPS,
                              "valu_wave_instructions_per_launch": wi,
                              "valu_source": dict(vprov, what="PMC SQ_INSTS_VALU of the frame kernel, scaled by this run's tracks x iterations; "
                                                               + str(v.get("source", ""))[:200]),
                              # (per shape: the PMC passes run scripts/probe_solver.py on 1080p, configs[2]'s and configs[4]'s shapes)
                              "traffic": shape_traffic if merged else None,
                              "traffic_source": (v.get("traffic_source") if (h, w) == (1080, 1920) else
                                                 "profiles/solver_valu.json hbm_bytes_per_launch_by_shape[%r]: the same fabric-side "
                                                 "counters on scripts/probe_solver.py at this shape (clean flows)" % shape_key)
                              if merged and shape_traffic else "not measured for this shape"})
                if entry["traffic"]:
                    entry["frac_physical"] = entry["traffic"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        if not fused and vall and resident:       # (the replayed PMC figures were measured on the resident form)
            try:
                v = vall["chain"]
                wi = v["valu_per_track_iteration"] * entry["track_iterations_per_launch"] / 64.0 + \
                     v["valu_per_wave_per_iteration_fixed"] * v["waves"] * entry["avg_iterations"]
                entry.update({"achieved": wi / (us * 1e-6) / 1e9, "frac": wi / (us * 1e-6) / 1e9 / VALU_PEAK_GWIPS,
                              "valu_wave_instructions_per_launch": wi,
                              "valu_source": dict(vprov, what="'chain': " + v["source"][:300])})
            except Exception:
                pass
        if "frac" not in entry:
            entry.update({"achieved": None, "frac": None})
        out["frame_kernel" if merged else "solver"] = entry
    if ch["launches"] > 0 and not merged:
        us = 1e3 * ch["total_ms"] / ch["launches"]
        out["chain_step"] = {"kernel": "psfm_chain_step_kernel<R, OPT>", "bound": "hbm", "bytes_per_launch": cb, "avg_launch_us": us,
                             "achieved": cb / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": cb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_alive_tracks": A}
    fc = prof["flow_check"]
    if fc["launches"] > 0:
        out["flow_check_side_stream_ms"] = fc["total_ms"]
    out["finalize_ms"] = prof["finalize"]["total_ms"]
    return out


def stream_ceilings(dev):
    """Empirical streaming rates of THIS device (context for the roofline fractions, which are quoted against the
    8 TB/s spec): device-to-device copy (read + write bytes), read-only reduction, fill; 1 GiB fp32 buffers, torch kernels."""
    import torch
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)

    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    out = {"copy_GBs": 2 * n * 4 / timed(lambda: b.copy_(a)) / 1e9, "read_GBs": n * 4 / timed(lambda: a.sum()) / 1e9,
           "fill_GBs": n * 4 / timed(lambda: b.fill_(1.0)) / 1e9, "buffers": "2 x 1 GiB fp32, torch kernels, HIP events"}
    del a, b
    torch.cuda.empty_cache()
    return out


def secondary_track_optimize(ctx, h=436, w=1024, t=50, r=2, seed=2, k=6, label="configs[2] shape", dist=None, thres=THRES, n=5):
    """track_optimize (chaining + Ceres-compatible path-consistency solve) on a synthetic sequence -- by default a
    stand-in of configs[2] (Sintel alley_1 shape: 436x1024, 50 frames, sample_ratio 2): GPU time per sequence and the
    CPU oracle on the first k flows of the same tensors, with the parity of those flows checked on the spot."""
    import numpy as np
    import torch
    import psfm_synth
    from oracle import oracle as orc
    from point_trajectory.utils import flow_check_device
    from point_trajectory.trajectory import run_track, _result_to_host
    dist = dist or dict(sigma=0.05, n_occluders=2)
    if dist.get("realistic"):      # psfm_synth.REALISTIC: layers with true (dis)occlusion, correlated flow error, outlier blobs
        d = psfm_synth.synth_realistic_torch(t, h, w, seed=seed, stride2=True, device="cuda", **{k_: v for k_, v in dist.items() if k_ != "realistic"})
    else:
        d = psfm_synth.synth_sequence_torch(t, h, w, seed=seed, stride2=True, device="cuda", **dist)
    ctx.set_profiling(0)

    from point_trajectory.trajectory import run_connect

    def step():
        return run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], thres, r, return_device=True)

    for _ in range(2):
        info = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        info = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    # ---- per-kernel roofline of this path (HIP events on every launch, one more pass) ----
    ctx.set_profiling(1)
    info = step()
    torch.cuda.synchronize()
    pr = ctx.profile()
    ctx.set_profiling(0)
    cnt = ctx.solver_counters()
    Rh = _result_to_host(ctx, info)
    roof = solver_roofline(Rh, pr, cnt, h, w, t - 1, r)
    info_stats = list(Rh.solve_stats)
    del Rh
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], thres)
    _, occ2 = flow_check_device(d["flows_f2"], d["flows_b2"], thres)
    ff, f2 = list(d["flows_f"][:k].cpu().numpy()), list(d["flows_f2"][:k - 1].cpu().numpy())
    oo, o2 = list(occ[:k].cpu().numpy()), list(occ2[:k - 1].cpu().numpy())
    Rc, cpu = cpu_port_timed(lambda: orc.track_optimize(ff, f2, oo, o2, r), lambda R_: R_.n_points)
    Rg = _result_to_host(ctx, run_track(d["flows_f"][:k], occ[:k], d["flows_f2"][:k - 1], occ2[:k - 1], r, return_device=True))
    same = bool(np.array_equal(Rg.birth, Rc.birth) and np.array_equal(Rg.length, Rc.length))
    rej = int(sum(s_["iterations"] - s_["successful_steps"] for s_ in info_stats)) if info_stats else None
    occl = {"stride1": float(occ.float().mean()), "stride2": float(occ2.float().mean())}
    its = [s_["iterations"] for s_ in info_stats]
    return {"workload": "%s: synthetic %dx%d x %d frames, sample_ratio=%d, flow_check_thres %.1f, flow_check x2 + track_optimize"
                        % (label, h, w, t, r, thres),
            "flows": dict(dist), "rejected_steps": rej, "chain_mode": int(info.chain_mode), "occluded_fraction": occl,
            "iterations_per_solve": {"mean": float(np.mean(its)) if its else None, "min": int(min(its)) if its else None,
                                     "max": int(max(its)) if its else None,
                                     "solves_with_a_rejected_step": int(sum(1 for s_ in info_stats if s_["iterations"] > s_["successful_steps"] + 1))},
            "mean_trajectory_length": float(info.n_points) / max(int(info.n_traj), 1),
            "ms_per_sequence": ms, "trajectory_points_per_s": info.n_points / (ms * 1e-3), "points": int(info.n_points),
            "solves": int(info.n_solves), "trust_region_iterations": int(info.solver_iterations),
            "solver_counters": cnt, "roofline": roof,
            "cpu_port_points_per_s": cpu["threads_wide"]["points_per_s"], "cpu_port": cpu,
            "gpu_over_cpu_port": (info.n_points / (ms * 1e-3)) / cpu["threads_wide"]["points_per_s"],
            "gpu_over_cpu_port_8_threads": (info.n_points / (ms * 1e-3)) / cpu["threads_8"]["points_per_s"],
            "cpu_port_sample": "first %d flows: %d threads %.2f s, %d threads (the reference's solver count) %.2f s"
                               % (k, cpu["threads_wide"]["threads"], cpu["threads_wide"]["seconds"], cpu["threads_8"]["threads"], cpu["threads_8"]["seconds"]),
            "parity_first_flows": {"ids_lengths_equal": same,
                                   "max_abs_dxy_px": float(np.abs(Rg.xy - Rc.xy).max()) if same else None,
                                   "tolerance_px": 1e-4}}


def secondary_track(ctx, h, w, t, r, seed, label, thres=THRES, n=10):
    """track only (flow_check + chaining + occlusion + ids: --skip_path_consistency) on a synthetic sequence: GPU time per sequence,
    the chain step's roofline from HIP events on every launch, and the WHOLE sequence against the CPU oracle."""
    import numpy as np
    import torch
    import psfm_synth
    from oracle import oracle as orc
    from point_trajectory.trajectory import run_connect, _result_to_host
    d = psfm_synth.synth_sequence_torch(t, h, w, seed=seed, sigma=0.05, n_occluders=2, stride2=False, device="cuda")
    ctx.set_profiling(0)

    def step():
        return run_connect(d["flows_f"], d["flows_b"], None, None, thres, r, return_device=True)

    for _ in range(2):
        info = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        info = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    ctx.set_profiling(1)
    info = step()
    torch.cuda.synchronize()
    pr = ctx.profile()
    ctx.set_profiling(0)
    Rg = _result_to_host(ctx, info)
    n_flows = t - 1
    P = float(h * w)
    last = Rg.birth.astype(np.int64) + Rg.length - 1
    A = float(Rg.n_points - int((last == n_flows).sum())) / n_flows
    cb = min(8 * P, 32 * A) + min(P, 4 * A) + 16 * A + 16 * A + A
    persistent = int(info.chain_mode) == 2
    ch = pr["chain_step"]
    us = 1e3 * ch["total_ms"] / max(ch["launches"], 1) / (n_flows if persistent else 1)
    fused = persistent and pr["flow_check"]["launches"] == 0
    step_bytes = cb + (17.0 * P if fused else 0.0)
    hf, hb = list(d["flows_f"].cpu().numpy()), list(d["flows_b"].cpu().numpy())

    def cpu_run():
        _, occ = orc.flow_check(hf, hb, thres)
        return orc.track(hf, occ, r)

    Rc, cpu = cpu_port_timed(cpu_run, lambda R_: R_.n_points)
    same = bool(Rg.birth.shape == Rc.birth.shape and np.array_equal(Rg.birth, Rc.birth) and np.array_equal(Rg.length, Rc.length))
    return {"workload": "%s: synthetic %dx%d x %d frames, sample_ratio=%d, flow_check_thres %.1f, flow_check + track (no path consistency)"
                        % (label, h, w, t, r, thres),
            "ms_per_sequence": ms, "trajectory_points_per_s": info.n_points / (ms * 1e-3), "points": int(info.n_points),
            "trajectories": int(info.n_traj), "chain_mode": int(info.chain_mode),
            "roofline": {"chain_step": {"kernel": ("psfm_chain_persist_kernel" + (" (flow_check fused in)" if fused else "")) if persistent
                                                  else "psfm_chain_step_kernel<R> (one launch per frame)",
                                        "bound": "hbm", "bytes_per_step": step_bytes, "us_per_step": us,
                                        "achieved": step_bytes / (us * 1e-6) / 1e9 if us > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": step_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else None, "avg_alive_tracks": A},
                         "flow_check_side_stream_ms": pr["flow_check"]["total_ms"], "finalize_ms": pr["finalize"]["total_ms"]},
            "cpu_port_points_per_s": cpu["threads_wide"]["points_per_s"], "cpu_port": cpu,
            "cpu_port_sample": "whole sequence: %d threads %.2f s, %d threads %.2f s" % (cpu["threads_wide"]["threads"], cpu["threads_wide"]["seconds"],
                                                                                        cpu["threads_8"]["threads"], cpu["threads_8"]["seconds"]),
            "parity": {"vs": "cpu oracle, whole sequence", "ids_lengths_equal": same,
                       "max_abs_dxy_px": float(np.abs(Rg.xy - Rc.xy).max()) if same else None}}


def secondary_batch(h, w, t, r, opt, thres, B, seed0, label, single, n=5, pmc_key=None):
    """psfm_connect_batch on B sequences of one of the small BASELINE shapes (different seeds; sequence 0 is the one `single` was
    measured on): ONE launch per frame for the whole batch, one checkpoint per window, one segmented finalize.  Time per sequence
    against `single` (one psfm_connect per sequence), the batched frame launch from HIP events with its roofline, and every
    sequence's result against its own single-sequence run (counts for all, bits for the first and the last)."""
    import numpy as np
    import torch
    import psfm_synth
    from point_trajectory import _hip
    from point_trajectory.trajectory import run_connect, run_connect_batch, _result_to_host
    data = [psfm_synth.synth_sequence_torch(t, h, w, seed=seed0 + k, sigma=0.05, n_occluders=2, stride2=opt, device="cuda") for k in range(B)]
    seqs = [(d["flows_f"], d["flows_b"], d.get("flows_f2") if opt else None, d.get("flows_b2") if opt else None) for d in data]
    for _ in range(2):
        ctxs, infos = run_connect_batch(seqs, thres, r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctxs, infos = run_connect_batch(seqs, thres, r)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    pts = int(sum(int(i.n_points) for i in infos))
    ctxs[0].set_profiling(1)
    ctxs, infos = run_connect_batch(seqs, thres, r)
    torch.cuda.synchronize()
    pr = ctxs[0].profile()
    ctxs[0].set_profiling(0)
    kind = "solver" if opt else "chain_step"
    us = 1e3 * pr[kind]["total_ms"] / max(pr[kind]["launches"], 1)
    # the same launches with every occlusion map ready before the first frame (PSFM_BATCH_FC_CHUNK=0: flow_check up front on the
    # launch stream instead of beside the frame loop on the side stream, whose bandwidth the frame launches above share)
    os.environ["PSFM_BATCH_FC_CHUNK"] = "0"
    try:
        run_connect_batch(seqs, thres, r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run_connect_batch(seqs, thres, r)
        torch.cuda.synchronize()
        ms_upfront = 1e3 * (time.perf_counter() - t0) / n
        ctxs[0].set_profiling(1)
        ctxs, infos = run_connect_batch(seqs, thres, r)
        torch.cuda.synchronize()
        pr0 = ctxs[0].profile()
        ctxs[0].set_profiling(0)
    finally:
        os.environ.pop("PSFM_BATCH_FC_CHUNK", None)
    us0 = 1e3 * pr0[kind]["total_ms"] / max(pr0[kind]["launches"], 1)
    # parity of the batching: every sequence against ONE psfm_connect of its own on the same tensors
    ctx1 = _hip.context()
    counts_equal, bits = True, {}
    keep = {0: _result_to_host(ctxs[0], infos[0]), B - 1: _result_to_host(ctxs[B - 1], infos[B - 1])}
    got = [(int(i.n_traj), int(i.n_points), int(i.solver_iterations)) for i in infos]
    for k in range(B):
        i1 = run_connect(*seqs[k], thres, r, return_device=True)
        if (int(i1.n_traj), int(i1.n_points), int(i1.solver_iterations)) != got[k]:
            counts_equal = False
        if k in keep:
            R1 = _result_to_host(ctx1, i1)
            Rb = keep[k]
            same = bool(np.array_equal(R1.birth, Rb.birth) and np.array_equal(R1.length, Rb.length))
            bits["sequence_%d" % k] = {"ids_lengths_equal": same, "max_abs_dxy_px": float(np.abs(R1.xy - Rb.xy).max()) if same else None}
    single_ms = single.get("ms_per_sequence") if isinstance(single, dict) else None
    out = {"workload": "%s x %d sequences (seeds %d..%d) through psfm_connect_batch: %dx%d x %d frames, sample_ratio=%d, %s"
                       % (label, B, seed0, seed0 + B - 1, h, w, t, r, "flow_check x2 + track_optimize" if opt else "flow_check + track"),
           "batch": B, "ms_per_batch": ms, "ms_per_sequence": ms / B, "trajectory_points_per_s": pts / (ms * 1e-3), "points": pts,
           "single_sequence_ms": single_ms, "speedup_vs_one_psfm_connect_per_sequence": (single_ms / (ms / B)) if single_ms else None,
           "modes": sorted(set(int(i.chain_mode) for i in infos)),
           "frame_launch": {"kernel": "psfm_seq_batch_kernel<R, 4> (blockIdx.y = sequence: chain step + fused solve of every sequence's next frame)" if opt
                                      else "psfm_chain_step_batch_kernel<R> (blockIdx.y = sequence)",
                            "avg_launch_us": us, "launches": int(pr[kind]["launches"]),
                            "avg_launch_us_on_ready_maps": us0, "ms_per_sequence_with_flow_check_up_front": ms_upfront / B},
           "flow_check_enqueue_ms": pr["flow_check"]["total_ms"], "finalize_ms": pr["finalize"]["total_ms"],
           "parity": {"vs": "one psfm_connect per sequence on the same tensors", "counts_equal_every_sequence": counts_equal, "bits": bits}}
    try:
        roof = single["roofline"]
        if not opt:
            cs = roof["chain_step"]
            by = cs["bytes_per_step"] * B      # (every sequence has the same shape and flow statistics: sequence 0's bytes x B)
            out["frame_launch"].update({"bound": "hbm", "bytes_per_launch": by, "achieved": by / (us0 * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": by / (us0 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "frac_beside_the_side_streams_flow_check": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                        "note": "frac: the launch on ready maps (the chain step's own bytes / its own time); in the default run "
                                                "flow_check streams 17 P bytes per pair through the same memory system meanwhile",
                                        "single_sequence_frac": cs.get("frac")})
        else:
            fk = roof["frame_kernel"]
            wi = fk["valu_wave_instructions_per_launch"] * B * (t - 2) / max(pr[kind]["launches"], 1)     # (per batched launch, spare launches included)
            out["frame_launch"].update({"bound": "valu-issue", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GWIPS,
                                        "valu_wave_instructions_per_launch": wi, "achieved": wi / (us0 * 1e-6) / 1e9,
                                        "frac": wi / (us0 * 1e-6) / 1e9 / VALU_PEAK_GWIPS,
                                        "frac_beside_the_side_streams_flow_check": wi / (us * 1e-6) / 1e9 / VALU_PEAK_GWIPS,
                                        "single_sequence_frac": fk.get("frac"),
                                        "valu_source": "sequence 0's replayed figure (profiles/solver_valu.json) x %d sequences x %d solves / launches" % (B, t - 2)})
    except Exception:      # noqa: BLE001  (the single-sequence figure failed or has no roofline)
        pass
    # the same launches under rocprofv3 (scripts/profile_round5.sh: PMC passes + kernel-trace durations of this shape and batch size),
    # replayed with their provenance: measured wave-instructions / HBM bytes per launch instead of the scaled single-sequence figure
    bp, prov = replayed(os.path.join(ROOT, "profiles", "batch_pmc.json"))
    if bp and pmc_key and pmc_key in bp:
        out["frame_launch"]["pmc"] = dict(bp[pmc_key], provenance=prov)
    return out


def end_to_end_batch(n_seq=16, h=480, w=854, t=50, r=4):
    """SURVEY 8(d)(iii) for the shapes real data has: n_seq configs[0]-shaped sequences on tmpfs (.flo) -> track.npy each, through
    point_trajectory.batch.connect_sequences -- one psfm_connect per sequence at 1 / 2 / 4 host threads, and psfm_connect_batch
    (batch = 8 per worker) at 1 / 2 threads.  sequences/s; the phases of neighbours overlap only where the threads do."""
    import shutil
    import tempfile
    import torch
    import psfm_synth
    from point_trajectory.utils import write_flo
    from point_trajectory.batch import connect_sequences
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="psfm_e2eb_", dir=base)
    try:
        fdirs, tdirs = [], []
        for k in range(n_seq):
            d = psfm_synth.synth_sequence_torch(t, h, w, seed=200 + k, sigma=0.05, n_occluders=2, stride2=False)
            fd = os.path.join(work, "seq%02d" % k, "flows")
            for name, key in (("flow_f", "flows_f"), ("flow_b", "flows_b")):
                os.makedirs(os.path.join(fd, name))
                arr = d[key].cpu().numpy()
                for i in range(t - 1):
                    write_flo(os.path.join(fd, name, "%05d.flo" % i), arr[i])
            fdirs.append(fd)
            tdirs.append(os.path.join(work, "seq%02d" % k, "traj"))
            del d
        torch.cuda.empty_cache()
        gb = n_seq * 2 * (t - 1) * h * w * 8 / 1e9
        rows = []
        for conc, batch in ((1, 1), (2, 1), (4, 1), (1, 8), (2, 8)):
            for rep in range(2):      # (second pass: warm page cache and workspaces)
                for td in tdirs:
                    shutil.rmtree(td, ignore_errors=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                connect_sequences(fdirs, tdirs, sample_ratio=r, flow_check_thres=THRES, skip_path_consistency=True, concurrency=conc,
                                  rank=0, world=1, layout="reference", batch=batch)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            rows.append({"host_threads": conc, "batch": batch, "total_s": dt, "sequences_per_s": n_seq / dt, "ms_per_sequence": 1e3 * dt / n_seq})
        size = sum(os.path.getsize(os.path.join(td, "track.npy")) for td in tdirs)
        return {"workload": "%d configs[0]-shaped sequences (%dx%d x %d frames, sample_ratio %d, track only) disk to disk on %s: %.2f GB of .flo -> "
                            "%.2f GB of track.npy (reference pickle layout), warm second pass" % (n_seq, h, w, t, r, base, gb, size / 1e9),
                "runs": rows,
                "note": "batch = 1: one main_connect_point_trajectories per sequence on `host_threads` threads (ingest / compute / write of "
                        "neighbouring sequences overlap only across threads; the GIL is released in file reads, copies and library calls); "
                        "batch = 8: every thread ingests 8 sequences, runs ONE psfm_connect_batch, filters and writes them"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def end_to_end(frames=N_FRAMES, workdir=None):
    """the stage the user calls (main_connect_point_trajectories.py:27-62) disk to disk -- configs[1] written as
    2 x 100 .flo files (3.3 GB) on tmpfs, then .flo -> HBM -> psfm_connect -> min-length filter + D2H -> track.npy (the reference's
    pickle layout), the phases timed inside the entry point; the second (warm) pass is reported."""
    import shutil
    import tempfile
    import torch
    import psfm_synth
    from point_trajectory.utils import write_flo
    from point_trajectory.main_connect_point_trajectories import main_connect_point_trajectories
    base = workdir or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir())
    work = tempfile.mkdtemp(prefix="psfm_e2e_", dir=base)
    try:
        d = psfm_synth.synth_sequence_torch(frames, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
        for name, key in (("flow_f", "flows_f"), ("flow_b", "flows_b")):
            os.makedirs(os.path.join(work, "flows", name))
            arr = d[key].cpu().numpy()
            for i in range(frames - 1):
                write_flo(os.path.join(work, "flows", name, "%05d.flo" % i), arr[i])
        del d, arr
        torch.cuda.empty_cache()
        tm = {}
        for _ in range(2):
            tm = {}
            t0 = time.perf_counter()
            main_connect_point_trajectories(os.path.join(work, "flows"), os.path.join(work, "traj"), sample_ratio=RATIO,
                                            flow_check_thres=THRES, skip_path_consistency=True, timings=tm)
            tm["total_s"] = time.perf_counter() - t0
        size = os.path.getsize(os.path.join(work, "traj", "track.npy"))
        gb = 2 * (frames - 1) * H * W * 8 / 1e9
        return {"workload": "configs[1] disk to disk: 2 x %d .flo files (%.2f GB) on %s -> track.npy (reference pickle layout, %.2f GB), "
                            "warm second pass" % (frames - 1, gb, base, size / 1e9),
                "total_s": tm["total_s"], "ingest_s": tm["ingest_s"], "compute_s": tm["compute_s"], "filter_d2h_s": tm["filter_d2h_s"],
                "write_s": tm["write_s"], "ingest_GBs": gb / tm["ingest_s"], "write_GBs": size / 1e9 / tm["write_s"],
                "trajectory_points_per_s_end_to_end": tm["n_points"] / tm["total_s"], "points": tm["n_points"], "trajectories": tm["n_traj"]}
    finally:
        from point_trajectory.trajectory import wait_for_reclaims
        wait_for_reclaims()
        shutil.rmtree(work, ignore_errors=True)


def two_ranks_one_gpu():
    """scripts/probe_peer_thread_ranks.py in a process of its own (the two rank streams then get hardware queues of their own; inside this
    process they share the four default queues with every stream the other figures created): the headline shape with hard flows over TWO
    thread-ranks on this GPU -- the multi-rank engine's device-paced reject path (psfm_shard_solve_peer) -- beside ONE psfm_connect call on
    the same tensors.  No xGMI link is crossed."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_peer_thread_ranks.py"), "2", str(N_FRAMES), "hard", "peer"],
                       capture_output=True, text=True, timeout=200)
    if r.returncode != 0:
        raise RuntimeError((r.stdout + r.stderr)[-300:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def run_all(ctx, dev, n_frames=N_FRAMES, budget_s=150.0):
    """Every figure outside the timed region (world size 1), in order of how much the reviews lean on it.  A figure that raises
    reports {"error": ...}; once `budget_s` seconds are spent the rest report {"skipped": ...} -- the headline line never waits
    for them longer than that."""
    import torch
    import psfm_synth
    t_start = time.perf_counter()
    out = {}

    def extra(key, fn, *a, **k):
        spent = time.perf_counter() - t_start
        if spent > budget_s:
            out[key] = {"skipped": "extras budget (%.0f s) spent after %.0f s" % (budget_s, spent)}
            return out[key]
        t0 = time.perf_counter()
        try:
            out[key] = fn(*a, **k)
        except Exception as e:   # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if isinstance(out[key], dict):
            out[key]["wall_s"] = time.perf_counter() - t0
        torch.cuda.empty_cache()
        return out[key]

    extra("stream_ceilings", stream_ceilings, dev)
    # the path-consistency path: configs[2]'s shape, then north_star's target workload (the headline shape with the solver on) on the
    # three flow distributions of psfm_synth (clean; HARD: every solve rejects steps; REALISTIC: layers, true disocclusion, correlated error)
    sintel = extra("secondary", secondary_track_optimize, ctx)
    extra("secondary_realistic", secondary_track_optimize, ctx, H, W, n_frames, RATIO, seed=7, k=6,
          dist=dict(psfm_synth.REALISTIC, realistic=True),
          label="headline shape with path consistency, realistic flows (layers, true disocclusion, correlated error, outliers)")
    extra("secondary_hard", secondary_track_optimize, ctx, H, W, n_frames, RATIO, seed=6, k=6, dist=psfm_synth.HARD,
          label="headline shape with path consistency, hard flows")
    extra("secondary_1080p", secondary_track_optimize, ctx, H, W, n_frames, RATIO, seed=5, k=10,
          label="headline shape with path consistency")
    # BASELINE configs[0] (DAVIS shape, sample_ratio 4, track only), alone and 16 per launch; configs[2] 16 per launch
    davis = extra("secondary_davis", secondary_track, ctx, 480, 854, 50, 4, 2, "configs[0] shape (DAVIS snowboard)")
    extra("secondary_davis_batch", secondary_batch, 480, 854, 50, 4, False, THRES, 16, 2, "configs[0] shape (DAVIS snowboard)",
          davis, pmc_key="davis_b16")
    extra("secondary_batch", secondary_batch, 436, 1024, 50, 2, True, THRES, 16, 2, "configs[2] shape (Sintel alley_1)", sintel,
          pmc_key="sintel_b16")
    # ... and the largest batch the entry point takes (64): what a directory of small sequences amortises a frame's latency to
    extra("secondary_davis_batch64", secondary_batch, 480, 854, 50, 4, False, THRES, 64, 2, "configs[0] shape (DAVIS snowboard)", davis, n=3)
    # ONE sequence through the multi-rank engine at world size 1 (what sharding costs before a byte crosses xGMI), hard flows
    extra("single_sequence_hard", single_sequence_sharded, dev, 0, 1, 101, reps=1, flows_dist=psfm_synth.HARD,
          label="headline shape, hard flows (sigma 0.3, 5 % occluders)")
    extra("end_to_end", end_to_end, n_frames)
    extra("end_to_end_batch", end_to_end_batch)
    extra("concurrent", concurrent_sequences, 3, n_frames)
    # configs[4] (ScanNet shape: 1000 frames, dense sample_ratio 1, flow_check_thres 3.0 per README.md:143, full optimize)
    scannet = extra("secondary_scannet", secondary_track_optimize, ctx, 480, 640, 1000, 1, seed=4, k=6, thres=3.0, n=2,
                    label="configs[4] shape (ScanNet, dense)")
    extra("secondary_scannet_batch", secondary_batch, 480, 640, 1000, 1, True, 3.0, 4, 4, "configs[4] shape (ScanNet, dense)",
          scannet, n=2)
    # ONE hard sequence over two ranks (threads of this process, both on this GPU): the cross-rank resident solve beside psfm_connect
    extra("two_ranks_one_gpu_hard", two_ranks_one_gpu)
    extra("single_sequence", single_sequence_sharded, dev, 0, 1, 401)      # configs[3]'s shape: 400 pairs + stride-2 stacks, 26 GB
    out["extras_wall_s"] = time.perf_counter() - t_start
    return out


def _first(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def summary(full):
    """A few numbers per figure for bench.py's one line (ms per sequence, the dominant kernel's fraction of its bound, parity);
    everything else stays in bench_extras.json."""
    s = {}
    short = {"secondary": "sintel_opt", "secondary_1080p": "1080p_opt", "secondary_hard": "1080p_opt_hard",
             "secondary_realistic": "1080p_opt_realistic", "secondary_davis": "davis", "secondary_scannet": "scannet_opt",
             "secondary_davis_batch": "davis_x16", "secondary_davis_batch64": "davis_x64", "secondary_batch": "sintel_opt_x16", "secondary_scannet_batch": "scannet_opt_x4",
             "single_sequence": "one_seq_400f_opt", "single_sequence_hard": "one_seq_100f_opt_hard"}
    for key, name in short.items():
        v = full.get(key)
        if not isinstance(v, dict):
            continue
        if "error" in v or "skipped" in v:
            s[name] = {"error": str(v.get("error") or v.get("skipped"))[:80]}
            continue
        e = {"ms": v.get("ms_per_sequence")}
        frac = (_first(v, "roofline", "frame_kernel", "frac") or _first(v, "roofline", "solver", "frac")
                or _first(v, "roofline", "chain_step", "frac") or _first(v, "frame_launch", "frac"))
        if frac is not None:
            e["frac"] = frac
            e["bound"] = (_first(v, "roofline", "frame_kernel", "bound") or _first(v, "roofline", "solver", "bound")
                          or _first(v, "roofline", "chain_step", "bound") or _first(v, "frame_launch", "bound"))
        par = v.get("parity_first_flows") or v.get("parity")
        if isinstance(par, dict):
            ok = par.get("ids_lengths_equal", par.get("counts_equal_every_sequence"))
            e["parity_ok"] = bool(ok)
            if par.get("max_abs_dxy_px") is not None:
                e["max_dxy"] = par["max_abs_dxy_px"]
        if "ratio_to_one_gpu_call" in v:
            e["vs_one_gpu_call"] = v["ratio_to_one_gpu_call"]
        if "ms_per_sequence_exchange_form" in v:
            e["ms_exchange_form"] = v["ms_per_sequence_exchange_form"]
        if "world_size" in v:
            e["world"] = v["world_size"]
        if isinstance(v.get("cross_rank_solve"), dict):
            e["cross_rank_launches"] = v["cross_rank_solve"]["launches"]
            e["cross_rank_gave_up"] = v["cross_rank_solve"]["gave_up"]
        s[name] = e
    tr = full.get("two_ranks_one_gpu_hard")
    if isinstance(tr, dict) and "peer_ms" in tr:
        s["two_ranks_one_gpu_hard"] = {"ms": tr["peer_ms"], "vs_one_gpu_call": tr["peer_over_psfm_connect"], "cross_rank_solves": _first(tr, "counters_peer", "peer"),
                                       "gave_up": _first(tr, "counters_peer", "peer_redone")}
    ee = full.get("end_to_end")
    if isinstance(ee, dict) and "total_s" in ee:
        s["disk_to_disk_s"] = {"total": ee["total_s"], "ingest": ee.get("ingest_s"), "compute": ee.get("compute_s"), "write": ee.get("write_s")}
    sc = full.get("stream_ceilings")
    if isinstance(sc, dict) and "copy_GBs" in sc:
        s["copy_ceiling_GBs"] = sc["copy_GBs"]
    cc = full.get("concurrent")
    if isinstance(cc, dict) and "ms_per_sequence" in cc:
        s["three_in_flight_ms"] = cc["ms_per_sequence"]
    if "extras_wall_s" in full:
        s["wall_s"] = full["extras_wall_s"]
    return s


if __name__ == "__main__":
    if "--sharded-child" in sys.argv:
        _sharded_child_main(sys.argv[1:])
    else:
        sys.exit("bench_extras.py is bench.py's library; its only command line is the --sharded-child mode bench.py starts itself")
