#!/usr/bin/env python3
"""bench.py -- throughput of the point-trajectory hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one synthetic sequence: BASELINE.json configs[1] =
100 x (1920x1080) forward/backward flow pairs, sample_ratio=2, chaining + occlusion only, i.e.
    flow_check(100 pairs) -> track(100 flows) -> finalize (ids, lengths, id-ordered CSR result),
= ONE psfm_connect call (the compute part of the stage entry main_connect_point_trajectories.py:36-53),
with the flow stacks already resident in HBM when the timed region starts and the result left in HBM.
metric  = trajectory points per second (sum over all trajectories of their length / wall time).
N > 1   = N independent sequences, one per rank/GPU (sequences are the unit the reference's driver
          loops over, run_particlesfm.py:168-176); no data-path collective, weak scaling.

Rank 0 prints ONE JSON line, kept under 4 KB (tests/test_bench_line.py): the contract's keys, `roofline` of the
flow-chaining kernel (HIP events on the launch stream, inside the timed region), `cpu_baseline` (the CPU oracle on a
bounded sample of the same workload), `parity`, and `extras` = a few numbers per figure measured outside the timed
region.  The full records of those figures (bench_extras.py) go to bench_extras.json, never into the line.
"""
import argparse
import json
import os
import sys
import time

from bench_common import (ROOT, H, W, N_FRAMES, RATIO, THRES, HBM_PEAK_GBS, quiet_gc, source_sha16, replayed, cpu_baseline)

LINE_LIMIT = 4096     # bytes of the one JSON line (the round-5 line was 33 KB and the driver could not parse it)


def _round(v, sig=6):
    """Floats to `sig` significant digits, recursively: the line is for reading and parsing, the full digits are in bench_extras.json."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v))
    if isinstance(v, dict):
        return {k: _round(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_round(x, sig) for x in v]
    return v


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "kernels", "extras", "extras_file", "dryrun")


def compact_line(out, limit=LINE_LIMIT):
    """The one line: HEADLINE_KEYS of `out`, floats rounded, and -- should it still not fit -- the optional parts dropped one by one
    (extras first).  The contract's keys, `roofline` and `cpu_baseline` are never dropped."""
    line = {k: out[k] for k in HEADLINE_KEYS if k in out and out[k] is not None or k == "vs_baseline"}
    line = _round(line)
    for victim in (None, "extras", "kernels", "parity"):
        if victim is not None:
            line.pop(victim, None)
        s = json.dumps(line, separators=(",", ":"))
        if len(s) <= limit:
            return s
    raise ValueError("bench line is %d bytes with every optional part dropped (limit %d)" % (len(s), limit))


def write_full(record):
    """The full record (headline + every figure measured outside the timed region) beside bench.py and, on a gpurun box, under
    gpurun_out/ so that it travels back.  Returns the path the line names."""
    paths = [os.path.join(ROOT, "bench_extras.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) or os.environ.get("GRAFT_REPO_ROOT"):
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_extras.json"))
    wrote = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(record, f, indent=1)
            wrote = wrote or os.path.relpath(p, ROOT)
        except OSError:
            pass
    return wrote



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=N_FRAMES, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pairs", type=int, default=100, help=argparse.SUPPRESS)   # whole workload: ~10 s of CPU
    ap.add_argument("--no-cpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--single-seq-frames", type=int, default=401, help=argparse.SUPPRESS)   # configs[3]: 400 pairs
    ap.add_argument("--extras-budget", type=float, default=150.0, help=argparse.SUPPRESS)   # seconds the extra figures may take in all
    ap.add_argument("--no-extras", action="store_true",
                    help="only warm-up + timed steps (what profiles/*_kernel_stats.csv is collected with): skips the figures "
                         "measured outside the timed region (overlapped psfm_connect, track_optimize, concurrent sequences)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched bare (`python bench.py --gpus N`): become the launcher -- one rank per GPU under torch.distributed.run,
        # exactly the command line the driver uses -- and hand its exit code back.  Never print a line for fewer GPUs than asked.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the job has %d rank(s) (WORLD_SIZE): refusing to report a line for another size"
                 % (args.gpus, world))
    # PSFM_BENCH_DRYRUN_ONE_GPU=1: every rank on cuda:0, collectives over gloo -- a dry run of the multi-rank CONTROL FLOW on a box with
    # one GPU (the ranks fight for the device: the line it prints is marked and is not a measurement)
    dryrun = world > 1 and os.environ.get("PSFM_BENCH_DRYRUN_ONE_GPU", "0") == "1"
    if dryrun:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d needs cuda:%d, %d device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dryrun:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import psfm_synth
    from point_trajectory import _hip
    from point_trajectory.utils import flow_check_device
    from point_trajectory.trajectory import run_connect, run_track
    import point_trajectory.shard, point_trajectory.batch, psfm_dist      # noqa: E401,F401  (everything imported before the freeze)
    from bench_extras import guarded, sharded_children

    quiet_gc()

    n_frames = args.frames
    n_flows = n_frames - 1
    # one sequence per rank, different seeds (rank 0 = BASELINE seed 0)
    d = psfm_synth.synth_sequence_torch(n_frames, H, W, seed=rank, sigma=0.05, n_occluders=2, stride2=False,
                                        device=dev)
    flows_f, flows_b = d["flows_f"], d["flows_b"]
    ctx = _hip.context(local_rank)

    def step():
        # psfm_connect: with the device to itself it is ONE persistent launch that checks flow consistency and runs the
        # recurrence (+ finalize).  The two-call form (flow_check, then psfm_track on the maps) and the per-frame-launch
        # path are measured below, outside the timed region, on the same data.
        return run_connect(flows_f, flows_b, None, None, THRES, RATIO, return_device=True)

    def step_two_calls():
        _, occ = flow_check_device(flows_f, flows_b, THRES)
        return run_track(flows_f, occ, None, None, RATIO, return_device=True)

    if os.environ.get("PSFM_BENCH_TWO_CALLS"):      # profiling only: the stand-alone flow_check + the loop on its maps as the step
        step = step_two_calls
    if os.environ.get("PSFM_BENCH_CHAIN_MODE"):     # profiling only: 1 = one chain_step launch per frame as the step
        ctx.set_chain_mode(int(os.environ["PSFM_BENCH_CHAIN_MODE"]))

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.set_profiling(False)
    for _ in range(args.warmup):
        info = step()
    # ---- timed region: exactly K steps, HIP events around every kernel family ----
    # HIP events around flow_check / finalize and around every 8th chain_step launch (hipExtLaunchKernelGGL
    # start/stop events = exact kernel begin/end): timing every launch would cost ~0.6 ms of host time per step
    ctx.set_profiling(0 if os.environ.get("PSFM_BENCH_NOPROF") else 8)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = ctx.profile()
    ctx.set_profiling(False)
    # ---- the other ways of running the same step, outside the timed region ----
    def timed(fn, mode, prof_stride):
        ctx.set_chain_mode(mode)
        fn()
        ctx.set_profiling(prof_stride)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            inf = fn()
        sync_all()
        ms = 1e3 * (time.perf_counter() - t1) / args.steps
        pr = ctx.profile()
        ctx.set_profiling(False)
        ctx.set_chain_mode(0)
        assert int(inf.n_points) == int(info.n_points) and int(inf.n_traj) == int(info.n_traj)
        return ms, pr, inf
    variants = None
    if not args.no_extras:
        us = lambda pr, k: 1e3 * pr[k]["total_ms"] / max(pr[k]["launches"], 1)
        ms2, pr2, inf2 = timed(step_two_calls, 0, 1)       # flow_check kernel, then the persistent loop on its maps
        ms1, pr1, inf1 = timed(step, 1, 8)                 # psfm_connect with one chain_step launch per frame
        variants = {
            "two_calls_flow_check_then_track": {"ms_per_step": ms2, "chain_mode": int(inf2.chain_mode),
                                                "flow_check_launch_us": us(pr2, "flow_check"),
                                                "chain_launch_us": us(pr2, "chain_step"),
                                                "chain_us_per_step": us(pr2, "chain_step") / (n_flows if int(inf2.chain_mode) == 2 else 1)},
            "per_frame_launches_overlapped": {"ms_per_step": ms1, "chain_mode": int(inf1.chain_mode),
                                              "chain_step_avg_launch_us": us(pr1, "chain_step")},
        }

    points = int(info.n_points)
    import psfm_dist
    # who took part: rank, device, and the collective library as torch reports it
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank), "points": points,
          "pid": os.getpid()}
    rank_info = [me]
    if world > 1:
        rank_info = [None] * world
        dist.all_gather_object(rank_info, me)
        try:
            me_v = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            me_v = None
        rank_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": me_v, "per_rank": rank_info}
    dt_max, total_points = psfm_dist.reduce_totals(dt, points, device=dev)   # max time, summed units over ranks

    # ---- ONE sequence over all ranks (exact track-sharded mode), outside the timed region ----
    single, single_hard, hung = None, None, False
    if world > 1:      # (world size 1: bench_extras.run_all times the same two figures, inside its budget)
        # configs[3], and the headline shape on flows whose solves reject steps (psfm_synth.HARD: the redo path of the sharded engine) --
        # each rank's share in a child process: a device fault in a cross-GPU form must not take the headline figure with it
        dist.barrier()
        res, hung = guarded(lambda: sharded_children(rank, world, args.single_seq_frames), 600)
        single, single_hard = res if isinstance(res, tuple) else (res, None)      # (a dict = guarded()'s own error record)

    if rank == 0:
        # ---- roofline of the flow-chaining kernel (K2): algorithmic bytes per launch / avg duration ----
        # chain_step/frame = min(8P,32A) + min(P,4A) + 16A + 16A + A, A = tracks alive at the step.
        import ctypes
        birth = np.empty(int(info.n_traj), np.int32)
        length = np.empty(int(info.n_traj), np.int32)
        _hip.check(_hip.lib().psfm_result_copy(ctx.handle, birth.ctypes.data_as(ctypes.c_void_p),
                                               length.ctypes.data_as(ctypes.c_void_p), None, None,
                                               _hip.current_stream_ptr()))
        last = birth.astype(np.int64) + length - 1
        alive_steps = float(points - int((last == n_flows).sum()))   # sum_t A_t over the n_flows launches
        A = alive_steps / n_flows
        P = float(H * W)
        chain_bytes = min(8 * P, 32 * A) + min(P, 4 * A) + 16 * A + 16 * A + A
        ch = prof["chain_step"]
        chain_us = 1e3 * ch["total_ms"] / max(ch["launches"], 1)
        persistent = int(info.chain_mode) == 2
        fused = persistent and prof["flow_check"]["launches"] == 0    # no stand-alone flow_check ran: it is in the loop
        frame_bytes = chain_bytes
        fc_step_bytes = 17.0 * P
        if persistent:
            # ONE launch runs all n_flows steps (psfm_chain_persist_kernel): algorithmic bytes per launch = the per-step
            # figure x n_flows -- the positions it keeps in registers between steps are still counted as read.  When
            # psfm_connect fuses flow_check into the loop, that kernel also moves flow_check's 17P bytes per step.
            frame_bytes = chain_bytes + (fc_step_bytes if fused else 0.0)
            chain_bytes = frame_bytes * n_flows
        achieved = chain_bytes / (chain_us * 1e-6) / 1e9 if chain_us > 0 else 0.0
        tfile = os.path.join(ROOT, "profiles", ("traffic_chain_fused.json" if fused else "traffic_chain_persist.json")
                             if persistent else "traffic_chain_step.json")
        tv, tprov = replayed(tfile)
        traffic = tv.get("hbm_bytes_per_launch") if tv else None
        fc = prof["flow_check"]
        fc_us = 1e3 * fc["total_ms"] / max(fc["launches"], 1)
        fc_bytes = 17.0 * P * n_flows
        if fused and variants:    # the stand-alone kernel, from the two-call variant
            fc_us = variants["two_calls_flow_check_then_track"]["flow_check_launch_us"]
        sha = source_sha16()
        out = {
            "metric": "trajectory-points/s", "value": total_points * args.steps / dt_max,
            "unit": "trajectory-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 sampling + f64 positions", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic %dx(1920x1080) flow pairs, sample_ratio=2, "
                                   "flow_check + chaining + occlusion + id assignment, 1 sequence per GPU" % n_flows,
                       "frames": n_frames, "height": H, "width": W, "sample_ratio": RATIO,
                       "flow_check_thres": THRES, "points_per_sequence": points, "trajectories": int(info.n_traj),
                       "parallelism": "sequence-per-gpu x%d" % world, "world_size": dist.get_world_size() if world > 1 else 1,
                       "source_sha16": sha},
            "roofline": {"bound": "hbm",
                         "kernel": ("psfm_chain_persist_kernel (flow_check fused in)" if fused else "psfm_chain_persist_kernel")
                         if persistent else "psfm_chain_step_kernel",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # physical HBM bytes of the same launch (fabric-side PMC counters, separate passes, replayed from profiles/) and
                         # the fraction of the peak THEY amount to
                         "traffic": traffic,
                         "frac_physical": (traffic / (chain_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (traffic and chain_us > 0) else None,
                         "traffic_file": tprov["file"] if tprov else None,
                         "traffic_same_sources": tprov["same_sources"] if tprov else None,
                         "bytes_per_launch": chain_bytes, "avg_launch_us": chain_us,
                         "steps_per_launch": n_flows if persistent else 1,
                         "bytes_per_step": frame_bytes,
                         "chain_step_bytes_per_step": frame_bytes - (fc_step_bytes if fused else 0.0),
                         "avg_alive_tracks": A},
            "kernels": {"flow_check_alone_us": fc_us,
                        "flow_check_alone_frac": fc_bytes / (fc_us * 1e-6) / 1e9 / HBM_PEAK_GBS if fc_us > 0 else None,
                        "finalize_avg_us": 1e3 * prof["finalize"]["total_ms"] / max(prof["finalize"]["launches"], 1)},
        }
        if variants:
            v2 = variants["two_calls_flow_check_then_track"]
            out["kernels"]["chain_alone_us_per_step"] = v2["chain_us_per_step"]
            out["kernels"]["chain_alone_frac"] = (frame_bytes - (fc_step_bytes if fused else 0.0)) / (v2["chain_us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS \
                if v2["chain_us_per_step"] > 0 else None
            out["kernels"]["two_calls_ms_per_step"] = v2["ms_per_step"]
            out["kernels"]["per_frame_launches_ms_per_step"] = variants["per_frame_launches_overlapped"]["ms_per_step"]
        if world > 1:
            out["config"]["backend"] = rank_info["backend"]
            out["config"]["rccl_version"] = rank_info["rccl_version"]
            out["config"]["points_per_rank"] = [r["points"] for r in rank_info["per_rank"]]
        if dryrun:
            out["dryrun"] = "%d ranks on ONE GPU over gloo: the multi-rank control flow only, NOT a measurement" % world
        full = {"headline": None, "ranks": rank_info, "variants": variants, "traffic_source": tprov}
        if single is not None:
            full["single_sequence"] = single
        if single_hard is not None:
            full["single_sequence_hard"] = single_hard
        if world == 1 and not args.no_cpu:
            n_cpu = min(args.cpu_pairs, n_flows)
            cb, Rc = cpu_baseline(flows_f, flows_b, n_cpu)
            full["cpu_baseline"] = cb
            refc = cb.get("reference_python_build_container") or {}
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "host_cores": cb["host_cores"],
                                   "sample": "%d of %d frame pairs of this workload (flow_check + track + id order), C restatement of the reference path, "
                                             "%d OpenMP threads" % (n_cpu, n_flows, cb["cores"]),
                                   "value_8_threads": cb["port_8_threads"]["value"],
                                   # the unmodified reference Python: timed in the build container only (it cannot travel to this box)
                                   "reference_python_points_per_s_build_container": refc.get("track_points_per_s")}
            if n_cpu == n_flows:
                # full-size parity, checked on the spot: ids / lengths bit-exact, positions bit-exact (track mode)
                from point_trajectory.trajectory import _result_to_host
                Rg = _result_to_host(ctx, info)
                same = bool(Rg.birth.shape == Rc.birth.shape and np.array_equal(Rg.birth, Rc.birth)
                            and np.array_equal(Rg.length, Rc.length))
                out["parity"] = {"vs": "cpu oracle, whole workload", "ids_lengths_equal": same,
                                 "max_abs_dxy_px": float(np.abs(Rg.xy - Rc.xy).max()) if same else None}
                del Rg
            del Rc
        if world == 1 and not args.no_extras:
            # every figure outside the timed region: bench_extras.py; the line carries a few numbers of each
            import bench_extras
            del flows_b, flows_f, d
            torch.cuda.empty_cache()
            full.update(bench_extras.run_all(ctx, dev, n_frames, budget_s=args.extras_budget))
        out["extras"] = bench_extras_summary(full)
        full["headline"] = dict(out)
        out["extras_file"] = write_full(full)
        print(compact_line(out), flush=True)
    if hung:            # a rank is stuck in a collective of the extra mode: the line is out, leave without the barrier
        sys.stdout.flush()
        os._exit(0)
    if world > 1:
        # (a rank that left early -- an error in its extra mode -- must not keep the others in this barrier for ever)
        _, stuck = guarded(lambda: (dist.barrier(), torch.cuda.synchronize()), 120)
        if stuck:
            os._exit(0)
        dist.destroy_process_group()


def bench_extras_summary(full):
    import bench_extras
    return bench_extras.summary(full)


if __name__ == "__main__":
    main()
