"""The exact single-sequence multi-rank mode on the GPU: psfm_dist.connect_sharded with the HIP engine
(point_trajectory/shard.py -> psfm_shard_* in libpsfm_hip.so) against the single-process oracle.  world = 1 is the plain
product path; world = 2, 3 run one thread per "rank" on the box's single GPU (tests/_thread_comm.py) -- same driver,
same kernels, same exchange as one process per GPU over RCCL, whose collectives this box cannot host."""
import numpy as np
import pytest

import psfm_synth
from _thread_comm import run_ranks

pytestmark = pytest.mark.gpu

# T, H, W, ratio, seed, sigma, occluders, optimize
CASES = [(9, 58, 76, 2, 21, 0.3, 2, False),
         (10, 60, 84, 2, 22, 0.05, 1, True),       # clean solves: the fused export / control path
         (8, 45, 63, 3, 23, 0.4, 2, True),         # noisy: every solve is redone by the chain (export per iteration)
         (7, 40, 56, 1, 24, 0.15, 1, True)]


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("T,H,W,r,seed,sigma,nocc,optimize", CASES)
def test_connect_sharded_hip_engine(world, T, H, W, r, seed, sigma, nocc, optimize):
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    stack = {k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    torch.cuda.synchronize()

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                eng = HipShardEngine()
                part = psfm_dist.connect_sharded(eng, stack["flows_f"], stack["flows_b"], stack["flows_f2"] if optimize else None,
                                                 stack["flows_b2"] if optimize else None, 1.0, r, flow_check_slice, comm=comm)
                full = psfm_dist.gather_result(part, comm=comm)
                return part, full, dict(eng.counters)
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    if optimize:
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    else:
        O = orc.track(d["flows_f"], occ, r)
    GW = (W + r - 1) // r
    n_local = 0
    for rank, (part, (birth, length, off, xy), cnt) in enumerate(res):
        assert len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
        assert float(np.abs(xy - O.xy).max()) <= 1e-4
        if not optimize:
            assert np.array_equal(xy, O.xy)
        assert [s["iterations"] for s in part["solve_stats"]] == [s["iterations"] for s in O.solves]
        assert [s["termination"] for s in part["solve_stats"]] == [s["termination"] for s in O.solves]
        g0, g1 = part["band"]
        first = part["xy"][part["off"][:-1]]
        gidx = (first[:, 1].astype(np.int64) // r) * GW + first[:, 0].astype(np.int64) // r
        assert ((gidx >= g0) & (gidx < g1)).all()
        n_local += len(part["birth"])
        if optimize:
            assert sum(cnt.values()) == len(O.solves)                         # every solve went exactly one way
            assert world == 1 or cnt["local"] + cnt["local_redone"] == 0     # (several ranks: the exchange form or the cross-rank launch)
            assert world > 1 or "peer" not in cnt
    assert n_local == O.n_traj
    if optimize and sigma <= 0.05:
        # clean sequence: (nearly) every solve in one launch -- the fused frame launch, or, in the window behind a solve that left the
        # Gauss-Newton path, this rank's launch of the cross-rank solve
        assert res[0][2]["fused"] + res[0][2].get("peer", 0) >= len(O.solves) - 3
    if optimize and sigma >= 0.4 and world > 1:
        # noisy sequence: the first solve is redone by the chain protocol, the windows behind it run cross-rank resident launches (or,
        # where those give up -- eight thread-ranks share the process's hardware queues -- the chain protocol again)
        c0 = res[0][2]
        assert c0["fused_redone"] >= 1 and c0["fused_redone"] + c0.get("peer", 0) + c0.get("peer_redone", 0) >= len(O.solves) // 2
    if optimize and sigma >= 0.4 and world == 1:
        # one rank: the first stalled solve sends the windows behind it to the one-GPU call's forms (resident solves)
        assert res[0][2]["local"] >= len(O.solves) - 3 and res[0][2]["fused_redone"] <= 2


def _mixed_sequence(T, H, W, seed, hard_until):
    """frames [0, hard_until) from the noisy distribution, the rest from the clean one (two synthetic sequences of one size spliced:
    parity needs the same input on both sides, not a plausible video)"""
    a = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.4, n_occluders=2, stride2=True)
    b = psfm_synth.synth_sequence(T, H, W, seed=seed + 1, sigma=0.02, n_occluders=0, stride2=True)
    d = {}
    for k in ("flows_f", "flows_b"):
        d[k] = [a[k][t] if t < hard_until else b[k][t] for t in range(len(a[k]))]
    for k in ("flows_f2", "flows_b2"):
        d[k] = [a[k][t] if t < hard_until else b[k][t] for t in range(len(a[k]))]
    return d


@pytest.mark.parametrize("variant", ["resident", "no-budget", "give-up", "two-launches", "control-launch"])
def test_connect_sharded_one_rank_takes_the_one_gpu_solver_forms(variant, monkeypatch):
    """World size 1 (the windowed engine for one long sequence on one GPU): windows whose solves reject steps run them as
    psfm_connect does -- resident solves enqueued behind the chain steps (psfm_shard_solve_local), a stalled solve redone by
    psfm_shard_solve_redo_local -- instead of export -> exchange -> control once per trust-region iteration; clean windows go back to
    fused exports.  Same trajectories, BIT FOR BIT, as the exchange form (PSFM_SHARD_LOCAL=0: the sums are added in the same order),
    decisions equal to the oracle's.  Variants: the caller's context has no say (default: the engine takes the whole resident budget
    for the run and gives it back); a resident launch that gives up (PSFM_PC_SPIN=0: stall flag, redo with launches); no budget to
    be had (PSFM_PC_PERSIST=0: `unroll` launches per solve); the two-launch frame form; fused frames with the export + control launch
    of the exchange form (default on one rank: the frame launch runs the control step itself, psfm_shard_frame with sums_out NULL)."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    T, H, W, r = 60, 45, 63, 2
    d = _mixed_sequence(T, H, W, 31, 20)      # (solves 1-19 reject steps, 20-58 are Gauss-Newton all the way)
    dev = torch.device("cuda", torch.cuda.current_device())
    stack = {k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}

    def run(local):
        with monkeypatch.context() as m:
            m.setenv("PSFM_SHARD_LOCAL", "1" if local else "0")
            if local and variant == "give-up":
                m.setenv("PSFM_PC_SPIN", "0")
            if local and variant == "no-budget":
                m.setenv("PSFM_PC_PERSIST", "0")
            if variant == "two-launches":
                m.setenv("PSFM_SHARD_MERGED", "0")
            if variant == "control-launch":       # (fused frames export their sums and a control launch follows, as on several ranks)
                m.setenv("PSFM_SHARD_LOCAL_CONTROL", "0")
            eng = HipShardEngine(_hip.Context(dev.index or 0))
            eng.ctx.set_capacity(2.0, 24.0)      # (the noisy part ends a track per grid point every other frame)
            part = psfm_dist.connect_sharded(eng, stack["flows_f"], stack["flows_b"], stack["flows_f2"], stack["flows_b2"], 1.0, r,
                                             flow_check_slice)
            launches = eng.ctx.solver_counters()
            assert eng.ctx.resident_budget == 0          # (the engine's own budget is given back)
            return part, dict(eng.counters), launches

    got, cnt, launches = run(True)
    ref, cnt0, _ = run(False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    for k in ("birth", "length", "off", "ids"):
        assert np.array_equal(got[k], ref[k])
    assert np.array_equal(got["xy"], ref["xy"])                          # bit for bit: same sums in the same order
    order = np.argsort(got["ids"])
    assert np.array_equal(got["birth"][order], O.birth) and np.array_equal(got["length"][order], O.length)
    assert [s["iterations"] for s in got["solve_stats"]] == [s["iterations"] for s in O.solves]
    assert [s["termination"] for s in got["solve_stats"]] == [s["termination"] for s in O.solves]
    assert cnt0["local"] + cnt0["local_redone"] == 0
    n = len(O.solves)
    assert cnt["fused"] + cnt["fused_redone"] + cnt["local"] + cnt["local_redone"] == n
    assert cnt["local"] + cnt["local_redone"] >= 16           # the noisy part: resident windows
    assert cnt["fused"] >= 8                                  # the clean tail: back to fused exports
    if variant in ("resident", "two-launches", "control-launch"):
        assert launches["resident_launches"] >= 16 and launches["resident_giveups"] == 0
    if variant == "give-up":
        assert cnt["local_redone"] >= 1 and launches["resident_giveups"] >= 1 and launches["iteration_launches"] > 0
    if variant == "no-budget":
        assert launches["resident_launches"] == 0 and launches["iteration_launches"] > 0


@pytest.mark.parametrize("case", [1, 2])
def test_connect_sharded_hip_engine_two_launches_per_frame(case, monkeypatch):
    """The default is one launch per frame (psfm_shard_frame = chain step + fused export); PSFM_SHARD_MERGED=0 keeps the
    two-launch form (psfm_shard_step, psfm_shard_solve_export) that other engines and callers of the C ABI use: same result."""
    monkeypatch.setenv("PSFM_SHARD_MERGED", "0")
    test_connect_sharded_hip_engine(2, *CASES[case])


@pytest.mark.parametrize("case", [1, 2])
def test_connect_sharded_hip_engine_eight_thread_ranks(case):
    """Eight ranks (the target node's GPU count; here eight threads on the one GPU): 30 / 15 grid rows in bands of 4 and 2 rows --
    clean solves (fused export + control) and noisy ones (every solve redone by the chain protocol, one export per iteration)."""
    test_connect_sharded_hip_engine(8, *CASES[case])


@pytest.mark.parametrize("sigma,world", [(0.05, 2), (0.4, 2), (0.05, 3), (0.4, 3), (0.4, 8)])
def test_connect_sharded_hip_engine_frame_pair_owned_stacks(world, sigma):
    """The sharded mode without replicated flows on the GPU engine: every thread-rank passes only its Stage-A slice of the four
    stacks (device tensors); Stage B's frames come by broadcast (psfm_dist.FrameWindow).  Same result as the oracle."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    # (sigma 0.4: every solve stalls and is redone at a checkpoint, the frames behind it are re-run from the windows)
    T, H, W, r, seed = 10, 60, 84, 2, 22
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=1, stride2=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    n, n2 = T - 1, T - 2

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                lo, hi = psfm_dist.shard_range(n, comm.rank, comm.world)
                lo2, hi2 = psfm_dist.shard_range(n2, comm.rank, comm.world)
                sl = lambda k, a, b: torch.from_numpy(np.stack(d[k][a:b])).to(dev)
                part = psfm_dist.connect_sharded(HipShardEngine(), sl("flows_f", lo, hi), sl("flows_b", lo, hi), sl("flows_f2", lo2, hi2),
                                                 sl("flows_b2", lo2, hi2), 1.0, r, flow_check_slice, comm=comm, n_flows_total=n)
                assert part["frames_read_from_own_slice"] == list(range(lo, hi))
                return psfm_dist.gather_result(part, comm=comm), [s["iterations"] for s in part["solve_stats"]]
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    for (birth, length, off, xy), its in res:
        assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length) and float(np.abs(xy - O.xy).max()) <= 1e-4
        assert its == [s["iterations"] for s in O.solves]


def _proc_worker(rank, world, port, ret):
    """one PROCESS per rank on the box's single GPU, torch.distributed over gloo (device tensors staged through the host by
    psfm_dist.TorchComm): the multi-process plumbing of connect_sharded with the HIP engine"""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth as synth
        from point_trajectory.shard import HipShardEngine, flow_check_slice
        torch.cuda.set_device(0)
        T, H, W, r = 10, 60, 84, 2
        d = synth.synth_sequence(T, H, W, seed=22, sigma=0.05, n_occluders=1, stride2=True)
        st = {k: torch.from_numpy(np.stack(d[k])).cuda() for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
        part = psfm_dist.connect_sharded(HipShardEngine(), st["flows_f"], st["flows_b"], st["flows_f2"], st["flows_b2"], 1.0, r,
                                         flow_check_slice)
        birth, length, off, xy = psfm_dist.gather_result(part)
        ret[rank] = (birth, length, xy, [s["iterations"] for s in part["solve_stats"]], len(part["birth"]))
    finally:
        dist.destroy_process_group()


def test_connect_sharded_two_processes_one_gpu():
    import os
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_proc_worker, args=(world, 33500 + os.getpid() % 2000, ret), nprocs=world, join=True)
    T, H, W, r = 10, 60, 84, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=22, sigma=0.05, n_occluders=1, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert len(ret) == world
    for rank in range(world):
        birth, length, xy, its, n_local = ret[rank]
        assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length) and float(np.abs(xy - O.xy).max()) <= 1e-4
        assert its == [s["iterations"] for s in O.solves] and 0 < n_local < O.n_traj


@pytest.mark.gpu
def test_rccl_operations_of_the_sharded_mode_on_a_one_rank_group():
    """The collectives psfm_dist.TorchComm issues under backend "nccl" (= RCCL): all_reduce(MAX) on the uint8 blocked map,
    all_gather_into_tensor on the f64 solver sums and on the bit-packed occlusion maps, all_gather_object, barrier -- on a
    1-rank group in a child process (dtype / op support of the installed RCCL; what several ranks add is covered by the gloo
    and thread-rank tests)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_rccl_ops.py")], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0 and "rccl ops ok: nccl" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_bench_multi_rank_control_flow_dry_run_on_one_gpu():
    """`bench.py --gpus 2` end to end on the one GPU of the box (PSFM_BENCH_DRYRUN_ONE_GPU=1: every rank on cuda:0, collectives over
    gloo): the rank bookkeeping, the max-time / summed-units reduction, `single_sequence` over frame-pair-owned stacks with their
    broadcasts, the guarded closing barrier -- the line is marked as a dry run and its figures mean nothing."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PSFM_BENCH_DRYRUN_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "11",
                        "--single-seq-frames", "13"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith("{") and len(r.stdout.strip().splitlines()[-1]) < 4096
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "dryrun" in line and line["value"] > 0
    assert line["config"]["world_size"] == 2 and line["config"]["backend"] == "gloo" and len(line["config"]["points_per_rank"]) == 2
    assert line["extras"]["one_seq_400f_opt"]["world"] == 2 and line["extras"]["one_seq_400f_opt"]["ms"] > 0
    full = json.load(open(os.path.join(ROOT, line["extras_file"])))          # the full record: beside bench.py, never in the line
    ranks = full["ranks"]
    assert ranks["world_size"] == 2 and ranks["backend"] == "gloo" and [q["rank"] for q in ranks["per_rank"]] == [0, 1]
    ss = full["single_sequence"]
    assert ss["world_size"] == 2 and ss["trajectories"] > 0 and 0 < ss["local_trajectories_rank0"] < ss["trajectories"]


def test_bench_headline_survives_a_rank_dying_in_the_one_sequence_mode():
    """The one-sequence-over-all-ranks figures run in child processes (bench_extras.sharded_children): a child that dies the way a device
    fault ends a process (SIGABRT) costs those figures, named as such in the line, and nothing else -- the headline line is printed and
    bench.py leaves with code 0."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(PSFM_BENCH_DRYRUN_ONE_GPU="1", PSFM_BENCH_CHILD_ABORT="1", PSFM_BENCH_CHILD_TIMEOUT="120")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "11",
                        "--single-seq-frames", "13"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "roofline" in line
    assert "error" in line["extras"]["one_seq_400f_opt"] and "child" in line["extras"]["one_seq_400f_opt"]["error"]
    assert "error" in line["extras"]["one_seq_100f_opt_hard"]


@pytest.mark.parametrize("lane_f", [2.0, 1.0], ids=["records-full", "lanes-and-records-full"])
@pytest.mark.parametrize("world", [1, 2])
def test_connect_sharded_grows_tables_that_run_full(world, lane_f):
    """A rank whose trajectory-record table is too small learns it when it finalizes (PSFM_ERR_CAPACITY); the ranks agree on it behind
    the recurrence's last collective and ALL run Stage B again with larger tables (psfm_dist.connect_sharded -> HipShardEngine.grow_tables),
    as run_connect does for the one-GPU call.  Result equal to the oracle's; the run in between left nothing behind."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    T, H, W, r = 36, 100, 160, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=41, sigma=0.6, n_occluders=3, stride2=True)     # noisy: 57 k trajectories on 4 000 grid points
    dev = torch.device("cuda", torch.cuda.current_device())
    stack = {k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    torch.cuda.synchronize()
    grown = []

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                eng = HipShardEngine(_hip.Context(dev.index or 0))
                eng.ctx.set_capacity(lane_f, 1.0)         # record tables for ~40 k trajectories (max(1, n_flows / 8) per grid point + head-room) where the sequence ends 57 k
                calls = []
                grow = eng.grow_tables
                eng.grow_tables = lambda: (calls.append(1), grow())
                part = psfm_dist.connect_sharded(eng, stack["flows_f"], stack["flows_b"], stack["flows_f2"], stack["flows_b2"], 1.0, r,
                                                 flow_check_slice, comm=comm)
                grown.append(len(calls))
                return psfm_dist.gather_result(part, comm=comm), part
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert len(grown) == world and min(grown) >= 1 and len(set(grown)) == 1      # every rank grew, the same number of times
    for (birth, length, off, xy), part in res:
        assert len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
        assert float(np.abs(xy - O.xy).max()) <= 1e-4
        assert [s["iterations"] for s in part["solve_stats"]] == [s["iterations"] for s in O.solves]


def _hard_sequence(T, H, W, seed):
    return psfm_synth.synth_sequence(T, H, W, seed=seed, stride2=True, **psfm_synth.HARD)


@pytest.mark.parametrize("variant", ["peer", "peer-give-up", "exchange"])
@pytest.mark.parametrize("world", [2, 3])
def test_cross_rank_resident_solve(world, variant, monkeypatch):
    """Several ranks, flows whose every solve rejects steps: after the first window the solves run as ONE resident launch per rank whose
    all-reduce crosses the ranks on the device (psfm_shard_solve_peer: leaders write into every rank's granule area; thread-ranks pass
    plain pointers) -- trajectory_optimize.cpp:74-82's global accept / reject with no host and no collective in the loop.  Same ids,
    lengths, per-solve iterations / accepted steps / terminations as the oracle; positions within rounding of the exchange form
    (PSFM_SHARD_PEER=0), which adds the ranks' sums in the same order.  peer-give-up: one block of one launch gives up (PSFM_PC_QUIT) --
    its poison reaches every rank, they all stall on that frame and redo it in the exchange form."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    T, H, W, r = 40, 58, 76, 2
    d = _hard_sequence(T, H, W, 61)
    dev = torch.device("cuda", torch.cuda.current_device())
    stack = {k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    torch.cuda.synchronize()
    if variant == "exchange":
        monkeypatch.setenv("PSFM_SHARD_PEER", "0")
    if variant == "peer-give-up":
        monkeypatch.setenv("PSFM_PC_QUIT", "0,3")           # block 0 of every launch leaves in round 3

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                eng = HipShardEngine(_hip.Context(dev.index or 0))
                eng.ctx.set_capacity(2.0, 24.0)
                part = psfm_dist.connect_sharded(eng, stack["flows_f"], stack["flows_b"], stack["flows_f2"], stack["flows_b2"], 1.0, r,
                                                 flow_check_slice, comm=comm)
                return psfm_dist.gather_result(part, comm=comm), part["solve_stats"], dict(eng.counters)
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert sum(s["iterations"] - s["successful_steps"] > 1 for s in O.solves) > len(O.solves) // 2        # the solves do reject steps
    for (birth, length, off, xy), stats, cnt in res:
        assert len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
        assert float(np.abs(xy - O.xy).max()) <= 1e-4
        for key in ("iterations", "successful_steps", "termination", "dogleg_nonGN"):
            assert [s[key] for s in stats] == [s[key] for s in O.solves], key
        assert cnt == res[0][2]                                  # every rank took the same path for every solve
        if variant == "exchange":
            assert "peer" not in cnt
        elif variant == "peer":
            # (thread-ranks share the process's hardware queues: when two ranks' streams land on one queue their launches run one behind the
            # other, never meet, and give up -- at most twice per run, then the run keeps the exchange form; rare, not an error)
            assert cnt["peer"] + cnt["peer_redone"] >= 1 and cnt["peer_redone"] <= 2, cnt
            assert cnt["peer"] >= len(O.solves) // 2 or cnt["peer_redone"] == 2, cnt
        else:
            assert cnt["peer_redone"] >= 1, cnt
    test_cross_rank_resident_solve.xy = getattr(test_cross_rank_resident_solve, "xy", {})
    test_cross_rank_resident_solve.xy[(world, variant)] = res[0][0][3]
    other = test_cross_rank_resident_solve.xy.get((world, "exchange" if variant != "exchange" else "peer"))
    if other is not None:
        assert float(np.abs(other - res[0][0][3]).max()) <= 1e-9


def _peer_proc_worker(rank, world, port, ret):
    """one PROCESS per rank on the box's single GPU: the granule areas travel as IPC handles (hipIpcGetMemHandle / OpenMemHandle), each
    process's resident launch takes half of the device's co-resident block slots"""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth as synth
        from point_trajectory.shard import HipShardEngine, flow_check_slice
        torch.cuda.set_device(0)
        T, H, W, r = 40, 58, 76, 2
        d = synth.synth_sequence(T, H, W, seed=61, stride2=True, **synth.HARD)
        st = {k: torch.from_numpy(np.stack(d[k])).cuda() for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
        eng = HipShardEngine()
        eng.ctx.set_capacity(2.0, 24.0)
        part = psfm_dist.connect_sharded(eng, st["flows_f"], st["flows_b"], st["flows_f2"], st["flows_b2"], 1.0, r, flow_check_slice)
        birth, length, off, xy = psfm_dist.gather_result(part)
        ret[rank] = (birth, length, xy, [(s["iterations"], s["termination"]) for s in part["solve_stats"]], dict(eng.counters))
    finally:
        dist.destroy_process_group()


def test_cross_rank_resident_solve_two_processes_one_gpu():
    import os
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_peer_proc_worker, args=(world, 35500 + os.getpid() % 2000, ret), nprocs=world, join=True)
    T, H, W, r = 40, 58, 76, 2
    d = _hard_sequence(T, H, W, 61)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    assert len(ret) == world
    for rank in range(world):
        birth, length, xy, its, cnt = ret[rank]
        assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length) and float(np.abs(xy - O.xy).max()) <= 1e-4
        assert its == [(s["iterations"], s["termination"]) for s in O.solves]
        # the cross-rank form ran (two processes' launches were co-resident on the one GPU) -- or every one of them gave up and was
        # redone in the exchange form, which is still correct; say which
        assert cnt["peer"] + cnt["peer_redone"] >= 1 and cnt["peer_redone"] <= 2, cnt
        print("rank %d counters: %s" % (rank, cnt))


def test_cross_rank_solve_new_engine_on_a_used_context():
    """The granule area belongs to the CONTEXT and outlives the engine object that drove it: a second run with a NEW engine on the same
    contexts must go on from the area's last epoch (psfm_shard_peer_epoch) -- starting over at epoch 1 would find the first run's granules
    under the same tags and add THEIR sums.  Two different hard sequences one after the other on the same two contexts, new engines."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    world, H, W, r = 2, 58, 76, 2
    seqs = [_hard_sequence(24, H, W, 71), _hard_sequence(24, H, W, 72)]
    dev = torch.device("cuda", torch.cuda.current_device())
    stacks = [{k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")} for d in seqs]
    torch.cuda.synchronize()

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                ctx = _hip.Context(dev.index or 0)
                ctx.set_capacity(2.0, 24.0)
                out = []
                for st in stacks:
                    eng = HipShardEngine(ctx)                 # a NEW engine (epoch counter at zero) on the USED context
                    part = psfm_dist.connect_sharded(eng, st["flows_f"], st["flows_b"], st["flows_f2"], st["flows_b2"], 1.0, r, flow_check_slice, comm=comm)
                    out.append((psfm_dist.gather_result(part, comm=comm), part["solve_stats"], dict(eng.counters), int(eng._epoch)))
                return out
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    for k, d in enumerate(seqs):
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        for rank in range(world):
            (birth, length, off, xy), stats, cnt, epoch = res[rank][k]
            assert np.array_equal(birth, O.birth) and np.array_equal(length, O.length) and float(np.abs(xy - O.xy).max()) <= 1e-4
            assert [s["iterations"] for s in stats] == [s["iterations"] for s in O.solves]
            assert cnt["peer"] + cnt["peer_redone"] >= 1 and cnt["peer_redone"] <= 2, cnt
    assert res[0][1][3] > res[0][0][3] >= 1           # the second run's epochs continue behind the first run's
