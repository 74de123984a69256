"""world_size-2 gloo tests (CPU) of the multi-process path: sequence sharding, the frame-pair-sharded
flow_check + all-gather stitch, and bench.py's time/units reduction.  The per-slice compute on CPU is the oracle
(test infrastructure); on the GPU box the same code path runs psfm_flow_check per rank over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth
        from oracle import oracle as orc
        n_pairs, H, W = 5, 37, 53                      # ragged: 5 pairs over 2 ranks, H*W not a multiple of 8
        d = psfm_synth.synth_sequence(n_pairs + 1, H, W, seed=3, sigma=0.4, n_occluders=2, stride2=False)
        ff = torch.from_numpy(np.stack(d["flows_f"]))
        fb = torch.from_numpy(np.stack(d["flows_b"]))
        calls = []

        def check(f, b, thres):
            calls.append(int(f.shape[0]))
            _, occ = orc.flow_check(list(f.numpy()), list(b.numpy()), thres)
            return torch.from_numpy(np.stack(occ).astype(np.uint8))

        occ = psfm_dist.flow_check_sharded(ff, fb, 1.0, check)
        _, ref = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        ok_fc = bool(np.array_equal(occ.numpy().astype(bool), np.stack(ref)))
        lo, hi = psfm_dist.shard_range(n_pairs, rank, world)
        ok_slice = calls == [hi - lo]
        # sequences: round robin, every sequence exactly once
        mine = psfm_dist.shard_sequences(7, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok_seq = sorted(sum(gathered, [])) == list(range(7))
        # the exact recurrence runs whole on the rank that owns the sequence: identical to a single-process run
        R = orc.track(d["flows_f"], [o for o in occ.numpy()], 2)
        R1 = orc.track(d["flows_f"], ref, 2)
        ok_track = bool(np.array_equal(R.xy, R1.xy) and np.array_equal(R.length, R1.length))
        t, u = psfm_dist.reduce_totals(1.0 + rank, 10.0 * (rank + 1))
        ok_red = (t == float(world)) and (u == 10.0 * world * (world + 1) / 2)
        ret[rank] = (ok_fc, ok_slice, ok_seq, ok_track, ok_red)
    finally:
        dist.destroy_process_group()


def _sharded_worker(rank, world, port, ret):
    """psfm_dist.connect_sharded (the product's driver of the exact single-sequence multi-rank mode) over gloo, with
    the oracle's ShardEngine as each rank's compute: Stage A frame-pair shards, Stage B tracks by birth row band."""
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth
        from oracle import oracle as orc

        def check(f, b, thres):
            _, occ = orc.flow_check(list(f.numpy()), list(b.numpy()), thres)
            return torch.from_numpy(np.stack(occ).astype(np.uint8)) if len(occ) else torch.zeros((0,) + tuple(f.shape[1:3]), dtype=torch.uint8)

        out = {}
        # (T, H, W, ratio, seed, sigma, occluders, optimize): grid rows not divisible by the world size, deaths and
        # respawns in every band, a noisy sequence whose solves reject steps (every dogleg case goes through the hook)
        cases = [(7, 38, 52, 2, 11, 0.3, 2, False), (8, 45, 60, 3, 12, 0.1, 1, True), (6, 30, 44, 1, 13, 0.4, 2, True)]
        for ci, (T, H, W, r, seed, sigma, nocc, optimize) in enumerate(cases):
            d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
            tf = lambda k: torch.from_numpy(np.stack(d[k]))
            part = psfm_dist.connect_sharded(orc.ShardEngine(), tf("flows_f"), tf("flows_b"),
                                             tf("flows_f2") if optimize else None, tf("flows_b2") if optimize else None,
                                             1.0, r, check)
            birth, length, off, xy = psfm_dist.gather_result(part)
            _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
            if optimize:
                _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
                O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            else:
                O = orc.track(d["flows_f"], occ, r)
            same_ids = bool(len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length))
            err = float(np.abs(xy - O.xy).max()) if same_ids else None
            stats_ok = [s["iterations"] for s in part["solve_stats"]] == [s["iterations"] for s in O.solves] and \
                       [s["termination"] for s in part["solve_stats"]] == [s["termination"] for s in O.solves]
            g0, g1 = part["band"]
            GW = (W + r - 1) // r
            first = part["xy"][part["off"][:-1]]
            gidx = (first[:, 1].astype(np.int64) // r) * GW + first[:, 0].astype(np.int64) // r
            own_band = bool(((gidx >= g0) & (gidx < g1)).all())
            out[ci] = (same_ids, err, stats_ok, own_band, len(part["birth"]), O.n_traj)
        # every track dies in one step (an all-occluded map): SciPy's phantom-feature respawn rule needs the GLOBAL
        # "no survivor" flag
        T, H, W, r = 5, 24, 30, 2
        d = psfm_synth.synth_sequence(T, H, W, seed=31, sigma=0.05, stride2=False)

        def check_dead(f, b, thres):
            o = check(f, b, thres)
            return o
        ff, fb = torch.from_numpy(np.stack(d["flows_f"])), torch.from_numpy(np.stack(d["flows_b"]))
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        occ = [o.copy() for o in occ]
        occ[1][:] = True
        eng = orc.ShardEngine()
        # (drive Stage B directly with the modified maps)
        GW, GH = (W + r - 1) // r, (H + r - 1) // r
        g0, g1 = psfm_dist.band_range(GH, GW, rank, world)
        eng.begin(T - 1, H, W, r, g0, g1, False)
        for t in range(T - 1):
            x = eng.step(t, ff[t], torch.from_numpy(occ[t].astype(np.uint8)))
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            eng.after_exchange(t, x)
        b_, l_, o_, xy_, _ = eng.finish()
        ids, n = psfm_dist.global_ids(b_, l_, xy_[o_[:-1]], T - 1, r, GW)
        Bd, Ld, Od, XYd = psfm_dist.gather_result({"birth": b_, "length": l_, "off": o_, "xy": xy_, "ids": ids, "n_traj": n})
        O = orc.track(d["flows_f"], occ, r)
        out["alldie"] = bool(n == O.n_traj and np.array_equal(Bd, O.birth) and np.array_equal(Ld, O.length) and np.array_equal(XYd, O.xy))
        # a rank whose tables run full (PSFM_ERR_CAPACITY when it finalizes): ALL ranks run Stage B again with larger tables
        class FullOnce(orc.ShardEngine):
            def __init__(self, full_on_rank):
                super().__init__()
                self.full, self.grown, self.begun = (rank == full_on_rank), 0, 0

            def begin(self, *a):
                self.begun += 1
                return super().begin(*a)

            def grow_tables(self):
                self.grown += 1

            def finish(self):
                if self.full:
                    self.full = False
                    super().finish()
                    e = RuntimeError("libpsfm_hip status 3: lane table full")
                    e.status = 3
                    raise e
                return super().finish()

        T, H, W, r = 7, 38, 52, 2
        d = psfm_synth.synth_sequence(T, H, W, seed=11, sigma=0.3, n_occluders=2, stride2=True)
        tf = lambda k: torch.from_numpy(np.stack(d[k]))
        eng = FullOnce(world - 1)
        part = psfm_dist.connect_sharded(eng, tf("flows_f"), tf("flows_b"), tf("flows_f2"), tf("flows_b2"), 1.0, r, check)
        birth, length, off, xy = psfm_dist.gather_result(part)
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        out["capacity_retry"] = bool(eng.grown == 1 and eng.begun == 2 and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
                                     and float(np.abs(xy - O.xy).max()) <= 1e-9)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _owned_worker(rank, world, port, ret, lean=False):
    """connect_sharded with the stacks OWNED by frame-pair slices: every rank holds (and passes) only its slice of the four
    stacks; Stage B's frames arrive by broadcast from their owners (psfm_dist.FrameWindow)."""
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psfm_dist
        import psfm_synth
        from oracle import oracle as orc

        def check(f, b, thres):
            _, occ = orc.flow_check(list(f.numpy()), list(b.numpy()), thres)
            return torch.from_numpy(np.stack(occ).astype(np.uint8)) if len(occ) else torch.zeros((0,) + tuple(f.shape[1:3]), dtype=torch.uint8)

        class DeferredEngine(orc.ShardEngine):
            """The shape of the HIP engine on top of the oracle's: a frame (step + solve) is only ENQUEUED -- with the tensors the
            driver handed over, like kernels that read them later -- and a checkpoint runs the queue.  At the frames in `stall_at`
            it behaves like a solve that did not go as speculated: that frame is completed, everything enqueued behind it is
            dropped (on the device those launches were no-ops) and the driver is told to run the frames behind it again.  If the
            driver's frame window recycled a buffer too early, or rewound to the wrong frame, the trajectories change."""

            def __init__(self, stall_at):
                super().__init__()
                self.queue, self.stall_at, self.check_every, self.reruns, self.windows = [], set(stall_at), 16, 0, []

            def frame(self, t, flow_prev, flow_cur, occ, flow2_prev, occ2_prev, reduce_first, reduce):
                self.queue.append((t, flow_prev, flow_cur, occ, flow2_prev, occ2_prev, reduce_first))

            def stalled(self):
                return False

            def checkpoint(self, reduce):
                redo = None
                self.windows.append(len(self.queue))
                for k, (t, fp, fc, oc, f2, o2, reduce_first) in enumerate(self.queue):
                    x = orc.ShardEngine.step(self, t, fc, oc)
                    reduce_first(x)
                    orc.ShardEngine.after_exchange(self, t, x)
                    orc.ShardEngine.solve(self, t, fp, fc, f2, o2, reduce)
                    if t in self.stall_at:
                        self.stall_at.discard(t)
                        self.reruns += len(self.queue) - k - 1
                        redo = t
                        break
                self.queue = []
                return redo

        out = {}
        # the same sequence through an engine that only enqueues its frames and rewinds the driver three times (frames 3, 9 and 26; with
        # several ranks the checkpoints start 4 frames apart, the distance doubles behind every window without a stall and falls back to
        # 1 behind one -- every rank derives it from the stalls, which all ranks see alike): owned stacks, so the frame window has to
        # keep what a redo needs, across windows of 8 and 16 frames too
        T, H, W, r = 31, 36, 50, 2
        d = psfm_synth.synth_sequence(T, H, W, seed=44, sigma=0.25, n_occluders=2, stride2=True)
        n, n2 = T - 1, T - 2
        lo, hi = psfm_dist.shard_range(n, rank, world)
        lo2, hi2 = psfm_dist.shard_range(n2, rank, world)
        sl = lambda k, a, b: (torch.from_numpy(np.stack(d[k][a:b])) if b > a else torch.zeros((0, H, W, 2), dtype=torch.float32))
        eng = DeferredEngine([3, 9, 26])
        part = psfm_dist.connect_sharded(eng, sl("flows_f", lo, hi), sl("flows_b", lo, hi), sl("flows_f2", lo2, hi2),
                                         sl("flows_b2", lo2, hi2), 1.0, r, check, n_flows_total=n)
        birth, length, off, xy = psfm_dist.gather_result(part)
        _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
        O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        out["deferred"] = (bool(len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
                                and np.array_equal(xy, O.xy)),
                           [s["iterations"] for s in part["solve_stats"]] == [s["iterations"] for s in O.solves],
                           eng.reruns > 0 and not eng.stall_at and not eng.queue,
                           (eng.windows[0] <= 4 and max(eng.windows) >= 8 and max(eng.windows) <= 32) if world > 1 else True)
        out["windows"] = list(eng.windows)
        # (more ranks than stride-2 pairs in the last case: a rank with an EMPTY slice of a stack)
        cases = [(9, 38, 52, 2, 41, 0.3, 2, False), (8, 45, 60, 3, 42, 0.1, 1, True), (3, 30, 44, 1, 43, 0.2, 1, True)]
        if lean:        # (eight processes on a small box: the rewinding sequence above + one optimising case)
            cases = cases[1:2]
        for ci, (T, H, W, r, seed, sigma, nocc, optimize) in enumerate(cases):
            d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=True)
            n, n2 = T - 1, T - 2
            lo, hi = psfm_dist.shard_range(n, rank, world)
            lo2, hi2 = psfm_dist.shard_range(n2, rank, world)
            sl = lambda k, a, b: (torch.from_numpy(np.stack(d[k][a:b])) if b > a else torch.zeros((0, H, W, 2), dtype=torch.float32))
            # a rank only ever sees its own slice: the other frames are poisoned copies that would change the result if used
            part = psfm_dist.connect_sharded(orc.ShardEngine(), sl("flows_f", lo, hi), sl("flows_b", lo, hi),
                                             sl("flows_f2", lo2, hi2) if optimize else None, sl("flows_b2", lo2, hi2) if optimize else None,
                                             1.0, r, check, n_flows_total=n)
            birth, length, off, xy = psfm_dist.gather_result(part)
            _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
            if optimize:
                _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
                O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
            else:
                O = orc.track(d["flows_f"], occ, r)
            same = bool(len(birth) == O.n_traj and np.array_equal(birth, O.birth) and np.array_equal(length, O.length)
                        and np.array_equal(xy, O.xy))
            own = part["frames_read_from_own_slice"] == list(range(lo, hi))
            own2 = (not optimize) or part["stride2_frames_read_from_own_slice"] in (list(range(lo2, min(hi2, n - 1))), None)
            out[ci] = (same, own, own2, [s["iterations"] for s in part["solve_stats"]] == [s["iterations"] for s in O.solves])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_connect_sharded_with_frame_pair_owned_stacks(world):
    """the sharded mode without replicating the flows -- every rank passes only its Stage-A slice of the four
    stacks and receives Stage B's frames by broadcast two frames ahead.  Same trajectories, bit for bit, as the single-process
    oracle; a rank reads from its own slice exactly the frames it owns."""
    port = 32500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_owned_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for ci in (0, 1, 2, "deferred"):
            assert all(ret[r][ci]), (r, ci, ret[r][ci])


def test_connect_sharded_eight_ranks():
    """The target machine has 8 GPUs (BASELINE configs[3]): the exact single-sequence mode with EIGHT ranks over gloo -- grids whose
    row count is not a multiple of 8 (18 and 15 grid rows: bands of 3 / 2 rows), 13 frame pairs owned in slices of 2 and 1, an engine
    that only enqueues its frames and rewinds the driver twice (the ranks must agree on when, every 4 frames, without talking).
    Bit-identical to the single-process oracle."""
    world = 8
    port = 33500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_owned_worker, args=(world, port, ret, True), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for ci in (0, "deferred"):
            assert all(ret[r][ci]), (r, ci, ret[r][ci])


@pytest.mark.parametrize("world", [2, 3])
def test_connect_sharded_gloo(world):
    """ONE sequence over 2 and 3 ranks == the single-process oracle: ids / lengths / positions bit for bit, per-solve
    iteration counts and terminations equal; every rank only holds tracks born on its own row band."""
    port = 31500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        out = ret[r]
        for ci in (0, 1, 2):
            same_ids, err, stats_ok, own_band, n_local, n_total = out[ci]
            assert same_ids and err == 0.0 and stats_ok and own_band, (r, ci, out[ci])
            assert 0 < n_local < n_total
        assert out["alldie"], (r, "alldie")
        assert out["capacity_retry"], (r, "capacity retry: every rank reruns Stage B once, result equal to the oracle")


def test_world2_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert all(ret[r]), (r, ret[r])


def test_shard_range_and_bits():
    for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import psfm_dist
    for n in (0, 1, 5, 8, 100):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = psfm_dist.shard_range(n, r, w)
                assert 0 <= hi - lo <= (n + w - 1) // w
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    rng = np.random.default_rng(0)
    occ = torch.from_numpy(rng.uniform(size=(3, 7, 9)) < 0.4)
    pk = psfm_dist.pack_bits(occ)
    assert pk.shape == (3, 8) and pk.dtype == torch.uint8
    assert np.array_equal(pk.numpy(), np.packbits(occ.numpy().reshape(3, -1), axis=1, bitorder="little"))
    assert torch.equal(psfm_dist.unpack_bits(pk, 7, 9).bool(), occ)
