"""Multi-GPU plumbing for the point-trajectory path: one process per GPU, torch.distributed (RCCL on ROCm,
gloo on CPU for the tests).

What shards exactly:
  * whole sequences -- the unit the reference driver loops over (run_particlesfm.py:168-176).  No data-path
    collective; `shard_sequences` + `reduce_totals` are all that is needed (this is what bench.py --gpus N runs).
  * flow_check by frame pair (utils.py:94-105 is independent per pair) -- `flow_check_sharded`: every rank
    checks a contiguous slice of the pairs, then ONE all-gather of the bit-packed occlusion maps (H*W/8 bytes
    per pair: 259 KB at 1080p) gives every rank the full stack.  Bit-identical to the unsharded result.
  * the frame recurrence of ONE sequence -- `connect_sharded`: it has a loop-carried dependency (births at t+1 need
    every survivor of t; tails at t+1 are the optimised buffer of t), so frame ranges cannot be stitched exactly.  What
    does split exactly is the set of TRACKS: rank r owns the tracks born on its row band of the stride-r grid and runs
    every frame for them; per frame ONE all-reduce (max) of the grid-resolution `blocked` map (G bytes + a survivor
    byte: which grid points have a surviving track's pixel within distance r -- the EDT respawn rule) tells every rank
    where its band respawns; the path-consistency solve is block diagonal over tracks, its trust-region control needs
    only global sums (13 per iteration), all-gathered and combined in rank order on every rank (replicated control);
    ids follow from the keys (last valid time, birth frame, birth grid index) of all ranks.  Same trajectories, ids,
    lengths and positions as one process; every message is latency-bound (<= 0.5 MB), so for chain-only runs this is
    slower than one GPU -- it is the exact single-sequence mode north_star asks for, measured as such by bench.py.
"""
import os

import numpy as np


class TorchComm:
    """The collectives connect_sharded needs, over torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, gloo
    on CPU).  Tests that emulate several ranks inside one process pass an object with the same five members."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.gpu_only = self.on and dist.get_backend(group) == "nccl"      # RCCL moves device memory only
        self.cpu_only = self.on and not self.gpu_only                       # gloo: device tensors are staged through the host

    def all_reduce_max_(self, t):
        import torch.distributed as dist
        if self.world > 1:
            if self.cpu_only and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def all_gather_flat(self, t):
        """(n,) tensor -> (world * n,) tensor, rank-major."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return t.reshape(-1).clone()
        to_dev, to_host = self.gpu_only and not t.is_cuda, self.cpu_only and t.is_cuda
        dev = t.device
        if to_dev:
            t = t.cuda()
        elif to_host:
            t = t.cpu()
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous().reshape(-1), group=self.group)
        return out.to(dev) if (to_dev or to_host) else out

    def all_gather_object(self, obj):
        import torch.distributed as dist
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def broadcast_(self, t, src, async_op=False):
        """`t` of rank `src` (a rank of this group) into `t` on every rank.  Returns an object with .wait() (async_op: the
        transfer is in flight -- RCCL runs it on its own stream; wait() orders the caller's stream behind it)."""
        import torch.distributed as dist
        if self.world == 1:
            return _Done()
        gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
        if self.cpu_only and t.is_cuda:           # gloo moves host memory: stage (synchronous; tests only)
            h = t.cpu()
            dist.broadcast(h, src=gsrc, group=self.group)
            t.copy_(h)
            return _Done()
        w = dist.broadcast(t, src=gsrc, group=self.group, async_op=async_op)
        return w if async_op else _Done()


class _Done:
    def wait(self):
        return True


def _comm(comm, group=None):
    return comm if comm is not None else TorchComm(group)


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` of `world` (first n_items % world ranks get one more)."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sequences(n_sequences, rank, world):
    """Round-robin assignment of whole sequences to ranks."""
    return list(range(rank, int(n_sequences), int(world)))


def pack_bits(occ):
    """(n,H,W) bool/u8 torch tensor -> (n, ceil(H*W/8)) uint8, little-endian bit order, on the tensor's device."""
    import torch
    n, px = int(occ.shape[0]), int(occ.shape[1]) * int(occ.shape[2])
    flat = (occ.reshape(n, px) != 0).to(torch.uint8)       # (explicit sizes: an empty slice -- n == 0 -- has no "-1")
    pad = (-px) % 8
    if pad:
        flat = torch.nn.functional.pad(flat, (0, pad))
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=flat.device)
    return (flat.reshape(n, (px + pad) // 8, 8) * w).sum(-1, dtype=torch.int32).to(torch.uint8)


def unpack_bits(packed, H, W):
    """Inverse of pack_bits -> (n,H,W) uint8 0/1."""
    import torch
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=packed.device)
    bits = ((packed.unsqueeze(-1) & w) != 0).to(torch.uint8)
    n = int(packed.shape[0])
    return bits.reshape(n, int(packed.shape[1]) * 8)[:, :H * W].reshape(n, H, W)


def flow_check_sharded(flows_f, flows_b, thres, check_fn, group=None, comm=None, n_total=None):
    """Frame-pair-sharded flow_check with an all-gather stitch.

    flows_f / flows_b: (n,H,W,2) tensors present on every rank -- or, with n_total, ONLY this rank's slice
    [shard_range(n_total, rank, world)) of the n_total pairs (the stacks are owned by frame-pair shards);
    check_fn(f_slice, b_slice, thres) -> (k,H,W) uint8/bool tensor for a slice (on GPU:
    point_trajectory.utils.flow_check_device; in the CPU tests: the oracle).  Returns the full (n,H,W) uint8 stack
    on every rank."""
    import torch
    comm = _comm(comm, group)
    world, rank = comm.world, comm.rank
    H, W = int(flows_f.shape[1]), int(flows_f.shape[2])
    if n_total is None:
        n = int(flows_f.shape[0])
        lo, hi = shard_range(n, rank, world)
        f_mine, b_mine = flows_f[lo:hi], flows_b[lo:hi]
    else:
        n = int(n_total)
        lo, hi = shard_range(n, rank, world)
        assert int(flows_f.shape[0]) == hi - lo == int(flows_b.shape[0]), "this rank's slice of the pairs expected"
        f_mine, b_mine = flows_f, flows_b
    mine = check_fn(f_mine, b_mine, thres) if hi > lo else torch.zeros((0, H, W), dtype=torch.uint8, device=flows_f.device)
    if world == 1:
        return mine.to(torch.uint8)
    per = (n + world - 1) // world                       # all_gather needs equal shapes: pad the short shards
    nbytes = (H * W + 7) // 8
    buf = torch.zeros((per, nbytes), dtype=torch.uint8, device=mine.device)
    if hi > lo:
        buf[:hi - lo] = pack_bits(mine)
    out = comm.all_gather_flat(buf.reshape(-1)).reshape(world * per, nbytes)
    parts = []
    for r in range(world):
        l, h = shard_range(n, r, world)
        parts.append(out[r * per:r * per + (h - l)])
    return unpack_bits(torch.cat(parts, 0), H, W)


def reduce_totals(seconds, units, device=None, group=None):
    """bench.py's reduction: max over ranks of the elapsed time, sum over ranks of the processed units."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_backend(group) == "gloo":
        device = None                      # (gloo moves host memory)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), float(u.item())


# ---------------------------------------------------------------------------------------------------------------
# ONE sequence over several ranks, exactly (SURVEY.md 8e Stage A + Stage B)
# ---------------------------------------------------------------------------------------------------------------
def band_range(grid_h, grid_w, rank, world):
    """Grid points [g0, g1) whose births `rank` owns: a contiguous band of whole grid rows (row-major grid indices)."""
    lo, hi = shard_range(grid_h, rank, world)
    return lo * int(grid_w), hi * int(grid_w)


def make_reduce(group=None, comm=None):
    """reduce(vals, is_max): vals (1-D float64 tensor) = this rank's part on entry, the value over all ranks on return.
    One all-gather; sums are added in rank order on every rank, so every rank holds the same bits."""
    import torch
    comm = _comm(comm, group)
    world = comm.world

    def reduce(vals, is_max):
        if world == 1:
            return vals
        parts = comm.all_gather_flat(vals.reshape(-1)).reshape((world,) + tuple(vals.shape))
        mx = torch.as_tensor(is_max, dtype=torch.bool, device=vals.device)
        tot = parts[0].clone()
        for r in range(1, world):
            tot = torch.where(mx, torch.maximum(tot, parts[r]), tot + parts[r])
        vals.copy_(tot)
        return vals
    return reduce


KEY_TIME_BITS, KEY_GRID_BITS = 16, 31       # the packed id key: last << 47 | birth << 31 | grid index


def global_ids(birth, length, first_xy, n_flows, ratio, grid_w, group=None, comm=None):
    """Ids of this rank's trajectories in the order of the single-process run: rank of the key (last valid time, birth
    frame, birth grid index) among the keys of all ranks (SURVEY a-17: dead tracks by frame, then the still active ones,
    each group in active-list order = (birth frame, grid index)).  Returns (ids int64, total number of trajectories)."""
    import torch
    comm = _comm(comm, group)
    birth = np.asarray(birth, np.int64)
    last = birth + np.asarray(length, np.int64) - 1
    gidx = (np.asarray(first_xy[:, 1], np.int64) // ratio) * int(grid_w) + np.asarray(first_xy[:, 0], np.int64) // ratio
    key = (last << (KEY_GRID_BITS + KEY_TIME_BITS)) | (birth << KEY_GRID_BITS) | gidx         # 16 + 16 + 31 bits (csrc/psfm_shard.hip)
    if n_flows + 2 >= (1 << KEY_TIME_BITS) or (len(gidx) and int(gidx.max()) >= (1 << KEY_GRID_BITS)):
        raise ValueError("global_ids: %d flows / grid index %d do not fit the packed (last, birth, grid) key" % (n_flows, int(gidx.max()) if len(gidx) else 0))
    assert len(key) == 0 or (np.diff(key) > 0).all(), "local trajectories must come in key order"
    world = comm.world
    if world == 1:
        return np.arange(len(key), dtype=np.int64), len(key)
    cnts = comm.all_gather_flat(torch.tensor([len(key)], dtype=torch.int64)).tolist()
    nmax = max(max(cnts), 1)
    pad = torch.full((nmax,), np.iinfo(np.int64).max, dtype=torch.int64)
    pad[:len(key)] = torch.from_numpy(key)
    allk = comm.all_gather_flat(pad).reshape(world, nmax).numpy()
    ids = np.zeros(len(key), np.int64)
    for r in range(world):
        ids += np.searchsorted(allk[r][:cnts[r]], key, side="left")
    return ids, int(sum(cnts))


def global_ids_device(keys, comm):
    """global_ids on device tensors: keys (n,) int64 ascending (HipShardEngine.finish_device) -> (ids (n,) int64 on the same
    device, total number of trajectories).  One all-gather of the counts, one of the padded keys."""
    import torch
    world = comm.world
    if world == 1:
        return torch.arange(keys.numel(), dtype=torch.int64, device=keys.device), int(keys.numel())
    cnts = comm.all_gather_flat(torch.tensor([keys.numel()], dtype=torch.int64, device=keys.device)).tolist()
    nmax = max(max(cnts), 1)
    pad = torch.full((nmax,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=keys.device)
    pad[:keys.numel()] = keys
    allk = comm.all_gather_flat(pad).reshape(world, nmax)
    ids = torch.zeros(keys.numel(), dtype=torch.int64, device=keys.device)
    for r in range(world):
        ids += torch.searchsorted(allk[r][:cnts[r]].contiguous(), keys, right=False)
    return ids, int(sum(cnts))


class FrameWindow:
    """Stage B's view of a frame stack that is OWNED by frame-pair slices (Stage A's shards): frame k lives on
    rank owner(k) only; every rank receives it by broadcast `ahead` frames before the recurrence needs it -- in flight
    behind the frames being computed -- and drops it once the recurrence is past it.  Per rank: its slice of the stack
    plus ahead + 2 frames, instead of the whole stack.  world == 1: the local stack itself.
    `touched` records the frames this rank read from its OWN slice (tests: never a frame of another owner)."""

    def __init__(self, local, n_total, comm, ahead=2):
        self.local, self.n, self.comm, self.ahead = local, int(n_total), comm, int(ahead)
        self.lo, self.hi = shard_range(self.n, comm.rank, comm.world)
        assert int(local.shape[0]) == self.hi - self.lo, "this rank's slice of the frames expected"
        self.bounds = [shard_range(self.n, r, comm.world) for r in range(comm.world)]
        self.live, self.pool, self.touched = {}, [], []

    def owner(self, k):
        for r, (lo, hi) in enumerate(self.bounds):
            if lo <= k < hi:
                return r
        raise IndexError(k)

    def _request(self, k):
        import torch
        if k in self.live or not (0 <= k < self.n):
            return
        src = self.owner(k)
        buf = self.pool.pop() if self.pool else torch.empty(tuple(self.local.shape[1:]), dtype=self.local.dtype,
                                                            device=self.local.device)
        if src == self.comm.rank:
            buf.copy_(self.local[k - self.lo])
            self.touched.append(k)
        self.live[k] = (buf, self.comm.broadcast_(buf, src, async_op=True))

    def get(self, k):
        """frame k (requests k .. k + ahead on the way; every rank calls this with the same k in the same order)"""
        if self.comm.world == 1:
            return self.local[k]
        for j in range(k, min(self.n, k + self.ahead + 1)):
            self._request(j)
        buf, work = self.live[k]
        work.wait()
        return buf

    def release_below(self, k):
        for j in [j for j in self.live if j < k]:
            buf, work = self.live.pop(j)
            work.wait()
            self.pool.append(buf)


def _any_stalled(engine, comm):
    """Has a solve enqueued since the last checkpoint stalled on this rank's device?  The ranks must agree on when to
    checkpoint (it contains collectives), and the flag reaches their hosts at different times: world == 1 peeks, several ranks
    keep the fixed cadence."""
    return comm.world == 1 and hasattr(engine, "stalled") and engine.stalled()


def connect_sharded(engine, flows_f, flows_b, flows_f2, flows_b2, thres, sample_ratio, check_fn, group=None, comm=None,
                    keep_on_device=False, n_flows_total=None):
    """main_connect_point_trajectories.py:36-53 for ONE sequence on all ranks of `group`, exactly.

    flows_*: (n,H,W,2) float32 tensors on every rank (flows_f2 / flows_b2 None: track() instead of track_optimize());
    check_fn(f, b, thres) -> (k,H,W) uint8: flow_check of a slice (Stage A, frame-pair shards + all-gather);
    engine: this rank's share of the recurrence -- point_trajectory.shard.HipShardEngine on a GPU (RCCL),
    the engine of the CPU oracle in the gloo tests:
        begin(n_flows, H, W, ratio, g0, g1, optimize)
        step(t, flow_t, occ_t) -> uint8 tensor (G marks of this rank's survivors + 1 survivor byte), exchanged here
        after_exchange(t, x); solve(t, flow_{t-1}, flow_t, flow2_{t-1}, occ2_{t-1}, reduce); finish() -> CSR + stats
    n_flows_total: the four stacks are OWNED by frame-pair slices -- every rank passes only its slice
    [shard_range(n, rank, world)) of each stack (n = n_flows_total pairs, n - 1 stride-2 pairs;
    point_trajectory.utils.load_flows_device_slice(dir, rank, world) reads exactly that); Stage B's frames arrive by broadcast (FrameWindow).  Per-rank HBM for the
    stacks: 1 / world of the sequence + a window of a few frames.
    Returns {"birth","length","off","xy": this rank's trajectories; "ids": their ids in the single-process order;
    "n_traj": trajectories over all ranks; "solve_stats"; "occ","occ2"}."""
    comm = _comm(comm, group)
    world, rank = comm.world, comm.rank
    optimize = flows_f2 is not None
    owned = n_flows_total is not None
    n_flows, H, W = int(n_flows_total if owned else flows_f.shape[0]), int(flows_f.shape[1]), int(flows_f.shape[2])
    n2 = max(n_flows - 1, 0)
    r = int(sample_ratio)
    GW, GH = (W + r - 1) // r, (H + r - 1) // r
    # ---- Stage A: occlusion maps, frame-pair shards + one all-gather per stack ----
    occ = flow_check_sharded(flows_f, flows_b, thres, check_fn, comm=comm, n_total=n_flows if owned else None)
    occ2 = (flow_check_sharded(flows_f2, flows_b2, thres, check_fn, comm=comm, n_total=n2 if owned else None)
            if optimize and n2 > 0 else None)
    # (Stage A stays up front also at world size 1: in chunks on a side stream, a chunk ahead of the recurrence -- the way the one-GPU call
    # hides two thirds of its flow_check -- ONE sequence of configs[3] took 40.3-41.9 ms instead of 38.1-38.3: profiles/EXPERIMENTS.md 6.6;
    # round 5, with one launch per frame and no control launch behind it: 37.6-38.6 ms in chunks of 8 / 16 / 32 pairs, 37.5-37.9 up front: 7.8)
    # ---- Stage B: the recurrence, tracks split by birth row band ----
    # n_flows_total given: the four stacks are owned by Stage A's frame-pair shards (every rank passed its slice only); the
    # forward stacks reach the other ranks frame by frame, broadcast from their owner two frames ahead of the recurrence
    g0, g1 = band_range(GH, GW, rank, world)
    if hasattr(engine, "set_local"):     # one rank: nothing of a solve is exchanged -- rejecting solves take the one-GPU call's forms
        engine.set_local(world == 1)
    # A rank whose lane / record tables run full learns it when it finalizes (PSFM_ERR_CAPACITY from psfm_shard_finish: the launches behind
    # an overflow are harmless by themselves, csrc/psfm_solver.hip clamps what they count on).  That is behind the last collective of the
    # recurrence, so the ranks can agree on it: every rank reports, and if ANY table was too small ALL of them run Stage B again with
    # the tables doubled / quadrupled -- what run_connect does for the one-GPU call (trajectory.py), collectively.
    for attempt in range(6):
        wf = FrameWindow(flows_f, n_flows, comm) if owned else None
        w2 = FrameWindow(flows_f2, n2, comm) if owned and optimize and n2 > 0 else None
        engine.begin(n_flows, H, W, r, g0, g1, optimize)
        if world > 1 and optimize and hasattr(engine, "connect_peers"):
            # several ranks: solves that reject steps run as ONE resident launch per rank whose all-reduce crosses the ranks on the device
            # (peer-mapped granule rows) instead of export -> all-gather -> control per trust-region iteration with the host polling
            engine.connect_peers(comm)
        try:
            return _stage_b(engine, comm, flows_f, flows_f2, occ, occ2, n_flows, n2, H, W, r, GW, optimize, owned, wf, w2, g0, g1, keep_on_device)
        except _CapacityRetry:
            if hasattr(engine, "abort"):
                engine.abort()
            if attempt == 5 or not hasattr(engine, "grow_tables"):
                raise RuntimeError("connect_sharded: lane / trajectory tables still too small after %d enlargements" % attempt)
            engine.grow_tables()
        except BaseException:
            if hasattr(engine, "abort"):         # (gives back what the engine took for the run: enqueued solves, its resident budget)
                engine.abort()
            raise


class _CapacityRetry(Exception):
    """some rank's tables ran full: every rank leaves Stage B with this and runs it again with larger ones"""


def _finish_agreed(engine, comm, fn):
    """fn() = the engine's finalize on this rank; PSFM_ERR_CAPACITY there is shared with the other ranks before anybody goes on"""
    out, full = None, 0
    try:
        out = fn()
    except RuntimeError as e:
        if getattr(e, "status", None) != 3:      # (_hip.PSFM_ERR_CAPACITY; the CPU engines of the gloo tests have no tables to overflow)
            raise
        full = 1
    if any(comm.all_gather_object(full)) if comm.world > 1 else full:
        raise _CapacityRetry()
    return out


def _stage_b(engine, comm, flows_f, flows_f2, occ, occ2, n_flows, n2, H, W, r, GW, optimize, owned, wf, w2, g0, g1, keep_on_device):
    """Stage B of connect_sharded: the recurrence, tracks split by birth row band (engine.begin() has run)."""
    world = comm.world
    reduce = make_reduce(comm=comm)
    # Engines that only ENQUEUE a frame's solve (the HIP engine: no host round trip per solve) are asked every CHECK frames whether
    # one of them did not go as speculated; they redo it there and the frames behind it -- no-ops on the device since -- run again.
    # Every rank sees the same stall (the control step is replicated on the same totals), so the ranks rewind together.
    has_ck = optimize and hasattr(engine, "checkpoint")
    merged = has_ck and hasattr(engine, "frame") and os.environ.get("PSFM_SHARD_MERGED", "1") != "0"
    if has_ck:
        # one rank sees a stall early (_any_stalled: the flag follows every control step into pinned memory), so its window is the
        # sequence -- every synchronisation drains the launch queue, and the speculated K only moves at a checkpoint (16 frames per
        # window: 39.0 ms for configs[3], 7 solves redone; 64: 39.4; the sequence: 37.8, none).  Several ranks must agree on when
        # to rewind without talking: a window every rank derives from what every rank has seen -- the stalls, which are the same
        # everywhere because the control step runs on the same totals: 4 frames to start with, ONE behind a redone solve (the frames
        # enqueued behind a stall are no-ops, but each of them still costs its launches and exchanges -- on flows whose every solve
        # rejects steps a longer window would run each frame several times), doubled after every window without a stall, up to 32.
        fixed_window = int(os.environ.get("PSFM_SHARD_CHECK_EVERY", "0"))
        # (one rank: at most 128 frames per window -- the engine keeps the tensors of every enqueued frame alive for a redo and FrameWindow
        # keeps the frames since the last confirmed one: O(window), not O(sequence), for 1000-frame inputs)
        engine.check_every = fixed_window or (min(max(16, n_flows), 128) if world == 1 else 4)
    confirmed = 0                  # frames below this are final
    since = 0                      # frames enqueued since the last checkpoint
    t = 0
    while t < n_flows:
        f_t = wf.get(t) if owned else flows_f[t]
        if merged and t + 1 >= 2:
            # step(t) + the fused export of solve(t) as one launch (the solve of frame t needs the own tracks only: it does not
            # wait for the marks of step t), the two exchanges behind it
            f_prev = wf.get(t - 1) if owned else flows_f[t - 1]
            f2_prev = w2.get(t - 1) if owned else flows_f2[t - 1]
            engine.frame(t, f_prev, f_t, occ[t], f2_prev, occ2[t - 1], comm.all_reduce_max_, reduce)
        else:
            x = engine.step(t, f_t, occ[t])                                      # track.py:33-47 for the own tracks
            comm.all_reduce_max_(x)                                              # marks of every rank's survivors
            engine.after_exchange(t, x)
            if optimize and t + 1 >= 2:                                          # track_optimize.py:49-50
                f_prev = wf.get(t - 1) if owned else flows_f[t - 1]
                f2_prev = w2.get(t - 1) if owned else flows_f2[t - 1]
                engine.solve(t, f_prev, f_t, f2_prev, occ2[t - 1], reduce)
        since += 1
        # (every 16 frames -- or as soon as the engine sees, without synchronising, that a solve of the window has stalled)
        if has_ck and (since >= engine.check_every or t == n_flows - 1 or _any_stalled(engine, comm) or
                       (hasattr(engine, "window_full") and engine.window_full())):
            redone = engine.checkpoint(reduce)
            if redone is not None:
                t = redone                   # frames redone + 1 .. are run again
            confirmed = t + 1
            since = 0
            if world > 1 and not fixed_window:
                engine.check_every = 1 if redone is not None else min(32, 2 * engine.check_every)
        elif not has_ck:
            confirmed = t + 1
        if owned:                            # keep what a redo of an unconfirmed frame needs: its flow01 is frame - 1
            keep = min(t, confirmed - 1)
            wf.release_below(keep)           # (frame t is the next solve's flow01)
            if w2 is not None:
                w2.release_below(keep)
        t += 1
    if keep_on_device:      # the trajectories stay in the engine's HBM (psfm_result_device); only their ids are formed
        info, keys = _finish_agreed(engine, comm, lambda: engine.finish_device(r, W))
        ids, n_traj = global_ids_device(keys, comm)
        return {"info": info, "ids": ids, "n_traj": n_traj, "n_points_local": int(info.n_points), "n_solves": int(info.n_solves),
                "solver_iterations": int(info.solver_iterations), "occ": occ, "occ2": occ2, "band": (g0, g1),
                "frames_read_from_own_slice": (sorted(wf.touched) if owned else None)}
    birth, length, off, xy, stats = _finish_agreed(engine, comm, engine.finish)
    first = xy[off[:-1]] if len(birth) else np.zeros((0, 2))
    ids, n_traj = global_ids(birth, length, first, n_flows, r, GW, comm=comm)
    return {"birth": birth, "length": length, "off": off, "xy": xy, "ids": ids, "n_traj": n_traj, "solve_stats": stats,
            "occ": occ, "occ2": occ2, "band": (g0, g1), "frames_read_from_own_slice": (sorted(wf.touched) if owned else None),
            "stride2_frames_read_from_own_slice": (sorted(w2.touched) if w2 is not None else None)}


def gather_result(part, group=None, comm=None):
    """The whole sequence's CSR in id order on every rank (tests, small runs): all-gather of the per-rank parts."""
    comm = _comm(comm, group)
    parts = comm.all_gather_object({k: part[k] for k in ("birth", "length", "off", "xy", "ids")})
    n = part["n_traj"]
    birth = np.zeros(n, np.int32); length = np.zeros(n, np.int32)
    for p in parts:
        birth[p["ids"]] = p["birth"]; length[p["ids"]] = p["length"]
    off = np.zeros(n + 1, np.int64)
    np.cumsum(length, out=off[1:])
    xy = np.zeros((int(off[-1]), 2), np.float64)
    for p in parts:
        for j, i in enumerate(p["ids"]):
            xy[off[i]:off[i + 1]] = p["xy"][p["off"][j]:p["off"][j + 1]]
    return birth, length, off, xy
