"""Property tests (hypothesis) of the host-side pieces of the multi-rank mode (particle-sfm_amd/psfm_dist.py) that do not need
a GPU or a process group: the partitions, the bit packing of the occlusion maps, the id ranking over ranks, the rank-ordered
reduction and the frame window -- several "ranks" are threads exchanging through barriers (tests/_thread_comm.py)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import psfm_dist
from _thread_comm import run_ranks

FAST = settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])


@FAST
@given(n=st.integers(0, 5000), world=st.integers(1, 17))
def test_shard_range_is_a_balanced_partition(n, world):
    r = [psfm_dist.shard_range(n, k, world) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == n
    assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))                 # contiguous, in rank order
    sizes = [hi - lo for lo, hi in r]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert sorted(sum((psfm_dist.shard_sequences(n, k, world) for k in range(world)), [])) == list(range(n))


@FAST
@given(gh=st.integers(1, 300), gw=st.integers(1, 300), world=st.integers(1, 9))
def test_band_range_hands_out_whole_grid_rows(gh, gw, world):
    b = [psfm_dist.band_range(gh, gw, k, world) for k in range(world)]
    assert b[0][0] == 0 and b[-1][1] == gh * gw
    for k, (g0, g1) in enumerate(b):
        assert g0 % gw == 0 and g1 % gw == 0 and g0 <= g1
        if k:
            assert b[k - 1][1] == g0


@FAST
@given(n=st.integers(0, 4), h=st.integers(1, 13), w=st.integers(1, 21), seed=st.integers(0, 2**31 - 1))
def test_pack_bits_round_trip(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    occ = (torch.rand((n, h, w), generator=g) < 0.4).to(torch.uint8) * 255        # any non-zero byte counts as occluded
    p = psfm_dist.pack_bits(occ)
    assert p.dtype == torch.uint8 and tuple(p.shape) == (n, (h * w + 7) // 8)
    back = psfm_dist.unpack_bits(p, h, w)
    assert torch.equal(back, (occ != 0).to(torch.uint8))


def _random_trajectories(rng, n_flows, gh, gw, ratio, n):
    """n trajectories with distinct keys (last valid time, birth frame, birth grid index), as a single-process run orders them"""
    keys = set()
    while len(keys) < n:
        b = int(rng.integers(0, n_flows + 1))
        last = int(rng.integers(b, n_flows + 1))
        keys.add((last, b, int(rng.integers(0, gh * gw))))
    keys = sorted(keys)
    last, birth, g = (np.array([k[j] for k in keys], np.int64) for j in range(3))
    first = np.stack([(g % gw) * ratio, (g // gw) * ratio], 1).astype(np.float64).reshape(-1, 2)
    return birth, last - birth + 1, first, g


@FAST
@given(seed=st.integers(0, 2**31 - 1), world=st.integers(1, 4), n=st.integers(0, 200), ratio=st.integers(1, 4))
def test_global_ids_rank_the_keys_of_all_ranks(seed, world, n, ratio):
    """Every rank holds the trajectories born in its band, in key order; their ids must be their positions in the order of the
    single-process run -- on the host (global_ids) and on tensors (global_ids_device)."""
    rng = np.random.default_rng(seed)
    n_flows, gh, gw = 12, 9, 7
    birth, length, first, g = _random_trajectories(rng, n_flows, gh, gw, ratio, min(n, (n_flows + 1) * gh * gw // 4))

    def rank_fn(comm):
        g0, g1 = psfm_dist.band_range(gh, gw, comm.rank, comm.world)
        mine = np.nonzero((g >= g0) & (g < g1))[0]
        ids, tot = psfm_dist.global_ids(birth[mine], length[mine], first[mine], n_flows, ratio, gw, comm=comm)
        last = birth[mine] + length[mine] - 1
        keys = torch.from_numpy((last << 47) | (birth[mine] << 31) | g[mine])
        ids_d, tot_d = psfm_dist.global_ids_device(keys, comm)
        return mine, ids, tot, ids_d.numpy(), tot_d

    for mine, ids, tot, ids_d, tot_d in run_ranks(world, rank_fn):
        assert tot == tot_d == len(birth)
        assert np.array_equal(ids, mine) and np.array_equal(ids_d, mine)


def test_global_ids_of_a_three_thousand_flow_sequence_on_eight_ranks():
    """The packed (last, birth, grid) key holds 16 bits per time field (round 3: 11 -- sharded runs refused sequences beyond 2045
    flows): 3000 flows on a 1080p / sample_ratio 2 grid (518 400 points: 20 bits of the 31), eight ranks; 65 534 flows do not fit."""
    rng = np.random.default_rng(7)
    n_flows, gh, gw, ratio = 3000, 540, 960, 2
    birth, length, first, g = _random_trajectories(rng, n_flows, gh, gw, ratio, 4000)
    assert (birth + length - 1).max() > 2046

    def rank_fn(comm):
        g0, g1 = psfm_dist.band_range(gh, gw, comm.rank, comm.world)
        mine = np.nonzero((g >= g0) & (g < g1))[0]
        ids, tot = psfm_dist.global_ids(birth[mine], length[mine], first[mine], n_flows, ratio, gw, comm=comm)
        return mine, ids, tot

    for mine, ids, tot in run_ranks(8, rank_fn):
        assert tot == len(birth) and np.array_equal(ids, mine)
    with pytest.raises(ValueError):
        psfm_dist.global_ids(birth[:4], length[:4], first[:4], 65534, ratio, gw, comm=psfm_dist._comm(None, None))


@FAST
@given(seed=st.integers(0, 2**31 - 1), world=st.integers(1, 4), k=st.integers(1, 8))
def test_reduce_adds_in_rank_order_and_takes_maxima(seed, world, k):
    rng = np.random.default_rng(seed)
    vals = rng.normal(size=(world, k * 13)) * 10.0 ** rng.integers(-8, 8, size=(world, k * 13))
    mask = [(i % 13) == 5 for i in range(k * 13)]

    def rank_fn(comm):
        v = torch.from_numpy(vals[comm.rank].copy())
        psfm_dist.make_reduce(comm=comm)(v, mask)
        return v.numpy()

    out = run_ranks(world, rank_fn)
    want = vals[0].copy()
    for r in range(1, world):
        want = np.where(mask, np.maximum(want, vals[r]), want + vals[r])          # ((v0 + v1) + v2) ...: the same bits everywhere
    for o in out:
        assert np.array_equal(o, want)


@FAST
@given(seed=st.integers(0, 2**31 - 1), world=st.integers(1, 4), n=st.integers(1, 23))
def test_frame_window_serves_every_frame_from_its_owner(seed, world, n):
    """The recurrence walks the frames forward, sometimes steps back to an unconfirmed frame (a redone solve) and releases what it
    is past; every rank must see frame k's content whoever owns it, read only its own frames from its slice, and recycle buffers."""
    rng = np.random.default_rng(seed)
    frames = torch.from_numpy(rng.integers(0, 1 << 20, size=(n, 3, 2)).astype(np.float32))
    # the walk: (frame asked for, release bound) -- the same on every rank
    walk, t, confirmed = [], 0, 0
    while t < n:
        walk.append((t, min(t, max(confirmed - 1, 0))))
        if rng.random() < 0.15 and t > confirmed:
            t = int(rng.integers(confirmed, t + 1))          # rewind to an unconfirmed frame
            continue
        if rng.random() < 0.3:
            confirmed = t + 1
        t += 1

    def rank_fn(comm):
        lo, hi = psfm_dist.shard_range(n, comm.rank, comm.world)
        win = psfm_dist.FrameWindow(frames[lo:hi].clone(), n, comm)
        seen, peak = [], 0
        for k, keep in walk:
            seen.append(win.get(k).clone())
            if k >= 1:
                seen.append(win.get(k - 1).clone())          # (the solver's flow01)
            peak = max(peak, len(win.live))
            win.release_below(max(keep - 1, 0))
        return seen, sorted(set(win.touched)), (lo, hi), peak, len(win.pool) + len(win.live)

    for seen, touched, (lo, hi), peak, buffers in run_ranks(world, rank_fn):
        i = 0
        for k, _ in walk:
            assert torch.equal(seen[i], frames[k]); i += 1
            if k >= 1:
                assert torch.equal(seen[i], frames[k - 1]); i += 1
        if world > 1:
            assert all(lo <= k < hi for k in touched)
            assert buffers <= n                                    # buffers are recycled, never one per request


class _RecordingEngine:
    """the engine protocol of connect_sharded with nothing behind it: records the calls, raises where it is told to"""

    def __init__(self, G, fail_at=None):
        self.G, self.fail_at, self.calls, self.local = G, fail_at, [], None

    def set_local(self, on):
        self.local = bool(on)

    def begin(self, *a):
        self.calls.append("begin")

    def step(self, t, flow, occ):
        if t == self.fail_at:
            raise RuntimeError("frame %d failed" % t)
        self.calls.append("step%d" % t)
        return torch.zeros(self.G + 1, dtype=torch.uint8)

    def after_exchange(self, t, x):
        pass

    def abort(self):
        self.calls.append("abort")

    def finish(self):
        self.calls.append("finish")
        return (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros((0, 2)), [])


@pytest.mark.parametrize("world", [1, 2])
def test_connect_sharded_tells_the_engine_whether_it_is_alone_and_aborts_it_on_an_exception(world):
    """connect_sharded's contract with an engine beyond the frame calls: set_local(world == 1) before begin() (one rank exchanges nothing:
    the GPU engine then takes the one-GPU call's solver forms) and abort() when the run ends in an exception -- on every rank, before the
    exception leaves the call (the GPU engine gives back its resident budget and drops the solves it had enqueued)."""
    T, H, W, r = 5, 12, 16, 2
    flows = torch.zeros((T, H, W, 2), dtype=torch.float32)
    check = lambda f, b, thres: torch.zeros(f.shape[:3], dtype=torch.uint8)
    G = ((W + r - 1) // r) * ((H + r - 1) // r)

    def rank_fn(comm):
        ok = _RecordingEngine(G)
        psfm_dist.connect_sharded(ok, flows, flows, None, None, 1.0, r, check, comm=comm)
        bad = _RecordingEngine(G, fail_at=2)
        with pytest.raises(RuntimeError, match="frame 2 failed"):
            psfm_dist.connect_sharded(bad, flows, flows, None, None, 1.0, r, check, comm=comm)
        return ok, bad

    for ok, bad in run_ranks(world, rank_fn):
        assert ok.local is (world == 1) and ok.calls == ["begin"] + ["step%d" % t for t in range(T)] + ["finish"]
        assert bad.local is (world == 1) and bad.calls == ["begin", "step0", "step1", "abort"]
