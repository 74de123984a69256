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
