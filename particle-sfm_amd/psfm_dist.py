"""Multi-GPU plumbing for the point-trajectory path: one process per GPU, torch.distributed (RCCL on ROCm,
gloo on CPU for the tests).

What shards exactly:
  * whole sequences -- the unit the reference driver loops over (run_particlesfm.py:168-176).  No data-path
    collective; `shard_sequences` + `reduce_totals` are all that is needed (this is what bench.py --gpus N runs).
  * flow_check by frame pair (utils.py:94-105 is independent per pair) -- `flow_check_sharded`: every rank
    checks a contiguous slice of the pairs, then ONE all-gather of the bit-packed occlusion maps (H*W/8 bytes
    per pair: 259 KB at 1080p) gives every rank the full stack.  Bit-identical to the unsharded result.
The frame recurrence itself has a loop-carried dependency (births at t+1 need every survivor of t), so it is
NOT split across ranks; a rank that needs it for a sequence runs it whole.
"""
import numpy as np


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
    n = occ.shape[0]
    flat = (occ.reshape(n, -1) != 0).to(torch.uint8)
    pad = (-flat.shape[1]) % 8
    if pad:
        flat = torch.nn.functional.pad(flat, (0, pad))
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=flat.device)
    return (flat.reshape(n, -1, 8) * w).sum(-1, dtype=torch.int32).to(torch.uint8)


def unpack_bits(packed, H, W):
    """Inverse of pack_bits -> (n,H,W) uint8 0/1."""
    import torch
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=packed.device)
    bits = ((packed.unsqueeze(-1) & w) != 0).to(torch.uint8)
    return bits.reshape(packed.shape[0], -1)[:, :H * W].reshape(-1, H, W)


def flow_check_sharded(flows_f, flows_b, thres, check_fn, group=None):
    """Frame-pair-sharded flow_check with an all-gather stitch.

    flows_f / flows_b: (n,H,W,2) tensors present on every rank (or at least this rank's slice valid);
    check_fn(f_slice, b_slice, thres) -> (k,H,W) uint8/bool tensor for a slice (on GPU:
    point_trajectory.utils.flow_check_device; in the CPU tests: the oracle).  Returns the full (n,H,W) uint8 stack
    on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n, H, W = int(flows_f.shape[0]), int(flows_f.shape[1]), int(flows_f.shape[2])
    lo, hi = shard_range(n, rank, world)
    mine = check_fn(flows_f[lo:hi], flows_b[lo:hi], thres) if hi > lo else torch.zeros((0, H, W), dtype=torch.uint8,
                                                                                       device=flows_f.device)
    if world == 1:
        return mine.to(torch.uint8)
    per = (n + world - 1) // world                       # all_gather needs equal shapes: pad the short shards
    nbytes = (H * W + 7) // 8
    buf = torch.zeros((per, nbytes), dtype=torch.uint8, device=mine.device)
    if hi > lo:
        buf[:hi - lo] = pack_bits(mine)
    out = torch.empty((world * per, nbytes), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        l, h = shard_range(n, r, world)
        parts.append(out[r * per:r * per + (h - l)])
    return unpack_bits(torch.cat(parts, 0), H, W)


def reduce_totals(seconds, units, device=None, group=None):
    """bench.py's reduction: max over ranks of the elapsed time, sum over ranks of the processed units."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), float(u.item())
