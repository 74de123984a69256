"""Many sequences: the loop of the reference driver (run_particlesfm.py:168-176, `--root_dir`) as a parallel map.

Sequences are the unit that shards exactly:
  * across GPUs  -- one process per GPU (torch.distributed / RCCL for nothing but the final barrier): rank r takes
    sequences r, r+world, ...;
  * inside a GPU -- the frame recurrence of ONE sequence is a chain of short dependent launches that leaves both
    the memory system and the SIMDs idle about half of the time, so `concurrency` sequences are processed at once on
    separate HIP streams with separate psfm contexts (measured on MI355X, 100-frame 1080p sequences: 1.26x the
    throughput with 2, 1.34x with 3);
  * small sequences (DAVIS / Sintel / ScanNet sizes: a frame fills 5-50 % of the device) -- `batch` of them per
    psfm_connect_batch call: ONE launch per frame for the whole batch, one checkpoint and one finalize for all of them
    (3.5-5x the sequences per second of one psfm_connect each on a DAVIS-sized stack, 2.3x with path consistency on a
    Sintel-sized one; profiles/r05).  Host threads still overlap the .flo ingest / track.npy writing of one batch with the
    compute of another.
"""
import threading

from . import _hip
from .main_connect_point_trajectories import main_connect_point_trajectories


BATCH_MAX = 64                 # PSFM_BATCH_MAX of csrc/psfm_batch.hip: sequences per psfm_connect_batch call
BATCH_BYTES = 48 << 30         # flow stacks of one group resident in HBM before its compute starts (of 288 GB)


def _connect_batch_to_disk(flow_dirs, traj_dirs, sample_ratio, flow_check_thres, traj_min_len, skip_path_consistency, skip_exists, layout):
    """main_connect_point_trajectories.py:27-62 for a group of sequences through psfm_connect_batch: ingest all, ONE batched
    compute call per frame size, then filter + save each from its own context."""
    import os
    from .utils import load_flows_device
    from .trajectory import run_connect_batch, result_to_trajectory_set, save_track_npy
    import glob
    from .main_connect_point_trajectories import main_connect_point_trajectories

    def flush(group):
        ctxs, infos = run_connect_batch([sq for _, sq in group], flow_check_thres, sample_ratio)
        for (td, _), ctx, info in zip(group, ctxs, infos):
            ts = result_to_trajectory_set(ctx, info, traj_min_len, reuse_pinned=True)
            save_track_npy(os.path.join(td, "track.npy"), ts, layout=layout)

    # Groups of one frame size go through psfm_connect_batch as they fill up: at most PSFM_BATCH_MAX (64) sequences and BATCH_BYTES of
    # flow stacks in HBM at a time (a group is ingested whole before its compute starts).  A directory without flows, or one whose
    # frames cannot be read here, goes through the one-sequence entry point, which reports / handles it by itself -- it must not take
    # the other sequences of the call with it.
    by_shape, held = {}, {}
    for fd, td in zip(flow_dirs, traj_dirs):
        os.makedirs(td, exist_ok=True)
        if skip_exists and os.path.exists(os.path.join(td, "track.npy")):
            continue
        if not glob.glob(os.path.join(fd, "flow_f", "*.flo")):
            main_connect_point_trajectories(fd, td, sample_ratio=sample_ratio, flow_check_thres=flow_check_thres, traj_min_len=traj_min_len,
                                            skip_path_consistency=skip_path_consistency, skip_exists=skip_exists, layout=layout)
            continue
        f, b = load_flows_device(os.path.join(fd, "flow_f")), load_flows_device(os.path.join(fd, "flow_b"))
        f2 = b2 = None
        if not skip_path_consistency:
            f2, b2 = load_flows_device(os.path.join(fd, "flow_f2")), load_flows_device(os.path.join(fd, "flow_b2"))
        key = (int(f.shape[1]), int(f.shape[2]))
        by_shape.setdefault(key, []).append((td, (f, b, f2, b2)))
        held[key] = held.get(key, 0) + sum(int(t.numel()) * 4 for t in (f, b, f2, b2) if t is not None)
        if len(by_shape[key]) >= BATCH_MAX or held[key] >= BATCH_BYTES:
            flush(by_shape.pop(key))
            held.pop(key)
        del f, b, f2, b2
    for group in by_shape.values():
        flush(group)


def connect_sequences(flow_dirs, traj_dirs, sample_ratio=2, flow_check_thres=1.0, traj_min_len=3,
                      skip_path_consistency=False, skip_exists=False, concurrency=2, rank=None, world=None, layout="csr", batch=1):
    """main_connect_point_trajectories for a list of sequences; returns the indices this rank processed.
    layout: pickle state of the track.npy files -- "csr" (default here: the batch driver is this package's own entry
    point and its files are read back through this package) or "reference" (files for an unmodified checkout).
    batch > 1: every worker takes up to `batch` sequences at a time through psfm_connect_batch (small frames: see above)."""
    import torch
    import psfm_dist
    if rank is None or world is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        else:
            rank, world = 0, 1
    assert len(flow_dirs) == len(traj_dirs)
    mine = psfm_dist.shard_sequences(len(flow_dirs), rank, world)
    device = torch.cuda.current_device()
    lock = threading.Lock()
    todo = list(mine)
    errors = []

    n_threads = max(1, min(int(concurrency), (len(mine) + max(int(batch), 1) - 1) // max(int(batch), 1)))

    def worker():
        torch.cuda.set_device(device)
        stream = torch.cuda.Stream(device=device)
        try:
            if n_threads > 1:
                # several sequences in flight on this GPU: no persistent frame loop (it needs the device to itself and would take the
                # device gate, i.e. serialise the sequences).  Solves that reject steps keep their ONE-launch form: every worker's
                # resident solves get an equal share of the device's co-resident block slots (psfm_ctx_set_resident_budget) and run
                # beside the other workers' -- 3-5x the launch chain on such flows.
                cs = _hip.batch_contexts(batch) if batch > 1 else [_hip.context()]
                share = max(cs[0].resident_capacity() // n_threads, 1) if not skip_path_consistency else 0
                for c in cs:
                    c.set_chain_mode(1)
                    c.set_resident_budget(share)
            with torch.cuda.stream(stream):
                while True:
                    with lock:
                        if not todo or errors:
                            break
                        if batch > 1:
                            ks = [todo.pop(0) for _ in range(min(int(batch), len(todo)))]
                        else:
                            k = todo.pop(0)
                    if batch > 1:
                        _connect_batch_to_disk([flow_dirs[k] for k in ks], [traj_dirs[k] for k in ks], sample_ratio, flow_check_thres,
                                               traj_min_len, skip_path_consistency, skip_exists, layout)
                        continue
                    main_connect_point_trajectories(flow_dirs[k], traj_dirs[k], sample_ratio=sample_ratio,
                                                    flow_check_thres=flow_check_thres, traj_min_len=traj_min_len,
                                                    skip_path_consistency=skip_path_consistency, skip_exists=skip_exists,
                                                    layout=layout)
                stream.synchronize()
        except Exception as e:   # surface the first failure in the caller
            with lock:
                errors.append(e)
        finally:
            _hip.release_thread_contexts()

    threads = [threading.Thread(target=worker) for _ in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return mine
