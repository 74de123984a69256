"""Many sequences: the loop of the reference driver (run_particlesfm.py:168-176, `--root_dir`) as a parallel map.

Sequences are the unit that shards exactly:
  * across GPUs  -- one process per GPU (torch.distributed / RCCL for nothing but the final barrier): rank r takes
    sequences r, r+world, ...;
  * inside a GPU -- the frame recurrence of ONE sequence is a chain of short dependent launches that leaves both
    the memory system and the SIMDs idle about half of the time, so `concurrency` sequences are processed at once on
    separate HIP streams with separate psfm contexts (measured on MI355X, 100-frame 1080p sequences: 1.26x the
    throughput with 2, 1.34x with 3).
"""
import threading

from . import _hip
from .main_connect_point_trajectories import main_connect_point_trajectories


def connect_sequences(flow_dirs, traj_dirs, sample_ratio=2, flow_check_thres=1.0, traj_min_len=3,
                      skip_path_consistency=False, skip_exists=False, concurrency=2, rank=None, world=None, layout="csr"):
    """main_connect_point_trajectories for a list of sequences; returns the indices this rank processed.
    layout: pickle state of the track.npy files -- "csr" (default here: the batch driver is this package's own entry
    point and its files are read back through this package) or "reference" (files for an unmodified checkout)."""
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

    n_threads = max(1, min(int(concurrency), len(mine)))

    def worker():
        torch.cuda.set_device(device)
        stream = torch.cuda.Stream(device=device)
        try:
            if n_threads > 1:
                # several sequences in flight on this GPU: launches only -- the persistent forms (the frame loop of track mode,
                # the trust-region loop of a solve that rejects steps) need the device to themselves and would take the
                # device gate, i.e. serialise the sequences
                _hip.context().set_chain_mode(1)
            with torch.cuda.stream(stream):
                while True:
                    with lock:
                        if not todo or errors:
                            break
                        k = todo.pop(0)
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
