#!/usr/bin/env python3
"""The stage entry disk to disk (configs[1], tmpfs) five times: cold, over the existing track.npy, into a fresh directory, and both again
-- where the write phase's time goes.  Replacing 0.94 GB of page cache cost the rename 50-70 ms (write 0.206-0.223 s against 0.150-0.159 into
a fresh directory) until save_track_npy handed the old inode to a helper thread (PSFM_PLAIN_REPLACE=1 here: the plain rename)."""
import os
import shutil
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench_common import H, W, N_FRAMES, RATIO, THRES      # noqa: E402  (puts the package on sys.path)
import torch                                               # noqa: E402
import psfm_synth                                          # noqa: E402
from point_trajectory.utils import write_flo               # noqa: E402
from point_trajectory.main_connect_point_trajectories import main_connect_point_trajectories      # noqa: E402

if os.environ.get("PSFM_PLAIN_REPLACE"):
    import point_trajectory.trajectory as _t
    _t._replace_deferring_reclaim = os.replace
work = tempfile.mkdtemp(prefix="psfm_e2e_", dir="/dev/shm")
try:
    d = psfm_synth.synth_sequence_torch(N_FRAMES, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
    for name, key in (("flow_f", "flows_f"), ("flow_b", "flows_b")):
        os.makedirs(os.path.join(work, "flows", name))
        arr = d[key].cpu().numpy()
        for i in range(N_FRAMES - 1):
            write_flo(os.path.join(work, "flows", name, "%05d.flo" % i), arr[i])
    del d, arr
    torch.cuda.empty_cache()
    for label in ("cold", "over the existing file", "fresh directory", "over the existing file", "fresh directory"):
        if label == "fresh directory":
            shutil.rmtree(os.path.join(work, "traj"), ignore_errors=True)
        tm = {}
        t0 = time.perf_counter()
        main_connect_point_trajectories(os.path.join(work, "flows"), os.path.join(work, "traj"), sample_ratio=RATIO, flow_check_thres=THRES,
                                        skip_path_consistency=True, timings=tm)
        print("%-24s total %.3f s  ingest %.3f  compute %.4f  filter+D2H %.3f  write %.3f" % (
            label, time.perf_counter() - t0, tm["ingest_s"], tm["compute_s"], tm["filter_d2h_s"], tm["write_s"]), flush=True)
finally:
    from point_trajectory.trajectory import wait_for_reclaims
    wait_for_reclaims()
    shutil.rmtree(work, ignore_errors=True)
