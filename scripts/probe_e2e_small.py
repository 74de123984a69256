"""Where a SMALL sequence's disk-to-disk time goes (configs[0] shape: 2 x 49 .flo files of 3.3 MB): the stage's own phase timings
(main_connect_point_trajectories(timings=)), then the ingest alone with different reader / staging counts."""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch, psfm_synth
from point_trajectory.utils import write_flo, load_flows_device
from point_trajectory.main_connect_point_trajectories import main_connect_point_trajectories
H, W, T, r = 480, 854, 50, 4
base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
work = tempfile.mkdtemp(prefix="psfm_e2es_", dir=base)
try:
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=3, sigma=0.05, n_occluders=2, stride2=False)
    for name, key in (("flow_f", "flows_f"), ("flow_b", "flows_b")):
        os.makedirs(os.path.join(work, "flows", name))
        arr = d[key].cpu().numpy()
        for i in range(T - 1):
            write_flo(os.path.join(work, "flows", name, "%05d.flo" % i), arr[i])
    for rep in range(4):
        tm = {}
        t0 = time.perf_counter()
        main_connect_point_trajectories(os.path.join(work, "flows"), os.path.join(work, "traj"), sample_ratio=r, skip_path_consistency=True, timings=tm)
        tot = time.perf_counter() - t0
        print("pass %d: total %.1f ms | ingest %.1f | compute %.2f | filter + D2H %.1f | write %.1f  (%d trajectories, %d points, track.npy %.1f MB)" % (
            rep, 1e3 * tot, 1e3 * tm["ingest_s"], 1e3 * tm["compute_s"], 1e3 * tm["filter_d2h_s"], 1e3 * tm["write_s"], tm["n_traj"], tm["n_points"],
            os.path.getsize(os.path.join(work, "traj", "track.npy")) / 1e6), flush=True)
    for readers, staging in ((1, 2), (4, 8), (8, 16), (16, 32)):
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            a = load_flows_device(os.path.join(work, "flows", "flow_f"), n_readers=readers, n_staging=staging)
            b = load_flows_device(os.path.join(work, "flows", "flow_b"), n_readers=readers, n_staging=staging)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("ingest of 2 x %d files (%.0f MB) with %2d readers / %2d staging buffers: %.1f ms (%.1f GB/s)" % (T - 1, 2 * (T - 1) * H * W * 8 / 1e6, readers, staging,
              1e3 * min(ts[1:]), 2 * (T - 1) * H * W * 8 / 1e9 / min(ts[1:])), flush=True)
finally:
    shutil.rmtree(work, ignore_errors=True)
