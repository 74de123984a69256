"""Run on the GPU box: the stage entry point on a small synthetic sequence -> gpurun_out/track_fixture/{track.npy,
track_legacy.npy}.  The files are committed as tests/golden/track_gpu_60x80*.npy and consumed, in the build container,
by the REFERENCE's own unmodified sfm/matches_from_flow.py (tests/test_reference_consumers.py)."""
import os, sys, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import psfm_synth
from point_trajectory import main_connect_point_trajectories
from point_trajectory.utils import write_flo

T, H, W, r = 7, 60, 80, 2
d = psfm_synth.synth_sequence(T, H, W, seed=77, sigma=0.1, n_occluders=1, stride2=True)
tmp = tempfile.mkdtemp()
fd = os.path.join(tmp, "optical_flows")
for key, sub in (("flows_f", "flow_f"), ("flows_b", "flow_b"), ("flows_f2", "flow_f2"), ("flows_b2", "flow_b2")):
    os.makedirs(os.path.join(fd, sub))
    for i, f in enumerate(d[key]):
        write_flo(os.path.join(fd, sub, "%05d.flo" % i), f)
out = os.path.join(ROOT, "gpurun_out", "track_fixture")
os.makedirs(out, exist_ok=True)
main_connect_point_trajectories(fd, os.path.join(tmp, "traj"), sample_ratio=r, layout="csr")
shutil.copy(os.path.join(tmp, "traj", "track.npy"), os.path.join(out, "track.npy"))
main_connect_point_trajectories(fd, os.path.join(tmp, "traj_legacy"), sample_ratio=r)   # default: reference layout
shutil.copy(os.path.join(tmp, "traj_legacy", "track.npy"), os.path.join(out, "track_legacy.npy"))
print("ok", os.listdir(out))
