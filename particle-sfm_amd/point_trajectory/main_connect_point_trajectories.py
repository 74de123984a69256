"""Mirror of point_trajectory/main_connect_point_trajectories.py (reference :27-62): same signature, same
inputs (flow_dir/{flow_f,flow_b[,flow_f2,flow_b2]}/*.flo) and the same output (traj_dir/track.npy)."""
import argparse
import os

import numpy as np

from .utils import load_flows_device
from . import _hip
from .trajectory import run_connect, result_to_trajectory_set, save_track_npy


def main_connect_point_trajectories(flow_dir, traj_dir, sample_ratio=2, flow_check_thres=1.0, traj_min_len=3,
                                    skip_path_consistency=False, skip_exists=False, layout=None, timings=None):
    """Reference signature (:27) plus `layout`: the pickle state of track.npy -- "reference" (default; the file an
    unmodified particle-sfm checkout, pybind module included, reads) or "csr" (this package's compact arrays;
    PSFM_TRACK_LAYOUT=csr selects it for callers that cannot pass the argument) -- and `timings`: a dict that receives the
    seconds of the stage's phases (ingest / compute / filter_d2h / write)."""
    import time
    import torch
    t0 = time.perf_counter()
    if layout is None:
        layout = os.environ.get("PSFM_TRACK_LAYOUT", "reference")
    os.makedirs(traj_dir, exist_ok=True)
    output_npy_fname = os.path.join(traj_dir, "track.npy")
    if skip_exists and os.path.exists(output_npy_fname):
        return

    # load data (.flo -> HBM once; the error maps of the reference are never consumed, :39-40)
    flows_f = load_flows_device(os.path.join(flow_dir, "flow_f"))
    flows_b = load_flows_device(os.path.join(flow_dir, "flow_b"))
    flows_f2 = flows_b2 = None
    if not skip_path_consistency:
        flows_f2 = load_flows_device(os.path.join(flow_dir, "flow_f2"))
        flows_b2 = load_flows_device(os.path.join(flow_dir, "flow_b2"))

    if timings is not None:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    # fwd/bwd checks (utils.py:94-105) + connecting tracks into point trajectories (track.py / track_optimize.py):
    # one call, the occlusion maps stream into the frame loop from a side stream
    info = run_connect(flows_f, flows_b, flows_f2, flows_b2, flow_check_thres, sample_ratio, return_device=True)
    if timings is not None:
        torch.cuda.synchronize()
    t2 = time.perf_counter()

    # save the outputs (:56-62): ids are indices into the full list, short trajectories dropped -- filtered on the
    # device, staged through pinned memory, written as the same .npy/pickle container np.save produces
    trajectories = result_to_trajectory_set(_hip.context(), info, traj_min_len, reuse_pinned=True)
    t3 = time.perf_counter()
    save_track_npy(output_npy_fname, trajectories, layout=layout)
    if timings is not None:
        timings.update({"ingest_s": t1 - t0, "compute_s": t2 - t1, "filter_d2h_s": t3 - t2, "write_s": time.perf_counter() - t3,
                        "n_traj": int(info.n_traj), "n_points": int(info.n_points)})


def main(args):
    main_connect_point_trajectories(args.flow_dir, args.traj_dir, sample_ratio=args.sample_ratio,
                                    flow_check_thres=args.flow_check_thres, traj_min_len=args.traj_min_len,
                                    skip_path_consistency=args.skip_path_consistency)


if __name__ == "__main__":
    parser = argparse.ArgumentParser("Connecting and optimizing point trajectories from pairwise flows")
    parser.add_argument("--flow_dir", help="path to the folder of optical flows")
    parser.add_argument("--traj_dir", help="trajectory output")
    parser.add_argument("--sample_ratio", type=int, default=2, help="sample ratio of trajectories")
    parser.add_argument("--traj_min_len", type=int, default=3, help="minimum length of the trajectories")
    parser.add_argument("--flow_check_thres", type=float, default=1.0, help="flow consistency check threshold")
    parser.add_argument("--skip_path_consistency", action='store_true', help='whether to skip the path consistency optimization or not')
    main(parser.parse_args())
