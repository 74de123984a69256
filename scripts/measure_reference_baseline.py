#!/usr/bin/env python3
"""Time the REFERENCE's own point_trajectory Python, unmodified, on the host cores of the build container
(BASELINE.md section 2 / SURVEY 8d "CPU reference timing") and write BASELINE_MEASURED.json.

    python scripts/measure_reference_baseline.py [--frames 11] [--out BASELINE_MEASURED.json]

What runs: /root/reference/point_trajectory/{utils,trajectory,track,track_optimize}.py through oracle/ref_shim.py
(torch-CPU grid_sample, SciPy EDT, NumPy -- the real ones); the pybind11/Ceres module cannot be built in this image, so
`particlesfm` is the shim's restated Trajectory + the C restatement of optimize_location (oracle/psfm_oracle.c, OpenMP,
8 threads like solver_options.num_threads = 8 at trajectory_optimize.cpp:79).  Workload: the headline shape
(1920x1080, sample_ratio 2, thres 1.0), synthetic flows of bench.py's generator, `--frames` frames; the per-frame cost
is ~constant once the grid is full, so points/s on 10+ frames is what a 100-frame run would give.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=11)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--ratio", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "BASELINE_MEASURED.json"))
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    import numpy as np
    import torch
    import psfm_synth
    from oracle import oracle as orc
    from oracle import ref_shim
    ref = ref_shim.load()
    T, H, W, r = args.frames, args.height, args.width, args.ratio
    d = psfm_synth.synth_sequence(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=True)
    out = {"what": "reference Python (unmodified, via oracle/ref_shim.py) on the build container's host cores",
           "host": {"nproc": os.cpu_count(), "torch_num_threads": torch.get_num_threads(),
                    "solver_threads": orc.num_threads(), "torch": torch.__version__, "numpy": np.__version__},
           "workload": {"height": H, "width": W, "frames": T, "sample_ratio": r, "flow_check_thres": 1.0,
                        "data": "psfm_synth.synth_sequence(seed=0, sigma=0.05, n_occluders=2)"}}
    t0 = time.perf_counter()
    _, occ = ref.flow_check(d["flows_f"], d["flows_b"], 1.0)
    t1 = time.perf_counter()
    _, occ2 = ref.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    t2 = time.perf_counter()
    trajs = ref.track(d["flows_f"], occ, r)
    t3 = time.perf_counter()
    pts_track = sum(t.length() for t in trajs)
    del trajs
    trajs = ref.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    t4 = time.perf_counter()
    pts_opt = sum(t.length() for t in trajs)
    out["flow_check_s_per_pair"] = (t1 - t0) / (T - 1)
    out["track"] = {"seconds": t3 - t2, "s_per_frame": (t3 - t2) / (T - 1), "points": pts_track,
                    "points_per_s": pts_track / (t3 - t2),
                    "with_flow_check_points_per_s": pts_track / ((t3 - t2) + (t1 - t0))}
    out["track_optimize"] = {"seconds": t4 - t3, "s_per_frame": (t4 - t3) / (T - 1), "points": pts_opt,
                             "points_per_s": pts_opt / (t4 - t3),
                             "with_both_flow_checks_points_per_s": pts_opt / ((t4 - t3) + (t2 - t0)),
                             "solver": "C restatement of the Ceres loop (oracle/psfm_oracle.c), %d threads" % orc.num_threads()}
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
