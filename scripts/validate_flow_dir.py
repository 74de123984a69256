#!/usr/bin/env python3
"""One command for someone who HAS real flows (DAVIS / Sintel / ScanNet .flo directories from the reference's RAFT stage,
scripts/download_examples.sh + run_particlesfm.py): parity and timings of this package on them.

    python scripts/validate_flow_dir.py <flow_dir> [--optimize] [--sample-ratio 2] [--thres 1.0] [--frames N] [--no-gpu] [--json out.json]

<flow_dir> is what main_connect_point_trajectories reads (main_connect_point_trajectories.py:27-35): flow_f/ flow_b/ and, with
--optimize, flow_f2/ flow_b2/ holding Middlebury .flo files.  What runs on the same files:

  hip        the product path (psfm_connect through point_trajectory.trajectory.run_connect)            -- needs a GPU, skipped with --no-gpu
  oracle     oracle/psfm_oracle.c (the CPU restatement the parity tests use)                             -- always
  reference  the UNMODIFIED reference Python from $PSFM_REFERENCE_ROOT (default /root/reference) through oracle/ref_shim.py,
             when that tree exists (with the real pybind module when PSFM_REF_PARTICLESFM_SO / oracle/_ref/ holds one, else with the
             shim's stand-in whose optimize_location is the C restatement)                             -- bounded by --ref-frames

and prints one JSON record: per pair of engines whether ids / lengths are equal, max |dxy| in px, per-solve iteration / termination
agreement; seconds and trajectory points per second per engine, and `cpu_baseline` (kind "reference" when the reference ran, else
"port") in bench.py's format for THIS host.  Exit code 1 when any compared pair differs in ids / lengths or by more than 1e-4 px."""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
TOL = 1e-4


def compare(a, b):
    """a, b: dicts with birth, length, xy (+ solves: list of stat dicts or None)."""
    import numpy as np
    same = bool(len(a["birth"]) == len(b["birth"]) and np.array_equal(a["birth"], b["birth"]) and np.array_equal(a["length"], b["length"]))
    out = {"ids_lengths_equal": same, "trajectories": [int(len(a["birth"])), int(len(b["birth"]))],
           "max_abs_dxy_px": float(np.abs(a["xy"] - b["xy"]).max()) if same and len(a["xy"]) else (0.0 if same else None)}
    if a.get("solves") is not None and b.get("solves") is not None:
        ia, ib = [s["iterations"] for s in a["solves"]], [s["iterations"] for s in b["solves"]]
        ta, tb = [s["termination"] for s in a["solves"]], [s["termination"] for s in b["solves"]]
        out["solves"] = len(ia)
        out["solve_iterations_equal"] = ia == ib
        out["solve_terminations_equal"] = ta == tb
        if ia != ib and len(ia) == len(ib):
            out["solves_with_other_iteration_counts"] = [k for k, (x, y) in enumerate(zip(ia, ib)) if x != y][:20]
    out["ok"] = bool(same and out["max_abs_dxy_px"] is not None and out["max_abs_dxy_px"] <= TOL)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("flow_dir")
    ap.add_argument("--optimize", action="store_true", help="track_optimize (path consistency) instead of track")
    ap.add_argument("--sample-ratio", type=int, default=2)
    ap.add_argument("--thres", type=float, default=1.0)
    ap.add_argument("--frames", type=int, default=0, help="use only the first N flow pairs (0 = all)")
    ap.add_argument("--ref-frames", type=int, default=12, help="flow pairs the reference Python runs on (it takes ~1 s per 1080p frame)")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args(argv)
    import numpy as np
    from oracle import oracle as orc
    from oracle import ref_shim
    from point_trajectory.utils import read_flo

    def stack(sub, n=None):
        names = sorted(glob.glob(os.path.join(args.flow_dir, sub, "*.flo")))
        if n is not None:
            names = names[:n]
        return [np.ascontiguousarray(read_flo(f), np.float32) for f in names]

    # (small maps: the oracle's OpenMP loops crawl when spread over every core of a 256-core host)
    orc.set_num_threads(min(int(os.environ.get("PSFM_CPU_THREADS", "0")) or 32, os.cpu_count() or 1))
    n = args.frames or None
    ff, fb = stack("flow_f", n), stack("flow_b", n)
    if not ff or len(ff) != len(fb):
        sys.exit("validate_flow_dir: %s/flow_f and flow_b must hold the same (non-zero) number of .flo files" % args.flow_dir)
    f2 = b2 = None
    if args.optimize:
        f2, b2 = stack("flow_f2", len(ff) - 1), stack("flow_b2", len(ff) - 1)
        if len(f2) != len(ff) - 1 or len(b2) != len(ff) - 1:
            sys.exit("validate_flow_dir: --optimize needs %d files in flow_f2 and flow_b2" % (len(ff) - 1))
    H, W = ff[0].shape[:2]
    r = args.sample_ratio
    rec = {"flow_dir": os.path.abspath(args.flow_dir), "frames": len(ff) + 1, "height": H, "width": W, "sample_ratio": r, "thres": args.thres,
           "mode": "track_optimize" if args.optimize else "track", "host_cores": os.cpu_count(), "engines": {}, "parity": {}}
    res = {}

    # ---- the CPU oracle ----
    t0 = time.perf_counter()
    _, occ = orc.flow_check(ff, fb, args.thres)
    occ2 = orc.flow_check(f2, b2, args.thres)[1] if args.optimize and f2 else None
    O = orc.track_optimize(ff, f2, occ, occ2, r) if args.optimize else orc.track(ff, occ, r)
    dt = time.perf_counter() - t0
    res["oracle"] = {"birth": O.birth, "length": O.length, "xy": O.xy, "solves": O.solves if args.optimize else None}
    rec["engines"]["oracle"] = {"seconds": dt, "points": int(O.n_points), "points_per_s": O.n_points / dt, "threads": orc.num_threads()}

    # ---- the product path ----
    if not args.no_gpu:
        import torch
        if not torch.cuda.is_available():
            sys.exit("validate_flow_dir: no GPU visible (pass --no-gpu to compare the CPU engines only)")
        from point_trajectory.trajectory import run_connect
        dev = torch.device("cuda", torch.cuda.current_device())
        up = lambda s: torch.from_numpy(np.stack(s)).to(dev) if s else None
        t_ff, t_fb, t_f2, t_b2 = up(ff), up(fb), up(f2), up(b2)
        run_connect(t_ff, t_fb, t_f2, t_b2, args.thres, r)                      # warm-up (workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = run_connect(t_ff, t_fb, t_f2, t_b2, args.thres, r)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["hip"] = {"birth": R.birth, "length": R.length, "xy": R.xy, "solves": R.solve_stats if args.optimize else None}
        rec["engines"]["hip"] = {"seconds": dt, "points": int(R.n_points), "points_per_s": R.n_points / dt,
                                 "device": torch.cuda.get_device_name(dev), "includes": "flow_check + recurrence + finalize + D2H of the result"}
        rec["parity"]["hip_vs_oracle"] = compare(res["hip"], res["oracle"])

    # ---- the reference's own Python, where its tree is reachable ----
    if ref_shim.available():
        k = min(len(ff), max(2, args.ref_frames))
        real = None
        so = os.environ.get("PSFM_REF_PARTICLESFM_SO") or (sorted(glob.glob(os.path.join(ROOT, "oracle", "_ref", "particlesfm*.so"))) or [None])[0]
        if so and os.path.isfile(so):
            import importlib.machinery
            import importlib.util
            loader = importlib.machinery.ExtensionFileLoader("particlesfm", so)
            real = importlib.util.module_from_spec(importlib.util.spec_from_file_location("particlesfm", so, loader=loader))
            loader.exec_module(real)
        ref = ref_shim.load(particlesfm_module=real)
        t0 = time.perf_counter()
        _, rocc = ref.flow_check(ff[:k], fb[:k], args.thres)
        if args.optimize:
            _, rocc2 = ref.flow_check(f2[:k - 1], b2[:k - 1], args.thres)
            trajs = ref.track_optimize(ff[:k], f2[:k - 1], rocc, rocc2, r)
        else:
            trajs = ref.track(ff[:k], rocc, r)
        dt = time.perf_counter() - t0
        if real is None:
            birth, length, off, xy = ref_shim.trajs_to_csr(trajs)
        else:      # the real pybind class: the same accessors as the product's mirror (bindings.cc:33-57)
            birth = np.array([t.as_dict()["frame_ids"][0] for t in trajs], np.int32)
            length = np.array([t.length() for t in trajs], np.int32)
            xy = np.concatenate([np.asarray(t.as_dict()["locations"], np.float64).reshape(-1, 2) for t in trajs]) if trajs else np.zeros((0, 2))
        pts = int(length.sum())
        res["reference"] = {"birth": birth, "length": length, "xy": xy, "solves": None}
        rec["engines"]["reference"] = {"seconds": dt, "points": pts, "points_per_s": pts / dt, "frames": k + 1,
                                       "solver": "real Ceres (%s)" % so if real is not None else "oracle/psfm_oracle.c behind the shim's stand-in module",
                                       "root": ref_shim.REFERENCE_ROOT}
        # the same k pairs through the oracle (and, by parity above, the product)
        _, oocc = orc.flow_check(ff[:k], fb[:k], args.thres)
        Ok = (orc.track_optimize(ff[:k], f2[:k - 1], oocc, orc.flow_check(f2[:k - 1], b2[:k - 1], args.thres)[1], r) if args.optimize
              else orc.track(ff[:k], oocc, r))
        rec["parity"]["reference_vs_oracle_first_%d_pairs" % k] = compare(res["reference"], {"birth": Ok.birth, "length": Ok.length, "xy": Ok.xy, "solves": None})
        rec["parity"]["occlusion_maps_equal_reference_oracle"] = bool(all(np.array_equal(np.asarray(a, bool), np.asarray(b, bool)) for a, b in zip(rocc, oocc)))
        rec["cpu_baseline"] = {"value": pts / dt, "unit": "trajectory-points/s", "cores": os.cpu_count(), "kind": "reference",
                               "sample": "first %d flow pairs of %s" % (k, os.path.basename(os.path.abspath(args.flow_dir)))}
    else:
        rec["engines"]["reference"] = "absent: no reference tree at %s (set PSFM_REFERENCE_ROOT)" % ref_shim.REFERENCE_ROOT
        e = rec["engines"]["oracle"]
        rec["cpu_baseline"] = {"value": e["points_per_s"], "unit": "trajectory-points/s", "cores": e["threads"], "kind": "port",
                               "sample": "all %d flow pairs" % len(ff)}
    if "hip" in rec["engines"]:
        rec["gpu_over_cpu_baseline"] = rec["engines"]["hip"]["points_per_s"] / rec["cpu_baseline"]["value"]
    rec["ok"] = all(v["ok"] for v in rec["parity"].values() if isinstance(v, dict)) and rec["parity"].get("occlusion_maps_equal_reference_oracle", True)
    text = json.dumps(rec, indent=1)
    print(text)
    if args.json:
        with open(args.json, "w") as f:
            f.write(text)
    return 0 if rec["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
