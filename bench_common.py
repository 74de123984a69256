"""Shared pieces of bench.py (the headline line) and bench_extras.py (every figure measured outside the timed region):
the workload constants, the peaks the fractions are priced against, and the CPU-oracle baseline of the headline workload."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

H, W, N_FRAMES, RATIO, THRES = 1080, 1920, 101, 2, 1.0
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# f64 VALU issue: 256 CUs x 4 SIMDs x 2.4 GHz, one wave64 f64 instruction per 4 cycles of its SIMD (MI355X_MICROARCH.md: SIMD-32,
# v_fma_f32 2 cycles; the 78.6 TFLOP/s vector f64 peak = half the f32 rate) -> 614.4 G wave-instructions/s
VALU_PEAK_GWIPS = 256 * 4 * 2.4 / 4.0


REFERENCE_SOLVER_THREADS = 8     # solver_options.num_threads at trajectory_optimize.cpp:79 (what BASELINE.md specifies)


def quiet_gc():
    """Python's cyclic collector out of a timed region that drives many frames from Python: a full collection of a process that has
    torch imported walks ~10^6 objects -- 30-35 ms on the GPU box's host, a whole 1080p sequence -- and lands wherever the allocation
    counts put it (round 5 found it inside the first timed `single_sequence` run: profiles/r05/r05_v_gc.txt).  Collect now and move
    everything alive out of the collector's reach; later collections walk only what the run allocates.  Skips no work of the measured
    path (INTEGRATION.md section 4 recommends the same to Python hosts)."""
    import gc
    gc.collect()
    gc.freeze()


def source_sha16():
    """Hash of the sources libpsfm_hip.so is built from (particle-sfm_amd/build.py::source_hash): what the PMC figures under
    profiles/ are stamped with.  Counters cannot be collected inside a timed run, so those figures are REPLAYED from the file --
    and the line says which sources they were measured on and whether they are this run's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("psfm_build", os.path.join(ROOT, "particle-sfm_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def replayed(path):
    """A PMC summary under profiles/ + where it came from: (dict or None, provenance string)."""
    if not os.path.exists(path):
        return None, None
    try:
        v = json.load(open(path))
    except Exception:
        return None, None
    mine = source_sha16()
    theirs = v.get("source_sha16")
    return v, {"file": "profiles/" + os.path.basename(path), "measured_on_sources": theirs, "this_run_sources": mine,
               "same_sources": bool(theirs is not None and theirs == mine), "round": v.get("round"),
               "how": "separate rocprofv3 --pmc passes (never inside the timed run), replayed"}


def reference_python_on_this_box(flows_f, flows_b, frames=4):
    """The reference's OWN Python (point_trajectory/utils.py flow_check + track.py track, unmodified, through oracle/ref_shim.py)
    timed on THIS box's host cores on the first `frames` frame pairs of the same tensors -- only where a reference tree is
    reachable (PSFM_REFERENCE_ROOT, default /root/reference: present in the build container, absent on the driver's GPU box;
    the sources are never copied into this repo).  Returns None when it is not."""
    from oracle import ref_shim
    if not ref_shim.available():
        return None
    import torch
    ref = ref_shim.load()
    ff = [f for f in flows_f[:frames].cpu().numpy()]
    fb = [f for f in flows_b[:frames].cpu().numpy()]
    t0 = time.perf_counter()
    _, occ = ref.flow_check(ff, fb, THRES)
    t1 = time.perf_counter()
    tr = ref.track(ff, occ, RATIO)
    t2 = time.perf_counter()
    pts = sum(t.length() for t in tr)
    return {"value": pts / (t2 - t0), "unit": "trajectory-points/s", "kind": "reference",
            "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
            "track_points_per_s": pts / (t2 - t1), "flow_check_s_per_pair": (t1 - t0) / frames,
            "sample": "first %d of %d frame pairs at 1080p, sample_ratio=2: the reference's flow_check + track (Python, torch-CPU "
                      "grid_sample with %d threads, SciPy EDT; per-track pybind-style objects): %d points in %.1f s"
                      % (frames, N_FRAMES - 1, torch.get_num_threads(), pts, t2 - t0)}


def cpu_threads_wide():
    """Threads of the 'wide' CPU figure: PSFM_CPU_THREADS, else min(32, host cores) -- the C restatement stops scaling there (the
    per-track list bookkeeping of extend_all is serial)."""
    return int(os.environ.get("PSFM_CPU_THREADS", "0")) or min(32, os.cpu_count() or 1)


def cpu_port_timed(fn, points_of):
    """fn() under the oracle at the reference's OWN thread count (8: solver_options.num_threads, trajectory_optimize.cpp:79 -- what
    BASELINE.md specifies) and at the wide count; every figure says how many threads it ran on (orc.num_threads() after setting it).
    Returns (result of the wide run, {"threads_8": {...}, "threads_wide": {...}})."""
    from oracle import oracle as orc
    out, res = {}, None
    for key, want in (("threads_8", min(REFERENCE_SOLVER_THREADS, os.cpu_count() or 1)), ("threads_wide", cpu_threads_wide())):
        orc.set_num_threads(want)
        t0 = time.perf_counter()
        res = fn()
        dt = time.perf_counter() - t0
        out[key] = {"points_per_s": points_of(res) / dt, "threads": orc.num_threads(), "seconds": dt}
    return res, out


def cpu_baseline(flows_f, flows_b, n_pairs):
    """The CPU oracle ("port": C restatement of the reference path, OpenMP over independent tracks / pixels) on the first
    n_pairs frame pairs of the same tensors, at the wide thread count (`value`, `cores`) and -- on a third of the sample -- at the
    reference's own 8 threads (`port_8_threads`); the reference runs its solver on 8 threads (trajectory_optimize.cpp:79) and
    everything else of the chain-only path on torch's intra-op pool."""
    import numpy as np
    from oracle import oracle as orc
    ff = [f for f in flows_f[:n_pairs].cpu().numpy()]
    fb = [f for f in flows_b[:n_pairs].cpu().numpy()]

    def run(n):
        _, occ = orc.flow_check(ff[:n], fb[:n], THRES)
        return orc.track(ff[:n], occ, RATIO)

    n8 = max(2, n_pairs // 3)
    orc.set_num_threads(min(REFERENCE_SOLVER_THREADS, os.cpu_count() or 1))
    t0 = time.perf_counter()
    R8 = run(n8)
    dt8 = time.perf_counter() - t0
    port8 = {"value": R8.n_points / dt8, "unit": "trajectory-points/s", "cores": orc.num_threads(),
             "sample": "first %d of %d frame pairs: %d points in %.1f s" % (n8, N_FRAMES - 1, R8.n_points, dt8),
             "why": "the reference's solver_options.num_threads = 8 (trajectory_optimize.cpp:79), BASELINE.md section 2"}
    orc.set_num_threads(cpu_threads_wide())
    t0 = time.perf_counter()
    R = run(n_pairs)
    dt = time.perf_counter() - t0
    out = {"value": R.n_points / dt, "unit": "trajectory-points/s", "cores": orc.num_threads(), "host_cores": os.cpu_count(),
           "kind": "port", "reference_solver_threads": REFERENCE_SOLVER_THREADS, "port_8_threads": port8,
           "sample": "first %d of %d frame pairs at 1080p, sample_ratio=2: flow_check + track + id order; %d points in %.1f s "
                     "(C restatement, OpenMP over tracks / pixels on %d threads; the per-track list bookkeeping of extend_all is serial)"
                     % (n_pairs, N_FRAMES - 1, R.n_points, dt, orc.num_threads())}
    try:
        here = reference_python_on_this_box(flows_f, flows_b)
    except Exception as e:     # noqa: BLE001
        here = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    out["reference_python_this_box"] = here if here is not None else \
        "absent (PSFM_REFERENCE_ROOT: no reference tree on this box -- it is not shipped to the GPU box and its sources are never copied here)"
    ref = os.path.join(ROOT, "BASELINE_MEASURED.json")
    if os.path.exists(ref):     # the reference's own Python, timed in the build container (scripts/measure_reference_baseline.py)
        try:
            m = json.load(open(ref))
            out["reference_python_build_container"] = {
                "track_points_per_s": m["track"]["points_per_s"], "track_optimize_points_per_s": m["track_optimize"]["points_per_s"],
                "flow_check_s_per_pair": m["flow_check_s_per_pair"], "host": m["host"], "frames": m["workload"]["frames"],
                "source": "BASELINE_MEASURED.json (replayed, not measured in this run)",
                "note": "unmodified reference Python through oracle/ref_shim.py, not this box: see BASELINE_MEASURED.json"}
        except Exception:
            pass
    return out, R
