"""ctypes front-end of the CPU oracle (oracle/psfm_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of psfm_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.

Function names follow the reference (paths relative to the reference root):
  flow_check      point_trajectory/utils.py:94-105
  grid_sample     point_trajectory/trajectory.py:25-37
  track           point_trajectory/track.py:24-50
  track_optimize  point_trajectory/track_optimize.py:24-53
  optimize_location  point_trajectory/optimize/src/trajectory_optimize.cpp:30-96
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libpsfm_oracle.so")
_lib = None


class SolveStats(ctypes.Structure):
    _fields_ = [
        ("iterations", ctypes.c_int32),
        ("successful_steps", ctypes.c_int32),
        ("termination", ctypes.c_int32),
        ("dogleg_nonGN", ctypes.c_int32),
        ("initial_cost", ctypes.c_double),
        ("final_cost", ctypes.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class _Result(ctypes.Structure):
    _fields_ = [
        ("n_traj", ctypes.c_int64),
        ("n_points", ctypes.c_int64),
        ("birth", ctypes.POINTER(ctypes.c_int32)),
        ("len", ctypes.POINTER(ctypes.c_int32)),
        ("off", ctypes.POINTER(ctypes.c_int64)),
        ("xy", ctypes.POINTER(ctypes.c_double)),
        ("n_solves", ctypes.c_int32),
        ("solves", ctypes.POINTER(SolveStats)),
    ]


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "psfm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c_p = ctypes.c_void_p
        L.orc_grid_sample.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p, ctypes.c_int64, c_p]
        L.orc_grid_sample.restype = None
        L.orc_flow_check.argtypes = [c_p, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_p, c_p]
        L.orc_flow_check.restype = None
        L.orc_optimize_location.argtypes = [c_p, c_p, c_p, c_p, c_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                            c_p, ctypes.POINTER(SolveStats)]
        L.orc_optimize_location.restype = ctypes.c_int
        L.orc_track.argtypes = [c_p, c_p, c_p, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_track.restype = ctypes.POINTER(_Result)
        L.orc_result_free.argtypes = [ctypes.POINTER(_Result)]
        L.orc_result_free.restype = None
        L.orc_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def set_variant(key, value):
    """Switch one of the readings of Ceres 2.0.0 the restatement rests on memory for (psfm_oracle.c: orc_set_variant; key = index
    into ceres_tr_numpy.VARIANT_KEYS or its name).  Returns the previous value.  Test infrastructure: the defaults are what ships."""
    from . import ceres_tr_numpy as ct
    k = ct.VARIANT_KEYS.index(key) if isinstance(key, str) else int(key)
    L = lib()
    L.orc_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
    L.orc_set_variant.restype = ctypes.c_int
    old = L.orc_set_variant(k, int(value))
    if old < 0:
        raise KeyError(key)
    return old


def set_num_threads(n):
    """Cap the host threads of the oracle's parallel loops (beyond ~16-32 the fork/join per loop costs more than it gains)."""
    L = lib()
    L.orc_set_num_threads.argtypes = [ctypes.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_set_num_threads(int(n))


def num_threads():
    """Host threads the oracle's parallel loops use (OMP_NUM_THREADS; results do not depend on it)."""
    return int(lib().orc_num_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def grid_sample(map_hwc, xy):
    """trajectory.py:25-37 on an HWC map.  map_hwc: (H,W,C) or (H,W); xy: (N,2) f64 -> (N,C) f32."""
    m = _f32(map_hwc)
    if m.ndim == 2:
        m = m[:, :, None]
    H, W, C = m.shape
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    out = np.empty((xy.shape[0], C), np.float32)
    lib().orc_grid_sample(_ptr(m), C, H, W, _ptr(xy), xy.shape[0], _ptr(out))
    return out


def flow_check(flows, flows_b, thres):
    """utils.py:94-105 -> (error_maps, occ_maps) lists of (H,W) f32 / bool arrays."""
    errs, occs = [], []
    for f, b in zip(flows, flows_b):
        f, b = _f32(f), _f32(b)
        H, W = f.shape[:2]
        occ = np.empty((H, W), np.uint8)
        err = np.empty((H, W), np.float32)
        lib().orc_flow_check(_ptr(f), _ptr(b), H, W, ctypes.c_float(thres), _ptr(occ), _ptr(err))
        errs.append(err)
        occs.append(occ.astype(bool))
    return errs, occs


def optimize_location(uv12, ref1, ref2, scale, flow12_map, total_num=None, width=None, height=None,
                      return_stats=False):
    """trajectory_optimize.cpp:30-96 (same argument order as the pybind entry)."""
    uv12 = np.ascontiguousarray(uv12, np.float64).reshape(-1, 4)
    n = uv12.shape[0] if total_num is None else int(total_num)
    ref1 = np.ascontiguousarray(ref1, np.float64).reshape(-1, 2)
    ref2 = np.ascontiguousarray(ref2, np.float64).reshape(-1, 2)
    scale = np.ascontiguousarray(scale, np.float64).reshape(-1)
    fm = _f32(flow12_map)
    H, W = fm.shape[:2]
    if width is not None:
        assert (int(width), int(height)) == (W, H)
    out = np.empty((n, 4), np.float64)
    st = SolveStats()
    rc = lib().orc_optimize_location(_ptr(uv12), _ptr(ref1), _ptr(ref2), _ptr(scale), _ptr(fm), n, W, H,
                                     _ptr(out), ctypes.byref(st))
    # rc != 0: Ceres' FAILURE -- ignored like the reference does (trajectory_optimize.cpp:81-82); `out` then equals uv12
    return (out, st.as_dict()) if return_stats else out


def path_consistency_eval(uv12, ref1, ref2, scale, flow12_map):
    """path_consistency_cost.h:42-59 as AutoDiffCostFunction<.., 6, 4> evaluates it: residuals (n,6), Jacobians (n,6,4)."""
    uv12 = np.ascontiguousarray(uv12, np.float64).reshape(-1, 4)
    n = uv12.shape[0]
    ref1 = np.ascontiguousarray(ref1, np.float64).reshape(n, 2)
    ref2 = np.ascontiguousarray(ref2, np.float64).reshape(n, 2)
    scale = np.ascontiguousarray(scale, np.float64).reshape(n)
    fm = _f32(flow12_map)
    H, W = fm.shape[:2]
    res, jac = np.empty((n, 6), np.float64), np.empty((n, 6, 4), np.float64)
    L = lib()
    L.orc_path_consistency_eval.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_path_consistency_eval.restype = None
    L.orc_path_consistency_eval(_ptr(uv12), _ptr(ref1), _ptr(ref2), _ptr(scale), _ptr(fm), n, W, H, _ptr(res), _ptr(jac))
    return res, jac


class TrackResult:
    """All trajectories in full_trajs order (index == saved id) as CSR arrays."""

    def __init__(self, birth, length, off, xy, solves=None):
        self.birth, self.length, self.off, self.xy, self.solves = birth, length, off, xy, solves or []

    @property
    def n_traj(self):
        return int(self.birth.shape[0])

    @property
    def n_points(self):
        return int(self.xy.shape[0])

    def traj(self, i):
        return self.birth[i], self.xy[self.off[i]:self.off[i + 1]]


def _run_track(flows, occ_maps, flows_f2, occ_maps_s2, sample_ratio):
    fl = [_f32(f) for f in flows]
    oc = [np.ascontiguousarray(o, dtype=np.uint8) for o in occ_maps]
    n = len(fl)
    H, W = fl[0].shape[:2]
    PA = ctypes.c_void_p * n
    fp = PA(*[f.ctypes.data for f in fl])
    op = PA(*[o.ctypes.data for o in oc])
    keep = [fl, oc]
    f2p = o2p = None
    if flows_f2 is not None:
        f2 = [_f32(f) for f in flows_f2]
        o2 = [np.ascontiguousarray(o, dtype=np.uint8) for o in occ_maps_s2]
        keep += [f2, o2]
        m = len(f2)
        PB = ctypes.c_void_p * max(m, 1)
        f2p = PB(*[f.ctypes.data for f in f2])
        o2p = PB(*[o.ctypes.data for o in o2])
    r = lib().orc_track(fp, op, f2p, o2p, n, H, W, int(sample_ratio))
    try:
        c = r.contents
        nt, npnt = c.n_traj, c.n_points
        birth = np.ctypeslib.as_array(c.birth, (max(nt, 1),))[:nt].copy()
        length = np.ctypeslib.as_array(c.len, (max(nt, 1),))[:nt].copy()
        off = np.ctypeslib.as_array(c.off, (nt + 1,)).copy()
        xy = np.ctypeslib.as_array(c.xy, (max(npnt, 1) * 2,))[:npnt * 2].copy().reshape(-1, 2)
        solves = [c.solves[i].as_dict() for i in range(c.n_solves)] if flows_f2 is not None else []
    finally:
        lib().orc_result_free(r)
    del keep
    return TrackResult(birth, length, off, xy, solves)


def track(flows, occ_maps, sample_ratio):
    """track.py:24-50"""
    return _run_track(flows, occ_maps, None, None, sample_ratio)


def track_optimize(flows, flows_f2, occ_maps, occ_maps_s2, sample_ratio):
    """track_optimize.py:24-53"""
    return _run_track(flows, occ_maps, flows_f2, occ_maps_s2, sample_ratio)


class ShardEngine:
    """One process's share of a track-sharded run (orc_shard_* in psfm_oracle.c): the engine that tests hand to
    psfm_dist.connect_sharded in place of the HIP engine -- same interface, the reference's semantics on the host."""

    device = "cpu"

    def __init__(self):
        self._s = None
        L = lib()
        c_p, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        self._REDUCE = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), i32, ctypes.POINTER(i32), c_p)
        L.orc_shard_begin.argtypes = [i32, i32, i32, i32, i64, i64, i32]
        L.orc_shard_begin.restype = c_p
        L.orc_shard_step.argtypes = [c_p, i32, c_p, c_p, c_p, ctypes.POINTER(i64)]
        L.orc_shard_step.restype = None
        L.orc_shard_set_blocked.argtypes = [c_p, c_p, i64]
        L.orc_shard_set_blocked.restype = None
        L.orc_shard_solve.argtypes = [c_p, i32, c_p, c_p, c_p, c_p, self._REDUCE, c_p]
        L.orc_shard_solve.restype = None
        L.orc_shard_finish.argtypes = [c_p]
        L.orc_shard_finish.restype = ctypes.POINTER(_Result)

    def begin(self, n_flows, H, W, ratio, g0, g1, optimize):
        import torch
        self.G = ((W + ratio - 1) // ratio) * ((H + ratio - 1) // ratio)
        self._s = lib().orc_shard_begin(int(n_flows), int(H), int(W), int(ratio), int(g0), int(g1), 1 if optimize else 0)
        self._x = torch.zeros(self.G + 1, dtype=torch.uint8)       # marks of this process + "a track survived" byte
        self._optimize = bool(optimize)

    def step(self, t, flow, occ):
        """births of frame t on the own band + chain step; returns the exchange tensor (uint8, G marks + 1 survivor byte)"""
        f = _f32(flow.numpy() if hasattr(flow, "numpy") else flow)
        o = np.ascontiguousarray(occ.numpy() if hasattr(occ, "numpy") else occ, dtype=np.uint8)
        n_alive = ctypes.c_int64(0)
        buf = self._x.numpy()
        lib().orc_shard_step(self._s, int(t), _ptr(f), _ptr(o), _ptr(buf), ctypes.byref(n_alive))
        buf[self.G] = 1 if n_alive.value > 0 else 0
        return self._x

    def after_exchange(self, t, x):
        buf = np.ascontiguousarray(x.numpy())
        lib().orc_shard_set_blocked(self._s, _ptr(buf), int(buf[self.G]))

    def solve(self, t, flow_prev, flow_cur, flow2_prev, occ2_prev, reduce):
        import torch
        arrs = [_f32(a.numpy() if hasattr(a, "numpy") else a) for a in (flow_prev, flow_cur, flow2_prev)]
        o2 = np.ascontiguousarray(occ2_prev.numpy() if hasattr(occ2_prev, "numpy") else occ2_prev, dtype=np.uint8)

        def cb(vals, n, is_max, user):
            v = torch.from_numpy(np.ctypeslib.as_array(vals, (n,)))       # a view: reduce() works in place
            reduce(v, [bool(is_max[i]) for i in range(n)])

        fn = self._REDUCE(cb)
        lib().orc_shard_solve(self._s, int(t), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(o2), fn, None)

    def finish(self):
        r = lib().orc_shard_finish(self._s)
        self._s = None
        try:
            c = r.contents
            nt, npnt = c.n_traj, c.n_points
            birth = np.ctypeslib.as_array(c.birth, (max(nt, 1),))[:nt].copy()
            length = np.ctypeslib.as_array(c.len, (max(nt, 1),))[:nt].copy()
            off = np.ctypeslib.as_array(c.off, (nt + 1,)).copy()
            xy = np.ctypeslib.as_array(c.xy, (max(npnt, 1) * 2,))[:npnt * 2].copy().reshape(-1, 2)
            solves = [c.solves[i].as_dict() for i in range(c.n_solves)] if self._optimize else []
        finally:
            lib().orc_result_free(r)
        return birth, length, off, xy, solves
