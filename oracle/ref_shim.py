"""Run the reference's own point_trajectory Python, UNMODIFIED, from /root/reference.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (the build
container).  It is what pins oracle/psfm_oracle.c: tests/golden/make_golden.py
executes the reference through this shim and commits the outputs as fixtures;
nothing on the GPU box imports this file.

Two import shims are needed:
  * `cv2` (utils.py:22) is absent; it is only used by load_images/draw_traj.
  * `point_trajectory.optimize.build.particlesfm` (trajectory.py:23,
    main_connect_point_trajectories.py:25) is the pybind11/Ceres module that
    cannot be built here (no Ceres/Eigen/glog).  The stand-in below restates
    Trajectory / TrajectorySet (optimize/src/trajectory_base.cpp:21-185) in
    Python and forwards optimize_location to the C restatement in
    psfm_oracle.c -- so the solver iterate itself stays "parity unpinned".

The reference package is loaded under the alias `psfm_reference_pt` so it cannot
collide with the product package that mirrors its name.
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("PSFM_REFERENCE_ROOT", "/root/reference")
ALIAS = "psfm_reference_pt"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "point_trajectory"))


class Trajectory:
    """trajectory_base.cpp:21-93 (ctor (time, xy, *, buffer_size), extend, clear_buffer, ...)."""

    def __init__(self, time=None, point=None, *, buffer_size=0, times=None, xys=None, labels=None):
        self._buffer_size = int(buffer_size)
        self.times, self.labels, self.xys, self.buffer_xys = [], [], [], []
        if isinstance(time, dict):
            d = time
            self.times = [int(t) for t in d.get("frame_ids", [])]
            self.xys = [np.asarray(p, np.float64) for p in d.get("locations", [])]
            self.labels = [bool(b) for b in d.get("labels", [])]
        elif time is not None:
            self.extend(int(time), point)

    def extend(self, time, xy):
        xy = np.asarray(xy, dtype=np.float64).copy()
        self.times.append(int(time))
        self.labels.append(False)
        if self._buffer_size == 0:
            self.xys.append(xy)
            return
        self.buffer_xys.append(xy)
        if len(self.buffer_xys) > self._buffer_size:
            self.xys.append(self.buffer_xys.pop(0))

    def clear_buffer(self):
        self.xys.extend(self.buffer_xys)
        self.buffer_xys = []

    def set_buffer_xy(self, index, xy):
        if index >= len(self.buffer_xys):
            raise RuntimeError("Error! Index out of bound for the buffer.")
        self.buffer_xys[index] = np.asarray(xy, dtype=np.float64).copy()

    def length(self):
        return len(self.xys) + len(self.buffer_xys)

    def get_tail_location(self):
        if self.length() == 0:
            raise RuntimeError("Error! The trajectory is empty!")
        return self.buffer_xys[-1] if self.buffer_xys else self.xys[-1]

    def as_dict(self):
        return {"frame_ids": list(self.times), "locations": list(self.xys), "labels": list(self.labels)}


class TrajectorySet:
    """trajectory_base.cpp:95-107 (only what the hot path touches)."""

    def __init__(self, trajs=None):
        self.trajs = dict(trajs or {})

    def as_dict(self):
        return {k: v.as_dict() for k, v in sorted(self.trajs.items())}

    # ---- what motion_seg/load_cut_seq.py:46-79 calls; restated from trajectory_base.cpp:115-185 (the C++ cannot be
    # built here).  std::map iteration = ascending keys; std::random_shuffle is not restated: fixtures stay below the cap.
    def build_invert_indexes(self):
        self.invert_maps = {}
        for traj_id in sorted(self.trajs):
            t = self.trajs[traj_id]
            for index, frame_id in enumerate(t.times):
                self.invert_maps.setdefault(int(frame_id), {})[traj_id] = index

    def sample_inside_window(self, frame_ids, min_length=3, max_num_tracks=100000):
        if not getattr(self, "invert_maps", None):
            raise RuntimeError("Error! The inverted index maps have not been built!")
        counter = {}
        for frame_id in frame_ids:
            for traj_id in sorted(self.invert_maps.get(int(frame_id), {})):
                counter[traj_id] = counter.get(traj_id, 0) + 1
        traj_ids = [i for i in sorted(counter) if counter[i] >= min_length]
        if len(traj_ids) > max_num_tracks:
            raise NotImplementedError("std::random_shuffle (unseeded) is not restated")
        K, L = len(traj_ids), len(frame_ids)
        X, Y, M = np.zeros((K, L)), np.zeros((K, L)), np.zeros((K, L), np.int32)
        for i, traj_id in enumerate(traj_ids):
            for j, frame_id in enumerate(frame_ids):
                fm = self.invert_maps.get(int(frame_id))
                if fm is not None and traj_id in fm:
                    xy = (self.trajs[traj_id].xys + self.trajs[traj_id].buffer_xys)[fm[traj_id]]
                    X[i, j], Y[i, j], M[i, j] = xy[0], xy[1], 1
        return {"locations": (X, Y), "masks": M, "traj_ids": traj_ids}


def _optimize_location(uv12, uv_ref1, uv_ref2, ref2_scale, flow12_map, total_num, width, height):
    from . import oracle  # noqa: PLC0415
    return oracle.optimize_location(uv12, uv_ref1, uv_ref2, ref2_scale, flow12_map, total_num, width, height)


_loaded = {}


def load_reference(root=None, particlesfm_module=None, optimize_location=None):
    """load() with a different `particlesfm`: the REAL pybind module (tests/test_ref_ceres.py) or the stand-in with
    another optimize_location (the NumPy restatement, tests/test_oracle_golden.py)."""
    global REFERENCE_ROOT
    if root:
        REFERENCE_ROOT = root
    return load(particlesfm_module=particlesfm_module, optimize_location=optimize_location)


def load(particlesfm_module=None, optimize_location=None):
    """Returns a namespace with the reference's flow_check, track, track_optimize, grid_sample, ..."""
    global ALIAS
    key = (id(particlesfm_module), id(optimize_location))
    if key in _loaded:
        return _loaded[key]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    ALIAS = "psfm_reference_pt" if not _loaded else "psfm_reference_pt_%d" % len(_loaded)
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "tqdm" not in sys.modules:
        try:
            importlib.import_module("tqdm")
        except ImportError:
            m = types.ModuleType("tqdm")
            m.tqdm = lambda it, *a, **k: it
            sys.modules["tqdm"] = m
    pkg = types.ModuleType(ALIAS)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "point_trajectory")]
    sys.modules[ALIAS] = pkg
    opt = types.ModuleType(ALIAS + ".optimize")
    opt.__path__ = []
    bld = types.ModuleType(ALIAS + ".optimize.build")
    bld.__path__ = []
    standin = types.ModuleType(ALIAS + ".optimize.build.particlesfm")
    standin.Trajectory = Trajectory
    standin.TrajectorySet = TrajectorySet
    standin.optimize_location = optimize_location or _optimize_location
    if particlesfm_module is not None:
        standin = particlesfm_module
    bld.particlesfm = standin
    opt.build = bld
    pkg.optimize = opt
    sys.modules[ALIAS + ".optimize"] = opt
    sys.modules[ALIAS + ".optimize.build"] = bld
    sys.modules[ALIAS + ".optimize.build.particlesfm"] = standin
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        utils = importlib.import_module(ALIAS + ".utils")
        trajectory = importlib.import_module(ALIAS + ".trajectory")
        track = importlib.import_module(ALIAS + ".track")
        track_optimize = importlib.import_module(ALIAS + ".track_optimize")
    # silence the progress bars of the frame loops
    track.tqdm = lambda it, *a, **k: it
    track_optimize.tqdm = lambda it, *a, **k: it
    ns = types.SimpleNamespace(
        utils=utils, trajectory=trajectory,
        flow_check=utils.flow_check, read_flo=utils.read_flo, load_flows=utils.load_flows,
        grid_sample=trajectory.grid_sample, step_forward=trajectory.step_forward,
        IncrementalTrajectorySet=trajectory.IncrementalTrajectorySet,
        track=track.track, track_optimize=track_optimize.track_optimize,
        particlesfm=standin,
    )
    _loaded[key] = ns
    return ns


def load_consumers():
    """The reference's consumers of track.npy, imported UNMODIFIED: sfm/matches_from_flow.py (traj_to_matches) and
    motion_seg/load_cut_seq.py (+ core/dataset/data_utils.py resize / normalise).  cv2 and cvbase are absent: cv2 gets a
    stub whose image functions return blank arrays of the right shape (the image / depth tensors are not what the
    fixtures pin), cvbase a stub module."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    cv2 = sys.modules.get("cv2") or types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda name, flag=1: np.zeros((48, 64, 3), np.uint8) if flag != -1 else np.zeros((48, 64), np.float64)
    cv2.cvtColor = lambda img, code: img
    cv2.resize = lambda img, wh: np.zeros((wh[1], wh[0]) + tuple(img.shape[2:]), img.dtype)
    sys.modules["cv2"] = cv2
    for name in ("cvbase", "cvbase.optflow", "cvbase.optflow.visualize"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["cvbase.optflow.visualize"].flow2rgb = None
    import importlib.util

    def _load(alias, path, extra_path=None):
        if extra_path and extra_path not in sys.path:
            sys.path.append(extra_path)
        spec = importlib.util.spec_from_file_location(alias, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    mff = _load("psfm_reference_matches_from_flow", os.path.join(REFERENCE_ROOT, "sfm", "matches_from_flow.py"))
    mff.tqdm = lambda it, *a, **k: it
    lcs = _load("psfm_reference_load_cut_seq", os.path.join(REFERENCE_ROOT, "motion_seg", "load_cut_seq.py"),
                os.path.join(REFERENCE_ROOT, "motion_seg"))
    return types.SimpleNamespace(traj_to_matches=mff.traj_to_matches, load_cut_seq=lcs.load_cut_seq, cv2=cv2)


def trajs_to_csr(trajs):
    """List of stand-in Trajectory (full_trajs order) -> (birth, length, off, xy)."""
    birth = np.array([t.times[0] for t in trajs], np.int32)
    length = np.array([t.length() for t in trajs], np.int32)
    off = np.zeros(len(trajs) + 1, np.int64)
    off[1:] = np.cumsum(length)
    xy = np.concatenate([np.stack(t.xys + t.buffer_xys, 0) for t in trajs], 0) if trajs else np.zeros((0, 2))
    for t in trajs:
        assert t.times == list(range(t.times[0], t.times[0] + t.length()))
    return birth, length, off, xy.astype(np.float64)
