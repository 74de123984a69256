"""`point_trajectory.optimize.build.particlesfm` -- same dotted path, same names as the reference's
pybind11 module (optimize/src/bindings.cc:27-77), backed by libpsfm_hip.so instead of Ceres.

  optimize_location   bindings.cc:31  -> psfm_optimize_location (HIP; trajectory_optimize.cpp:30-96)
  Trajectory          bindings.cc:33-57, optimize/src/trajectory_base.{h,cpp}
  TrajectorySet       bindings.cc:59-75

Because the classes live at the reference's module path, `np.load("track.npy", allow_pickle=True).item()`
in the unmodified consumers (motion_seg/load_cut_seq.py:46, sfm/matches_from_flow.py:56,
motion_seg/eval_traj_iou.py:28) resolves to them.  The DEFAULT pickle state is the reference's
`{id: {"frame_ids", "locations", "labels"}}` (bindings.cc:39-46,64-71), so files written by either implementation load
in the other -- also with the original pybind module on the path.  The compact CSR state (array-speed save / load,
readable by this class only) is opt-in: `TrajectorySet.pickle_layout = "csr"` on the object, which
`save_track_npy(..., layout="csr")` sets.
"""
import numpy as np


def optimize_location(uv12, uv_ref1, uv_ref2, ref2_scale, flow12_map, total_num, width, height):
    """bindings.cc:31.  All inputs are copied (like the pybind by-value signature); returns (N,4) f64."""
    import ctypes
    import torch
    from ... import _hip
    ctx = _hip.context()
    n = int(total_num)
    dev = torch.device("cuda", ctx.device)

    def dv(a, shape, dt):
        a = np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(shape))
        return torch.from_numpy(a).to(dev)

    fm = np.asarray(flow12_map)
    if fm.shape[0] != int(height) or fm.shape[1] != int(width):
        raise RuntimeError("optimize_location: flow12_map shape %s does not match (height=%d,width=%d)"
                           % (fm.shape, height, width))
    uv = dv(np.asarray(uv12)[:n], (n, 4), np.float64)
    r1 = dv(np.asarray(uv_ref1)[:n], (n, 2), np.float64)
    r2 = dv(np.asarray(uv_ref2)[:n], (n, 2), np.float64)
    sc = dv(np.asarray(ref2_scale)[:n], (n,), np.float64)
    # the reference force-casts the map to f64 (trajectory_optimize.h:40); an f32 map widens exactly
    fmt = dv(fm, (int(height), int(width), 2), np.float32)
    if fm.dtype == np.float64 and not np.array_equal(fmt.cpu().numpy().astype(np.float64), fm):
        raise RuntimeError("optimize_location: flow12_map carries f64 values that are not exact in f32")
    out = torch.empty((n, 4), dtype=torch.float64, device=dev)
    st = _hip.SolveStats()
    _hip.check(_hip.lib().psfm_optimize_location(ctx.handle, _hip.ptr(uv), _hip.ptr(r1), _hip.ptr(r2), _hip.ptr(sc),
                                                 _hip.ptr(fmt), n, int(width), int(height), _hip.ptr(out),
                                                 ctypes.byref(st), _hip.current_stream_ptr(ctx.device)))
    optimize_location.last_stats = st.as_dict()
    return out.cpu().numpy()


def path_consistency_eval(uv12, uv_ref1, uv_ref2, ref2_scale, flow12_map):
    """path_consistency_cost.h:42-59: residuals (N,6) and Jacobians (N,6,4) of the N residual blocks trajectory_optimize.cpp:56-65
    adds, at uv12 -- psfm_path_consistency_eval (the solver kernels' own arithmetic, without a solve around it).  Not part of the
    reference's module; there for verification."""
    import torch
    from ... import _hip
    ctx = _hip.context()
    dev = torch.device("cuda", ctx.device)
    uv = torch.from_numpy(np.ascontiguousarray(np.asarray(uv12, np.float64).reshape(-1, 4))).to(dev)
    n = uv.shape[0]
    r1 = torch.from_numpy(np.ascontiguousarray(np.asarray(uv_ref1, np.float64).reshape(n, 2))).to(dev)
    r2 = torch.from_numpy(np.ascontiguousarray(np.asarray(uv_ref2, np.float64).reshape(n, 2))).to(dev)
    sc = torch.from_numpy(np.ascontiguousarray(np.asarray(ref2_scale, np.float64).reshape(n))).to(dev)
    fm = torch.from_numpy(np.ascontiguousarray(np.asarray(flow12_map, np.float32))).to(dev)
    H, W = int(fm.shape[0]), int(fm.shape[1])
    res = torch.empty((n, 6), dtype=torch.float64, device=dev)
    jac = torch.empty((n, 6, 4), dtype=torch.float64, device=dev)
    _hip.check(_hip.lib().psfm_path_consistency_eval(ctx.handle, _hip.ptr(uv), _hip.ptr(r1), _hip.ptr(r2), _hip.ptr(sc), _hip.ptr(fm),
                                                     n, W, H, _hip.ptr(res), _hip.ptr(jac), _hip.current_stream_ptr(ctx.device)))
    return res.cpu().numpy(), jac.cpu().numpy()


class Trajectory:
    """optimize/src/trajectory_base.h:35-60.  Constructors (bindings.cc:34-37):
    Trajectory(time, point, *, buffer_size=0) | Trajectory(times, xys, *, labels=[]) | Trajectory(dict)."""

    __slots__ = ("_times", "_xys", "_labels", "_buffer", "_buffer_size")

    def __init__(self, time=None, point=None, *, buffer_size=0, labels=None, times=None, xys=None):
        self._buffer_size = int(buffer_size)
        self._times, self._labels, self._buffer = [], [], []
        self._xys = []
        if times is not None or xys is not None:
            time, point = times, xys
        if isinstance(time, dict):
            self._from_dict(time)
        elif time is None:
            pass
        elif np.ndim(time) == 0:
            self.extend(int(time), point)
        else:
            self._times = [int(t) for t in time]
            self._xys = [np.asarray(p, dtype=np.float64).reshape(2) for p in point]
            self._labels = [bool(b) for b in labels] if labels is not None and len(labels) else [False] * len(self._times)

    # -- internal fast constructor used by the HIP result views (no per-point Python work) --
    @classmethod
    def _from_arrays(cls, birth, xy):
        t = cls.__new__(cls)
        t._buffer_size = 0
        t._times = range(int(birth), int(birth) + xy.shape[0])
        t._xys = xy
        t._labels = None
        t._buffer = []
        return t

    def _from_dict(self, d):
        self._times = [int(t) for t in d["frame_ids"]] if "frame_ids" in d else []
        loc = d.get("locations", [])
        self._xys = np.asarray(loc, dtype=np.float64).reshape(-1, 2) if len(loc) else []
        self._labels = [bool(b) for b in d["labels"]] if "labels" in d else []

    # -- read-only properties: each access copies to Python lists, like pybind's def_readonly --
    @property
    def times(self):
        return list(self._times)

    @property
    def labels(self):
        return [False] * len(self._times) if self._labels is None else list(self._labels)

    @property
    def xys(self):
        return [np.array(p, dtype=np.float64) for p in self._xys]

    @property
    def buffer_xys(self):
        return [np.array(p, dtype=np.float64) for p in self._buffer]

    def _materialise(self):
        if not isinstance(self._times, list):
            self._times = list(self._times)
        if not isinstance(self._xys, list):
            self._xys = [np.array(p, dtype=np.float64) for p in self._xys]
        if self._labels is None:
            self._labels = [False] * len(self._times)

    def extend(self, time, xy):   # trajectory_base.cpp:55-67
        self._materialise()
        xy = np.asarray(xy, dtype=np.float64).reshape(2).copy()
        self._times.append(int(time))
        self._labels.append(False)
        if self._buffer_size == 0:
            self._xys.append(xy)
            return
        self._buffer.append(xy)
        if len(self._buffer) > self._buffer_size:
            self._xys.append(self._buffer.pop(0))

    def clear_buffer(self):       # trajectory_base.cpp:69-74
        self._materialise()
        self._xys.extend(self._buffer)
        self._buffer = []

    def set_buffer_xy(self, index, xy):   # trajectory_base.cpp:76-80
        if index >= len(self._buffer):
            raise RuntimeError("Error! Index out of bound for the buffer.")
        self._buffer[index] = np.asarray(xy, dtype=np.float64).reshape(2).copy()

    def set_label(self, index, label):
        self._materialise()
        self._labels[index] = bool(label)

    def set_labels(self, input_labels):
        self._labels = [bool(b) for b in input_labels]

    def length(self):             # trajectory_base.cpp:82-84
        return len(self._xys) + len(self._buffer)

    def get_tail_location(self):  # trajectory_base.cpp:86-93
        if self.length() == 0:
            raise RuntimeError("Error! The trajectory is empty!")
        return np.array(self._buffer[-1] if self._buffer else self._xys[-1], dtype=np.float64)

    def as_dict(self):            # trajectory_base.cpp:47-53 (xys only: buffered points are not exported)
        return {"frame_ids": self.times, "locations": self.xys, "labels": self.labels}

    def _state(self):
        """Compact pickle state: same keys; `locations` as one (N,2) array (every consumer does
        np.array(locations); pybind's vector<V2D> caster accepts it row by row)."""
        return {"frame_ids": list(self._times), "locations": np.asarray(self._xys, dtype=np.float64).reshape(-1, 2),
                "labels": self.labels}

    def __getstate__(self):
        return self._state()

    def __setstate__(self, d):
        self._buffer_size, self._buffer = 0, []
        self._from_dict(d)



class TrajectorySet:
    """optimize/src/trajectory_base.h:62-77, .cpp:95-185.

    Two interchangeable backings: the reference's `{id: Trajectory}` map (`trajs`), and -- what the HIP path
    produces -- a CSR over trajectories (ids, birth, length, off, xy[, labels]) from which the map is only
    materialised on demand.  At cfg-2 scale the reference layout means ~5e7 per-point Python objects; the CSR
    form keeps `np.save`, `build_invert_indexes` and `sample_inside_window` array-speed.

    Pickle state: by default the reference's `{id: {"frame_ids","locations","labels"}}` (bindings.cc:64-71), which the
    original pybind module loads as well; with `pickle_layout = "csr"` on the object (or PSFM_TRACK_LAYOUT=csr in the
    environment for objects that do not say) the compact CSR arrays, loadable by this class only.  `__setstate__`
    accepts both."""

    pickle_layout = None    # None: PSFM_TRACK_LAYOUT or "reference"; "reference" | "csr"

    def __init__(self, trajs=None):
        self._csr = None
        self._map = {}
        self._invert = None
        if trajs:
            for k in sorted(trajs):
                v = trajs[k]
                self._map[int(k)] = v if isinstance(v, Trajectory) else Trajectory(v)

    @classmethod
    def _from_csr(cls, ids, birth, length, off, xy, labels=None):
        ts = cls.__new__(cls)
        ts._csr = (np.asarray(ids, np.int64), np.asarray(birth, np.int32), np.asarray(length, np.int32),
                   np.asarray(off, np.int64), np.asarray(xy, np.float64).reshape(-1, 2),
                   None if labels is None else np.asarray(labels, bool))
        ts._map = None
        ts._invert = None
        return ts

    # -- the reference's `trajs` member (def_readonly): materialised lazily from the CSR --
    @property
    def trajs(self):
        if self._map is None:
            ids, birth, length, off, xy, labels = self._csr
            m = {}
            for j in range(len(ids)):
                t = Trajectory._from_arrays(birth[j], xy[off[j]:off[j + 1]])
                if labels is not None:
                    t._labels = labels[off[j]:off[j + 1]].tolist()
                m[int(ids[j])] = t
            self._map = m
        return self._map

    @trajs.setter
    def trajs(self, value):
        self._map = dict(value)
        self._csr = None
        self._invert = None

    def __len__(self):
        return len(self._csr[0]) if self._map is None else len(self._map)

    def as_dict(self):            # trajectory_base.cpp:95-101
        return {k: v.as_dict() for k, v in self.trajs.items()}

    def insert(self, traj_id, traj):   # trajectory_base.cpp:109-113
        m = self.trajs
        if traj_id in m:
            raise RuntimeError("Error! The trajectory id already exists!")
        m[int(traj_id)] = traj
        self._map = dict(sorted(m.items()))
        self._csr = None
        self._invert = None

    def _to_csr(self):
        """(ids, birth/frames, length, off, xy, labels) from whichever backing is current; `frames` per point."""
        if self._csr is not None and self._map is None:
            ids, birth, length, off, xy, labels = self._csr
            n = int(off[-1])
            frames = np.arange(n, dtype=np.int64) - np.repeat(off[:-1] - birth.astype(np.int64), length)
            return ids, off, frames, xy, labels
        ids, cnt = [], []
        for k, t in self._map.items():
            ids.append(k)
            cnt.append(len(t._xys))
        ids = np.asarray(ids, dtype=np.int64)
        cnt = np.asarray(cnt, dtype=np.int64)
        off = np.zeros(len(ids) + 1, np.int64)
        np.cumsum(cnt, out=off[1:])
        frames = np.empty(int(off[-1]), np.int64)
        xy = np.empty((int(off[-1]), 2), np.float64)
        labels = np.zeros(int(off[-1]), bool)
        for i, t in enumerate(self._map.values()):
            c = int(cnt[i])
            frames[off[i]:off[i + 1]] = np.fromiter(t._times, np.int64, count=len(t._times))[:c]
            xy[off[i]:off[i + 1]] = np.asarray(t._xys, dtype=np.float64).reshape(-1, 2)
            if t._labels is not None:
                labels[off[i]:off[i + 1]] = np.asarray(t._labels, bool)[:c]
        return ids, off, frames, xy, labels

    def build_invert_indexes(self):    # trajectory_base.cpp:115-125, as flat arrays instead of a map of maps
        ids, off, frames, xy, _ = self._to_csr()
        self._invert = (ids, off, frames, xy)

    def sample_inside_window(self, frame_ids, min_length=3, max_num_tracks=100000):   # trajectory_base.cpp:127-185
        if self._invert is None:
            raise RuntimeError("Error! The inverted index maps have not been built!")
        ids, off, frames, xy = self._invert
        frame_ids = [int(f) for f in frame_ids]
        L = len(frame_ids)
        owner = np.repeat(np.arange(len(ids)), np.diff(off))
        # count, per trajectory, the observations that fall on a window frame (duplicates in
        # frame_ids count twice, as in the reference's loop over frame_ids)
        uniq, mult = np.unique(np.asarray(frame_ids, np.int64), return_counts=True)
        if len(uniq) and len(frames):
            pos = np.minimum(np.searchsorted(uniq, frames), len(uniq) - 1)
            hit = uniq[pos] == frames
            w = np.where(hit, mult[pos], 0)
        else:
            hit = np.zeros(len(frames), bool)
            w = np.zeros(len(frames), np.int64)
        counter = np.bincount(owner, weights=w, minlength=len(ids)).astype(np.int64)
        present = np.bincount(owner, weights=hit, minlength=len(ids)) > 0
        sel = np.nonzero(present & (counter >= int(min_length)))[0]
        if len(sel) > int(max_num_tracks):
            # the reference uses an unseeded std::random_shuffle here (trajectory_base.cpp:150-153)
            sel = np.random.permutation(sel)[:int(max_num_tracks)]
        K = len(sel)
        X = np.zeros((K, L), np.float64)
        Y = np.zeros((K, L), np.float64)
        M = np.zeros((K, L), np.int32)
        row_of = -np.ones(len(ids), np.int64)
        row_of[sel] = np.arange(K)
        rows = row_of[owner] if len(owner) else np.zeros(0, np.int64)
        keep = hit & (rows >= 0)
        kr, kf = rows[keep], frames[keep]
        kx, ky = xy[keep, 0], xy[keep, 1]
        for j, f in enumerate(frame_ids):
            m = kf == f
            X[kr[m], j] = kx[m]
            Y[kr[m], j] = ky[m]
            M[kr[m], j] = 1
        return {"locations": (X, Y), "masks": M, "traj_ids": [int(i) for i in ids[sel]]}

    # ---- pickle (bindings.cc:64-71) ----
    def _legacy_state(self):
        """The reference's state (bindings.cc:64-71 over Trajectory::as_dict, trajectory_base.cpp:47-53):
        {id: {"frame_ids": [int], "locations": (n,2) f64 rows, "labels": [bool]}} -- straight from the CSR when that is
        the backing (no Trajectory objects; `locations` is a view of the point array, which pybind's vector<V2D> caster
        and np.array() both take row by row)."""
        if self._csr is not None and self._map is None:
            ids, birth, length, off, xy, labels = self._csr
            ids_l, b_l, o_l = ids.tolist(), birth.tolist(), off.tolist()
            state = {}
            for j in range(len(ids_l)):
                s, e = o_l[j], o_l[j + 1]
                state[ids_l[j]] = {"frame_ids": list(range(b_l[j], b_l[j] + e - s)), "locations": xy[s:e],
                                   "labels": [False] * (e - s) if labels is None else labels[s:e].tolist()}
            return state
        return {k: v._state() for k, v in self.trajs.items()}

    def __getstate__(self):
        import os
        if self.__dict__.get("_state_override") is not None:     # (the streaming writer's template instance, see reference_pickle.py)
            return self._state_override
        layout = self.pickle_layout or os.environ.get("PSFM_TRACK_LAYOUT", "reference")
        if layout == "csr" and self._csr is not None and self._map is None:
            ids, birth, length, off, xy, labels = self._csr
            return {"__psfm_csr__": 1, "ids": ids, "birth": birth, "length": length, "off": off, "xy": xy,
                    "labels": labels}
        return self._legacy_state()

    def __setstate__(self, state):
        self._invert = None
        if isinstance(state, dict) and state.get("__psfm_csr__") == 1:
            self._csr = (state["ids"], state["birth"], state["length"], state["off"], state["xy"], state["labels"])
            self._map = None
            return
        csr = self._csr_from_reference_state(state)
        if csr is not None:        # (array-speed consumers; `trajs` is materialised from it on demand)
            self._csr, self._map = csr, None
            return
        self._csr = None
        self._map = {int(k): (v if isinstance(v, Trajectory) else Trajectory(v)) for k, v in sorted(state.items())}

    @staticmethod
    def _csr_from_reference_state(state):
        """The reference's `{id: {"frame_ids", "locations", "labels"}}` state as CSR arrays, when it is what the builder writes:
        consecutive frame ids, no label set, (n, 2) locations.  None otherwise (the generic map takes over).  At 1e6+
        trajectories this is what makes loading a reference-layout file array-speed on this side (no Trajectory objects)."""
        if not isinstance(state, dict) or len(state) < 1024:
            return None
        try:
            keys = sorted(state)
            n = len(keys)
            ids = np.fromiter(keys, np.int64, n)
            birth, length, locs = np.empty(n, np.int32), np.empty(n, np.int32), []
            for j, k in enumerate(keys):
                v = state[k]
                f, loc = v["frame_ids"], v["locations"]
                m = len(f)
                if m == 0 or any(v["labels"]) or len(v["labels"]) != m or f != list(range(f[0], f[0] + m)):
                    return None
                if not isinstance(loc, np.ndarray):
                    loc = np.asarray(loc, np.float64)
                if loc.shape != (m, 2) or loc.dtype != np.float64:
                    return None
                birth[j], length[j] = f[0], m
                locs.append(loc)
            off = np.zeros(n + 1, np.int64)
            np.cumsum(length, out=off[1:])
            return (ids, birth, length, off, np.concatenate(locs), None)
        except (KeyError, TypeError, ValueError, IndexError):
            return None
