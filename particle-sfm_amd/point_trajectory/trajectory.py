"""Mirror of the reference's point_trajectory/trajectory.py on top of libpsfm_hip.so.

What the reference does per frame with Python lists of pybind objects (IncrementalTrajectorySet,
trajectory.py:98-194) lives on the device here: lanes + a frame-major position log (csrc/psfm_track.hip).
This module keeps the reference's *callable surface*: `grid_sample` (:25-37) and the result container that
`track()` / `track_optimize()` return -- a list-like of Trajectory in full_trajs order.
"""
import ctypes
import os
import sys

import numpy as np

from . import _hip
from .optimize.build import particlesfm
from . import reference_pickle


def grid_sample(data, xy):
    """trajectory.py:25-37.  data: [C,H,W] torch tensor (C in {1,2}); xy: [N,2] array -> [N,C] float32 ndarray.
    Bit-exact with the reference's torch-CPU F.grid_sample(bilinear, zeros, align_corners=True)."""
    import torch
    ctx = _hip.context()
    dev = torch.device("cuda", ctx.device)
    C, H, W = int(data.shape[0]), int(data.shape[1]), int(data.shape[2])
    m = data.detach().to(dev, torch.float32).permute(1, 2, 0).contiguous()   # HWC, the kernels' native layout
    pts = torch.from_numpy(np.ascontiguousarray(np.asarray(xy, dtype=np.float64).reshape(-1, 2))).to(dev)
    out = torch.empty((pts.shape[0], C), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().psfm_grid_sample(ctx.handle, _hip.ptr(m), C, H, W, _hip.ptr(pts), pts.shape[0],
                                           _hip.ptr(out), _hip.current_stream_ptr(ctx.device)))
    return out.cpu().numpy()


class TrajectoryList:
    """What track()/track_optimize() return: trajectories in full_trajs order (index == saved id,
    main_connect_point_trajectories.py:56-60), stored as CSR arrays copied once from HBM.

    Behaves like the reference's list of Trajectory objects (len, indexing, iteration; elements expose
    .length(), .times, .xys, .labels, .as_dict()) without materialising ~5e7 Python objects up front."""

    def __init__(self, birth, length, off, xy, info=None, solve_stats=None):
        self.birth = birth          # (n,) int32
        self.length = length        # (n,) int32
        self.off = off              # (n+1,) int64
        self.xy = xy                # (n_points, 2) float64
        self.info = info or {}
        self.solve_stats = solve_stats or []

    def __len__(self):
        return int(self.birth.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return particlesfm.Trajectory._from_arrays(self.birth[i], self.xy[self.off[i]:self.off[i + 1]])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def n_points(self):
        return int(self.xy.shape[0])

    def to_trajectory_set(self, traj_min_len=3):
        """main_connect_point_trajectories.py:56-61: ids are list indices, short tracks dropped."""
        keep_mask = self.length >= int(traj_min_len)
        keep = np.nonzero(keep_mask)[0]
        length = self.length[keep]
        off = np.zeros(len(keep) + 1, np.int64)
        np.cumsum(length, out=off[1:])
        xy = self.xy[np.repeat(keep_mask, self.length)]     # array-speed: no per-trajectory Python objects
        return particlesfm.TrajectorySet._from_csr(keep, self.birth[keep], length, off, xy)


def _host_arrays(sizes_bytes, pinned_limit=2 << 30):
    """One host allocation cut into 64-byte aligned uint8 views.  Page-locked (torch's caching host allocator: no
    hipHostMalloc per call once warm, D2H at the full PCIe rate) up to `pinned_limit` bytes, pageable beyond."""
    offs = np.concatenate([[0], np.cumsum([(int(x) + 63) // 64 * 64 for x in sizes_bytes])])
    total = int(offs[-1])
    raw = None
    if 0 < total <= pinned_limit:
        try:
            import torch
            raw = torch.empty(total, dtype=torch.uint8, pin_memory=True).numpy()   # the views keep the tensor alive
        except Exception:
            raw = None
    if raw is None:
        raw = np.empty(total, np.uint8)
    return [raw[offs[i]:offs[i] + int(sizes_bytes[i])] for i in range(len(sizes_bytes))]


def _result_to_host(ctx, info):
    n, npnt = int(info.n_traj), int(info.n_points)
    b_birth, b_len, b_off, b_xy = _host_arrays([4 * n, 4 * n, 8 * (n + 1), 16 * npnt])
    birth = b_birth.view(np.int32)
    length = b_len.view(np.int32)
    off = b_off.view(np.int64)
    off[:] = 0
    xy = b_xy.view(np.float64).reshape(-1, 2)
    _hip.check(_hip.lib().psfm_result_copy(ctx.handle, birth.ctypes.data_as(ctypes.c_void_p),
                                           length.ctypes.data_as(ctypes.c_void_p), off.ctypes.data_as(ctypes.c_void_p),
                                           xy.ctypes.data_as(ctypes.c_void_p), _hip.current_stream_ptr(ctx.device)))
    stats = []
    if info.n_solves:
        arr = (_hip.SolveStats * int(info.n_solves))()
        nout = ctypes.c_int32()
        _hip.check(_hip.lib().psfm_result_solve_stats(ctx.handle, arr, int(info.n_solves), ctypes.byref(nout)))
        stats = [arr[i].as_dict() for i in range(nout.value)]
    return TrajectoryList(birth, length, off, xy, info.as_dict(), stats)


def result_to_trajectory_set(ctx, info, traj_min_len=3, reuse_pinned=False):
    """The saved set of main_connect_point_trajectories.py:56-61 from the device-resident result: the min-length filter
    runs on the GPU (psfm_result_filter) and only the kept trajectories cross PCIe.  reuse_pinned=True stages them in
    a pinned buffer owned by the context -- twice the copy rate, but the arrays of the returned TrajectorySet are views
    of that buffer and are overwritten by the next call (for callers that save the set straight away)."""
    import torch
    L = _hip.lib()
    k, npt = ctypes.c_int64(0), ctypes.c_int64(0)
    _hip.check(L.psfm_result_filter(ctx.handle, int(traj_min_len), ctypes.byref(k), ctypes.byref(npt), _hip.current_stream_ptr(ctx.device)))
    k, npt = int(k.value), int(npt.value)
    sizes = [4 * k, 4 * k, 4 * k, 8 * (k + 1), 16 * npt]
    offs = np.concatenate([[0], np.cumsum([(x + 63) // 64 * 64 for x in sizes])])
    if reuse_pinned:
        buf = getattr(ctx, "_pinned_result", None)
        if buf is None or buf.numel() < int(offs[-1]):
            buf = torch.empty(int(offs[-1]) + (64 << 20), dtype=torch.uint8, pin_memory=True)
            ctx._pinned_result = buf
        raw = buf.numpy()
    else:
        raw = np.empty(int(offs[-1]), np.uint8)
    ids = raw[offs[0]:offs[0] + sizes[0]].view(np.int32)
    birth = raw[offs[1]:offs[1] + sizes[1]].view(np.int32)
    length = raw[offs[2]:offs[2] + sizes[2]].view(np.int32)
    off = raw[offs[3]:offs[3] + sizes[3]].view(np.int64)
    xy = raw[offs[4]:offs[4] + sizes[4]].view(np.float64).reshape(-1, 2)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _hip.check(L.psfm_result_filtered_copy(ctx.handle, vp(ids), vp(birth), vp(length), vp(off), vp(xy), _hip.current_stream_ptr(ctx.device)))
    return particlesfm.TrajectorySet._from_csr(ids.astype(np.int64), birth, length, off, xy)


def save_track_npy(path, trajectories, layout="reference"):
    """np.save(path, trajectories) for consumers that np.load(path, allow_pickle=True).item(): the same .npy container
    (object array header + pickle), written with pickle protocol 5 so that the point array streams to the file without an
    intermediate bytes copy.
    layout "reference" (default): the pickle state of the reference's pybind class (bindings.cc:64-71) -- the file loads
    with the original module as well; "csr": this package's compact array state (array-speed save / load at 1e6+
    trajectories, readable only where this package provides `point_trajectory.optimize.build.particlesfm`)."""
    import pickle
    if layout not in ("reference", "csr"):
        raise ValueError("save_track_npy: layout must be 'reference' or 'csr'")
    if isinstance(trajectories, particlesfm.TrajectorySet):
        trajectories.pickle_layout = layout
    arr = np.empty((), dtype=object)
    arr[()] = trajectories
    # written beside the target and moved over it: a TrajectorySet loaded from the SAME path maps its points from the old file
    # (reference_pickle.load), and truncating that file in place would take the pages away under it (SIGBUS)
    target = path if str(path).endswith(".npy") else str(path) + ".npy"
    tmp = "%s.tmp-%d" % (target, os.getpid())
    try:
        with open(tmp, "w+b") as fp:
            np.lib.format.write_array_header_1_0(fp, np.lib.format.header_data_from_array_1_0(arr))
            if layout == "reference" and reference_pickle.can_stream(trajectories):
                reference_pickle.dump(fp, trajectories)      # the reference's object graph as opcodes, straight from the CSR
            else:
                pickle.dump(arr, fp, protocol=5)
        _replace_deferring_reclaim(tmp, target)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


_reclaims = []      # helper threads that drop replaced files (joined by wait_for_reclaims / at interpreter exit: they are not daemons)


def _replace_deferring_reclaim(tmp, target):
    """os.replace(tmp, target), without waiting for the kernel to give back the pages of the file it replaces: rename() frees the old
    inode's page cache before it returns (0.94 GB of track.npy on tmpfs: 50-70 ms of the stage's 0.29 s when it runs over an
    existing output, scripts/micro/e2e_overwrite.py).  A second link keeps the old inode alive across the rename; a helper thread
    drops it.  File systems without hard links take the plain rename."""
    import threading
    doomed = None
    if os.path.exists(target):
        doomed = "%s.old-%d-%d" % (target, os.getpid(), len(_reclaims))
        try:
            os.link(target, doomed)
        except OSError:
            doomed = None
    os.replace(tmp, target)
    if doomed is not None:
        def drop():
            try:
                os.unlink(doomed)
            except OSError:
                pass
        th = threading.Thread(target=drop, name="psfm-reclaim")
        th.start()
        _reclaims[:] = [t for t in _reclaims if t.is_alive()] + [th]


def wait_for_reclaims():
    """Blocks until the files replaced by save_track_npy are gone (tests; callers that are about to remove the directory)."""
    for t in list(_reclaims):
        t.join()
    del _reclaims[:]


def load_track_npy(path):
    """track.npy -> TrajectorySet, for this package's own consumers.  A file written by save_track_npy in the reference layout is
    read back through the footer the writer leaves behind the pickle (CSR arrays straight from the file: ~1 s instead of the
    ~25 s the reference layout takes to unpickle at 1.9 M trajectories); any other file through
    np.load(path, allow_pickle=True).item()."""
    return reference_pickle.load(path)


def _as_device_stack(maps, dtype, trailing):
    """list of (H,W[,2]) arrays | (n,H,W[,2]) array/tensor -> contiguous device tensor."""
    import torch
    ctx = _hip.context()
    dev = torch.device("cuda", ctx.device)
    if isinstance(maps, torch.Tensor):
        t = maps
    else:
        if isinstance(maps, (list, tuple)):
            if len(maps) and isinstance(maps[0], torch.Tensor):
                t = torch.stack(list(maps))
            else:
                t = torch.from_numpy(np.stack([np.asarray(m) for m in maps])) if len(maps) else torch.zeros((0,) + trailing)
        else:
            t = torch.from_numpy(np.asarray(maps))
    if t.dtype == torch.bool and dtype == torch.uint8:
        t = t.to(torch.uint8)
    return t.to(device=dev, dtype=dtype).contiguous()


def _capacity_of(ctx, key):
    """Table factors that worked for this shape on this context (default 2 x lanes, 8 x trajectory records)."""
    if not isinstance(getattr(ctx, "_capacity", None), dict):
        ctx._capacity = {}
    return ctx._capacity.get(key, (2.0, 8.0))


def run_connect(flows_f, flows_b, flows_f2, flows_b2, thres, sample_ratio, return_device=False):
    """flow_check + track / track_optimize in ONE psfm_connect call (the compute part of
    main_connect_point_trajectories.py:36-53): the occlusion maps are produced on a side stream while the frame
    loop consumes them.  Inputs: (n,H,W,2) float32 device tensors (stride-2 stacks None to skip path consistency)."""
    import torch
    ctx = _hip.context()
    n, H, W = int(flows_f.shape[0]), int(flows_f.shape[1]), int(flows_f.shape[2])
    if n < 1:
        raise ValueError("connect: need at least one flow field")
    n = min(n, int(flows_b.shape[0]))
    f2 = b2 = None
    if flows_f2 is not None:
        f2, b2 = flows_f2, flows_b2
        if n > 1 and (f2.shape[0] < n - 1 or b2.shape[0] < n - 1):
            raise ValueError("connect: need %d stride-2 flows" % (n - 1))
        if f2.numel() == 0:
            f2 = torch.zeros((1, H, W, 2), dtype=torch.float32, device=flows_f.device)
            b2 = f2
    info = _hip.TrackInfo()
    cap_key = ("connect", n, H, W, int(sample_ratio), f2 is not None)
    lane_f, traj_f = _capacity_of(ctx, cap_key)
    for attempt in range(6):
        ctx.set_capacity(lane_f, traj_f)
        ctx._capacity[cap_key] = (lane_f, traj_f)      # remembered per shape: a sequence that needed larger tables keeps them
        st = _hip.lib().psfm_connect(ctx.handle, _hip.ptr(flows_f), _hip.ptr(flows_b), _hip.ptr(f2), _hip.ptr(b2), n, H, W,
                                     float(thres), int(sample_ratio), None, None, ctypes.byref(info),
                                     _hip.current_stream_ptr(ctx.device))
        if st != _hip.PSFM_ERR_CAPACITY:
            break
        lane_f, traj_f = lane_f * 2.0, traj_f * 4.0
    _hip.check(st)
    if return_device:
        return info
    return _result_to_host(ctx, info)


def run_connect_batch(seqs, thres, sample_ratio):
    """psfm_connect_batch: flow_check + track / track_optimize for a BATCH of same-shape sequences (the directory of sequences the
    reference's driver walks, run_particlesfm.py:168-176) with every frame launch covering the whole batch.
    seqs: list of (flows_f, flows_b, flows_f2 | None, flows_b2 | None) device tensors (n_i,H,W,2) -- all with stride-2 stacks or
    none.  Returns (contexts, infos): context i holds sequence i's result (`_result_to_host(ctx, info)`,
    `result_to_trajectory_set(ctx, info)`) until the calling thread's next batch."""
    import torch
    n_seq = len(seqs)
    if n_seq < 1:
        return [], []
    H, W = int(seqs[0][0].shape[1]), int(seqs[0][0].shape[2])
    opt = seqs[0][2] is not None
    keep = []           # (tensors created here must outlive the call)
    nf = (ctypes.c_int * n_seq)()
    vp = ctypes.c_void_p
    ff, fb, f2, b2 = (vp * n_seq)(), (vp * n_seq)(), (vp * n_seq)(), (vp * n_seq)()
    for i, (a, b, a2, b2_) in enumerate(seqs):
        if int(a.shape[1]) != H or int(a.shape[2]) != W or (a2 is not None) != opt:
            raise ValueError("connect_batch: sequence %d differs in frame size / stride-2 stacks from sequence 0" % i)
        n = min(int(a.shape[0]), int(b.shape[0]))
        if n < 1:
            raise ValueError("connect_batch: sequence %d has no flow field" % i)
        if opt and n > 1 and (a2.shape[0] < n - 1 or b2_.shape[0] < n - 1):
            raise ValueError("connect_batch: sequence %d needs %d stride-2 flows" % (i, n - 1))
        nf[i] = n
        ff[i], fb[i] = a.data_ptr(), b.data_ptr()
        if opt:
            if a2.numel() == 0:
                a2 = b2_ = torch.zeros((1, H, W, 2), dtype=torch.float32, device=a.device)
                keep.append(a2)
            f2[i], b2[i] = a2.data_ptr(), b2_.data_ptr()
    ctxs = _hip.batch_contexts(n_seq)
    handles = (vp * n_seq)(*[c.handle for c in ctxs])
    infos = (_hip.TrackInfo * n_seq)()
    cap_key = ("batch", H, W, int(sample_ratio), opt)
    lane_f, traj_f = _capacity_of(ctxs[0], cap_key)
    for attempt in range(6):
        for c in ctxs:
            c.set_capacity(lane_f, traj_f)
        ctxs[0]._capacity[cap_key] = (lane_f, traj_f)
        st = _hip.lib().psfm_connect_batch(handles, n_seq, ff, fb, f2 if opt else None, b2 if opt else None, nf, H, W, float(thres),
                                           int(sample_ratio), infos, _hip.current_stream_ptr(ctxs[0].device))
        if st != _hip.PSFM_ERR_CAPACITY:
            break
        lane_f, traj_f = lane_f * 2.0, traj_f * 4.0      # tables too small for one of the sequences: grow all and rerun
    _hip.check(st)
    return ctxs, [infos[i] for i in range(n_seq)]


def run_track(flows, occ_maps, flows_f2, occ_maps_s2, sample_ratio, return_device=False):
    """Shared driver of track() / track_optimize(): one psfm_track call (whole frame loop on the device)."""
    import torch
    ctx = _hip.context()
    fl = _as_device_stack(flows, torch.float32, (1, 1, 2))
    oc = _as_device_stack(occ_maps, torch.uint8, (1, 1))
    n, H, W = int(fl.shape[0]), int(fl.shape[1]), int(fl.shape[2])
    if n < 1:
        raise ValueError("track: need at least one flow field")
    if oc.shape[0] < n:
        raise ValueError("track: %d occlusion maps for %d flows" % (oc.shape[0], n))
    f2 = o2 = None
    if flows_f2 is not None:
        f2 = _as_device_stack(flows_f2, torch.float32, (1, 1, 2))
        o2 = _as_device_stack(occ_maps_s2, torch.uint8, (1, 1))
        if n > 1 and (f2.shape[0] < n - 1 or o2.shape[0] < n - 1):
            raise ValueError("track_optimize: need %d stride-2 flows / occlusion maps" % (n - 1))
        if f2.numel() == 0:   # single-pair sequence: the stride-2 stack is empty but must be non-NULL
            f2 = torch.zeros((1, H, W, 2), dtype=torch.float32, device=fl.device)
            o2 = torch.zeros((1, H, W), dtype=torch.uint8, device=fl.device)
    info = _hip.TrackInfo()
    cap_key = ("track", n, H, W, int(sample_ratio), f2 is not None)
    lane_f, traj_f = _capacity_of(ctx, cap_key)
    for attempt in range(6):
        ctx.set_capacity(lane_f, traj_f)
        ctx._capacity[cap_key] = (lane_f, traj_f)
        st = _hip.lib().psfm_track(ctx.handle, _hip.ptr(fl), _hip.ptr(oc), _hip.ptr(f2), _hip.ptr(o2), n, H, W,
                                   int(sample_ratio), ctypes.byref(info), _hip.current_stream_ptr(ctx.device))
        if st != _hip.PSFM_ERR_CAPACITY:
            break
        if os.environ.get("PSFM_VERBOSE"):
            print("psfm_track: %s -> retry with larger tables" % _hip.lib().psfm_last_error().decode(), file=sys.stderr)
        lane_f, traj_f = lane_f * 2.0, traj_f * 4.0   # tables too small for this sequence: grow and rerun
    _hip.check(st)
    if return_device:
        return info
    return _result_to_host(ctx, info)
