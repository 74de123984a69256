"""ctypes binding of libpsfm_hip.so (include/psfm.h).

The HIP library is the product; there is NO CPU fallback.  If the library is missing, or there is
no GPU, every compute entry point raises -- loudly -- instead of silently computing elsewhere.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("PSFM_HIP_LIB", os.path.join(_PKG, "lib", "libpsfm_hip.so"))

PSFM_OK, PSFM_ERR_ARG, PSFM_ERR_HIP, PSFM_ERR_CAPACITY, PSFM_ERR_SOLVER = 0, 1, 2, 3, 4
PROF_KINDS = {"flow_check": 0, "chain_step": 1, "respawn": 2, "solver": 3, "finalize": 4}

EXPORTS = [
    "psfm_last_error", "psfm_version", "psfm_device_count", "psfm_ctx_create", "psfm_ctx_destroy",
    "psfm_ctx_set_capacity", "psfm_flow_check", "psfm_grid_sample", "psfm_optimize_location", "psfm_track",
    "psfm_connect",
    "psfm_result_device", "psfm_result_copy", "psfm_result_solve_stats", "psfm_ctx_set_profiling",
    "psfm_profile_get", "psfm_ctx_set_chain_mode", "psfm_window_sample", "psfm_result_filter", "psfm_result_filtered_copy",
    "psfm_ctx_set_solver", "psfm_solver_counters", "psfm_traj_to_matches", "psfm_matches_copy",
    "psfm_shard_begin", "psfm_shard_step", "psfm_shard_solve_export", "psfm_shard_solve_control", "psfm_shard_solve_restore",
    "psfm_shard_solve_writeback", "psfm_shard_solve_record", "psfm_shard_finish", "psfm_result_keys",
    "psfm_shard_solve_control_async", "psfm_shard_window_state", "psfm_shard_peek_stall", "psfm_shard_frame",
    "psfm_shard_solve_control_chain_async", "psfm_shard_solve_poll", "psfm_shard_solve_local", "psfm_shard_solve_redo_local", "psfm_connect_batch", "psfm_solver_launches", "psfm_ctx_set_resident_budget", "psfm_resident_capacity", "psfm_load_flo_stack",
    "psfm_path_consistency_eval", "psfm_sort_records",
    "psfm_shard_peer_area", "psfm_shard_peer_epoch", "psfm_shard_peer_open", "psfm_shard_peer_connect", "psfm_shard_solve_blocks", "psfm_shard_solve_peer",
]


class SolveStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("successful_steps", ctypes.c_int32),
                ("termination", ctypes.c_int32), ("dogleg_nonGN", ctypes.c_int32),
                ("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TrackInfo(ctypes.Structure):
    _fields_ = [("n_traj", ctypes.c_int64), ("n_points", ctypes.c_int64), ("n_lanes_peak", ctypes.c_int64),
                ("lane_capacity", ctypes.c_int64), ("solver_iterations", ctypes.c_int64),
                ("n_solves", ctypes.c_int32), ("chain_mode", ctypes.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PsfmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libpsfm_hip status %d: %s" % (status, msg))
        self.status = status


_lib = None


def lib():
    """Load libpsfm_hip.so; raises RuntimeError (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libpsfm_hip.so not found at %s -- build it with `python particle-sfm_amd/build.py` "
            "(the HIP extension is mandatory, there is no CPU fallback)" % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one in the process.  Loaded before torch, libpsfm_hip.so would pull
    # the system libamdhip64 in, torch would add its own copy, and the two runtimes do not see each other's devices.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
    L.psfm_last_error.restype = ctypes.c_char_p
    L.psfm_last_error.argtypes = []
    L.psfm_version.restype = i32
    L.psfm_device_count.restype = i32
    L.psfm_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.psfm_ctx_destroy.argtypes = [vp]
    L.psfm_ctx_set_capacity.argtypes = [vp, f64, f64]
    L.psfm_load_flo_stack.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), i32, i32, i32, vp, i32, vp]
    L.psfm_flow_check.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp]
    L.psfm_grid_sample.argtypes = [vp, vp, i32, i32, i32, vp, i64, vp, vp]
    L.psfm_optimize_location.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, ctypes.POINTER(SolveStats), vp]
    L.psfm_path_consistency_eval.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]
    L.psfm_sort_records.argtypes = [vp, vp, vp, i64, i32, vp]
    L.psfm_track.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, ctypes.POINTER(TrackInfo), vp]
    L.psfm_connect.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, ctypes.POINTER(TrackInfo), vp]
    L.psfm_connect_batch.argtypes = [ctypes.POINTER(vp), i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                     ctypes.POINTER(i32), i32, i32, f32, i32, ctypes.POINTER(TrackInfo), vp]
    L.psfm_result_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.psfm_result_copy.argtypes = [vp, vp, vp, vp, vp, vp]
    L.psfm_result_solve_stats.argtypes = [vp, ctypes.POINTER(SolveStats), i32, ctypes.POINTER(ctypes.c_int32)]
    L.psfm_ctx_set_profiling.argtypes = [vp, i32]
    L.psfm_ctx_set_chain_mode.argtypes = [vp, i32]
    L.psfm_ctx_set_solver.argtypes = [vp, i32, i32]
    L.psfm_solver_counters.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32)]
    L.psfm_ctx_set_resident_budget.argtypes = [vp, i32]
    L.psfm_resident_capacity.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.psfm_solver_launches.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.psfm_result_filter.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), vp]
    L.psfm_result_filtered_copy.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.psfm_window_sample.argtypes = [vp, i32, i32, i32, i32, i64, ctypes.c_uint64, i32, i32, i32, i32, i64, vp, vp, vp, vp,
                                     ctypes.POINTER(i64), vp]
    L.psfm_traj_to_matches.argtypes = [vp, i32, i32, vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64), vp]
    L.psfm_matches_copy.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.psfm_shard_begin.argtypes = [vp, i32, i32, i32, i32, i64, i64, i32, vp, i64, vp]
    L.psfm_shard_step.argtypes = [vp, vp, vp, i32, vp]
    L.psfm_shard_solve_export.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.psfm_shard_frame.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    L.psfm_shard_solve_control.argtypes = [vp, i32, i32, i32, vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                           ctypes.POINTER(SolveStats), vp]
    L.psfm_shard_solve_control_async.argtypes = [vp, i32, i32, vp, vp]
    L.psfm_shard_solve_control_chain_async.argtypes = [vp, i32, i32, vp, vp]
    L.psfm_shard_solve_poll.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(SolveStats), vp]
    L.psfm_shard_solve_local.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.psfm_shard_peer_area.argtypes = [vp, ctypes.POINTER(vp), vp, vp]
    L.psfm_shard_peer_open.argtypes = [vp, vp, i32, ctypes.POINTER(vp)]
    L.psfm_shard_peer_epoch.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), i32, vp]
    L.psfm_shard_peer_connect.argtypes = [vp, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
    L.psfm_shard_solve_blocks.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.psfm_shard_solve_peer.argtypes = [vp, vp, vp, vp, vp, i32, ctypes.c_uint32, vp]
    L.psfm_shard_solve_redo_local.argtypes = [vp, vp, vp, vp, vp, i32, i32, ctypes.POINTER(SolveStats), vp]
    L.psfm_shard_window_state.argtypes = [vp, i32, i32, ctypes.POINTER(SolveStats), ctypes.POINTER(ctypes.c_int32), vp]
    L.psfm_shard_peek_stall.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.psfm_shard_solve_restore.argtypes = [vp, i32, vp]
    L.psfm_shard_solve_writeback.argtypes = [vp, i32, ctypes.POINTER(SolveStats), vp]
    L.psfm_shard_solve_record.argtypes = [vp, ctypes.POINTER(SolveStats)]
    L.psfm_shard_finish.argtypes = [vp, ctypes.POINTER(TrackInfo), vp]
    L.psfm_result_keys.argtypes = [vp, i32, i32, vp, vp]
    L.psfm_profile_get.argtypes = [vp, i32, ctypes.POINTER(f64), ctypes.POINTER(i64)]
    for name in EXPORTS:
        if name != "psfm_last_error":
            getattr(L, name).restype = i32
    _lib = L
    return L


def check(status):
    if status != PSFM_OK:
        raise PsfmError(status, lib().psfm_last_error().decode("utf-8", "replace"))


class Context:
    """One psfm_ctx (device workspace).  Not thread-safe; cached per device by `context()`."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        check(lib().psfm_ctx_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)
        self.resident_budget = 0           # (what set_resident_budget was last given)

    @property
    def handle(self):
        return self._h

    def set_capacity(self, lane_factor, traj_factor):
        check(lib().psfm_ctx_set_capacity(self._h, float(lane_factor), float(traj_factor)))

    def set_chain_mode(self, mode):
        """0 auto (persistent frame loop when the grid fits the device), 1 per-frame launches, 2 persistent loop only."""
        check(lib().psfm_ctx_set_chain_mode(self._h, int(mode)))

    def set_solver(self, mode, k=0):
        """track_optimize: 0 adaptive (fused solve, launch chain for windows whose solves reject steps), 1 launch chain,
        2 fused solve; k = trust-region iterations per fused launch (0: adaptive)."""
        check(lib().psfm_ctx_set_solver(self._h, int(mode), int(k)))

    def set_resident_budget(self, blocks):
        """> 0: this context's resident solves (flows whose solves reject steps) use at most `blocks` blocks and may run beside other
        contexts' -- the budgets of all contexts in flight on the device must add up to at most resident_capacity(); 0: only with the
        device to itself (default)."""
        check(lib().psfm_ctx_set_resident_budget(self._h, int(blocks)))
        self.resident_budget = int(blocks)

    def resident_capacity(self):
        n = ctypes.c_int32()
        check(lib().psfm_resident_capacity(self._h, ctypes.byref(n)))
        return int(n.value)

    def solver_counters(self):
        a, b, c_, k = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        check(lib().psfm_solver_counters(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c_), ctypes.byref(k)))
        out = {"fused": a.value, "fused_redone": b.value, "chain": c_.value, "k": k.value}
        r, g, it = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(lib().psfm_solver_launches(self._h, ctypes.byref(r), ctypes.byref(g), ctypes.byref(it)))
        out.update({"resident_launches": r.value, "resident_giveups": g.value, "iteration_launches": it.value})
        return out

    def set_profiling(self, enable):
        check(lib().psfm_ctx_set_profiling(self._h, int(enable)))   # 0 off, 1 every launch, N>1 every N-th chain_step

    def profile(self):
        out = {}
        for name, k in PROF_KINDS.items():
            ms, n = ctypes.c_double(), ctypes.c_int64()
            check(lib().psfm_profile_get(self._h, k, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = {"total_ms": ms.value, "launches": n.value}
        return out

    def close(self):
        if self._h:
            lib().psfm_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}
_contexts_lock = __import__("threading").Lock()   # batch.py worker threads create and release contexts concurrently


def context(device=None):
    """The psfm context of the calling THREAD on `device` (a context is not thread-safe; worker threads that process
    different sequences concurrently each get their own -- see point_trajectory.batch)."""
    import threading
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("point_trajectory (MI355X build) needs a HIP device: torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    if device is None:
        device = torch.cuda.current_device()
    device = int(device)
    key = (device, threading.get_ident())
    with _contexts_lock:
        ctx = _contexts.get(key)
    if ctx is None:
        ctx = Context(device)            # (outside the lock: creating a context allocates pinned memory)
        with _contexts_lock:
            _contexts[key] = ctx
    return ctx


def batch_contexts(n, device=None):
    """n psfm contexts of the calling thread on `device` for psfm_connect_batch (one per sequence of a batch; the first is the
    thread's ordinary context and owns the batch's shared workspace).  Cached: workspaces stay warm across batches."""
    import threading
    first = context(device)
    out = [first]
    for k in range(1, int(n)):
        key = (first.device, threading.get_ident(), "batch", k)
        with _contexts_lock:
            ctx = _contexts.get(key)
        if ctx is None:
            ctx = Context(first.device)
            with _contexts_lock:
                _contexts[key] = ctx
        out.append(ctx)
    return out


def release_thread_contexts():
    """Destroy the calling thread's contexts (worker threads call this before they exit)."""
    import threading
    tid = threading.get_ident()
    with _contexts_lock:
        mine = [_contexts.pop(k) for k in [k for k in _contexts if k[1] == tid]]
    for ctx in mine:
        ctx.close()


def current_stream_ptr(device=None):
    """torch's current stream ON `device` (default: the current device) -- pass the context's device when it may differ
    from the current one: a stream belongs to one device."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device (or host) address of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())
