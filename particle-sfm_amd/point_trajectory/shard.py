"""This rank's share of a track-sharded run on its GPU: the engine psfm_dist.connect_sharded drives (one process per GPU,
RCCL over xGMI).  Everything computes in libpsfm_hip.so (psfm_shard_* in include/psfm.h); this class only owns the two
exchange tensors -- the stamped grid-resolution `blocked` maps (+ survivor byte) and the solver sums -- because they must
be torch tensors for torch.distributed, and sequences the export -> reduce -> control steps of a solve.
"""
import ctypes
import os

import numpy as np

from . import _hip

SUM_GMAX, N_SUM, K_MAX = 5, 13, 8          # csrc/psfm_solver.hip: slot combined by max, sums per iteration, fused iterations


class HipShardEngine:
    def __init__(self, ctx=None, k=4):
        self.ctx = ctx or _hip.context()
        self.k = int(k)                    # iterations speculated per fused solve; follows what the sequence needs
        self._need = []                    # ... = the most a clean solve of the last 16 frames needed
        self._pending = []                 # solves enqueued since the last checkpoint: (frame, the tensors a redo needs)
        self.check_every = 16              # frames between two checkpoints (earlier when stalled() says a solve has to be redone)
        self.counters = {"fused": 0, "fused_redone": 0}
        # iterations speculated beyond what the clean solves of the last 16 frames needed: an iteration more costs a few us per frame,
        # a solve that needed one more than speculated costs a redo through the launch chain and the frames behind it once again
        self.k_margin = int(os.environ.get("PSFM_SHARD_K_MARGIN", "0"))

    @property
    def device(self):
        import torch
        return torch.device("cuda", self.ctx.device)

    def _sp(self):
        return _hip.current_stream_ptr(self.ctx.device)

    def begin(self, n_flows, H, W, ratio, g0, g1, optimize):
        import torch
        # a run of this engine that was aborted between solve() and checkpoint() (an exception in a collective, a solve that did
        # not terminate) must not hand its enqueued solves -- frame numbers and flow tensors of ANOTHER sequence -- to this one
        self._pending = []
        self._need = []
        self.counters = {"fused": 0, "fused_redone": 0}
        self.G = ((W + ratio - 1) // ratio) * ((H + ratio - 1) // ratio)
        self.pitch = (self.G + 1 + 255) // 256 * 256
        self.maps = torch.zeros(2 * self.pitch, dtype=torch.uint8, device=self.device)
        self.sums = torch.zeros(K_MAX * N_SUM, dtype=torch.float64, device=self.device)
        self.n_flows, self.optimize = int(n_flows), bool(optimize)
        _hip.check(_hip.lib().psfm_shard_begin(self.ctx.handle, int(n_flows), int(H), int(W), int(ratio), int(g0), int(g1),
                                               1 if optimize else 0, _hip.ptr(self.maps), self.pitch, self._sp()))

    def step(self, t, flow, occ):
        """births of frame t on the own band + chain step; returns the tensor the ranks all-reduce (max)"""
        assert flow.is_cuda and flow.is_contiguous() and occ.is_contiguous()
        _hip.check(_hip.lib().psfm_shard_step(self.ctx.handle, _hip.ptr(flow), _hip.ptr(occ), int(t), self._sp()))
        o = (int(t) & 1) * self.pitch
        return self.maps[o:o + self.G + 1]

    def after_exchange(self, t, x):
        pass          # the reduced map is the buffer the next chain step reads

    def _control(self, t, kind, k):
        done, redo, st = ctypes.c_int32(0), ctypes.c_int32(0), _hip.SolveStats()
        _hip.check(_hip.lib().psfm_shard_solve_control(self.ctx.handle, int(t), kind, k, _hip.ptr(self.sums), ctypes.byref(done),
                                                       ctypes.byref(redo), ctypes.byref(st), self._sp()))
        return bool(done.value), bool(redo.value), st

    def solve(self, t, flow_prev, flow_cur, flow2_prev, occ2_prev, reduce):
        """track_optimize.py:49-50 for the own tracks: fused export -> sums over the ranks -> control, all ENQUEUED -- nothing is
        read back here.  A solve that does not go as speculated raises the device-side stall flag (every later launch of this
        context is a no-op from then on); checkpoint() finds out, redoes it with the launch chain (one export / reduce / control
        per trust-region iteration) and tells the driver where to resume."""
        L, h = _hip.lib(), self.ctx.handle
        p = (_hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev), _hip.ptr(occ2_prev))
        k = max(1, min(K_MAX, self.k))
        mask = [(i % N_SUM) == SUM_GMAX for i in range(k * N_SUM)]
        _hip.check(L.psfm_shard_solve_export(h, *p, int(t), 0, k, _hip.ptr(self.sums), self._sp()))
        reduce(self.sums[:k * N_SUM], mask)
        _hip.check(L.psfm_shard_solve_control_async(h, int(t), k, _hip.ptr(self.sums), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev)))    # (keeps the frames alive for a redo)

    def frame(self, t, flow_prev, flow_cur, occ, flow2_prev, occ2_prev, reduce_first, reduce):
        """step(t) and the fused export of solve(t) as ONE launch (psfm_shard_frame): the solve of frame t needs this rank's own
        tracks only, so it does not wait for the exchange of the frame's marks.  reduce_first(x): the exchange of the marks (the
        driver's all-reduce(max)), issued right behind the launch; then the sums over the ranks and the control step, enqueued
        like solve()'s."""
        L, h = _hip.lib(), self.ctx.handle
        assert flow_cur.is_cuda and flow_cur.is_contiguous() and occ.is_contiguous()
        k = max(1, min(K_MAX, self.k))
        mask = [(i % N_SUM) == SUM_GMAX for i in range(k * N_SUM)]
        _hip.check(L.psfm_shard_frame(h, _hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev), _hip.ptr(occ),
                                      _hip.ptr(occ2_prev), int(t), k, _hip.ptr(self.sums), self._sp()))
        o = (int(t) & 1) * self.pitch
        reduce_first(self.maps[o:o + self.G + 1])
        reduce(self.sums[:k * N_SUM], mask)
        _hip.check(L.psfm_shard_solve_control_async(h, int(t), k, _hip.ptr(self.sums), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev)))

    def stalled(self):
        """True once the device has got to a solve that did not go as speculated (read from pinned memory, no synchronisation)."""
        v = ctypes.c_int32(-1)
        _hip.check(_hip.lib().psfm_shard_peek_stall(self.ctx.handle, ctypes.byref(v)))
        return v.value >= 0

    def checkpoint(self, reduce):
        """One host synchronisation for all solves enqueued since the last call.  Returns None when every one of them went as
        speculated, else the frame whose solve has been redone here: the frames behind it were no-ops and must be run again."""
        if not self._pending:
            return None
        L, h = _hip.lib(), self.ctx.handle
        f_lo, f_hi = self._pending[0][0], self._pending[-1][0]
        stats = (_hip.SolveStats * (f_hi - f_lo + 1))()
        stalled = ctypes.c_int32(-1)
        _hip.check(L.psfm_shard_window_state(h, f_lo, f_hi, stats, ctypes.byref(stalled), self._sp()))
        fs = int(stalled.value)
        last_ok = f_hi if fs < 0 else fs - 1
        for t, _ in self._pending:
            if t > last_ok:
                break
            st = stats[t - f_lo]
            _hip.check(L.psfm_shard_solve_record(h, ctypes.byref(st)))
            self.counters["fused"] += 1
            self._adapt(st)
        redo = None
        if fs >= 0:
            frames = dict(self._pending)
            p = tuple(_hip.ptr(x) for x in frames[fs])
            self.counters["fused_redone"] += 1
            _hip.check(L.psfm_shard_solve_restore(h, fs, self._sp()))
            mask = [(i % N_SUM) == SUM_GMAX for i in range(N_SUM)]
            # The launch chain, a BATCH of trust-region rounds enqueued ahead (export -> sums over the ranks -> control, nothing read
            # back in between) and ONE host synchronisation per batch: rounds behind the one that ends the solve find the control block
            # done and return at once.  Every rank enqueues the same batches (8, 16, 32, 32, ...: the same totals, the same decision
            # everywhere).  (Round 4: one synchronisation per iteration -- PSFM_SHARD_ROUNDS_AHEAD=1.)
            ahead = max(1, int(os.environ.get("PSFM_SHARD_ROUNDS_AHEAD", "8")))
            kind, n = 1, 0
            done, st = ctypes.c_int32(0), _hip.SolveStats()
            while True:
                for _ in range(ahead):
                    _hip.check(L.psfm_shard_solve_export(h, *p, fs, kind, 1, _hip.ptr(self.sums), self._sp()))
                    reduce(self.sums[:N_SUM], mask)
                    _hip.check(L.psfm_shard_solve_control_chain_async(h, fs, kind, _hip.ptr(self.sums), self._sp()))
                    kind, n = 2, n + 1
                _hip.check(L.psfm_shard_solve_poll(h, ctypes.byref(done), ctypes.byref(st), self._sp()))
                if done.value:
                    break
                if n > 2 * 200 + 64:
                    raise RuntimeError("path-consistency solver did not terminate")
                ahead = min(32, ahead * 2) if ahead > 1 else 1
            _hip.check(L.psfm_shard_solve_writeback(h, fs, ctypes.byref(st), self._sp()))
            self._adapt(st)
            redo = fs
        self._pending = []
        return redo

    def _adapt(self, st):
        # the next solves speculate what the clean ones of the last 16 frames needed (same statistics, same choice on every rank)
        clean = st.dogleg_nonGN == 0 and st.termination != 5 and (st.iterations == st.successful_steps + 1 or
                                                                  (st.termination == 2 and st.iterations == st.successful_steps))
        if st.termination >= 0 and clean:
            self._need = (self._need + [min(K_MAX, st.successful_steps + 1)])[-16:]
            self.k = min(K_MAX, max(self._need) + self.k_margin)

    def finish(self):
        """the own trajectories on the HOST: (birth, length, off, xy, solve statistics)"""
        from .trajectory import _result_to_host
        if self._pending:      # (a real exception: assert is stripped under -O)
            raise RuntimeError("HipShardEngine.finish: solves enqueued since the last checkpoint()")
        info = _hip.TrackInfo()
        _hip.check(_hip.lib().psfm_shard_finish(self.ctx.handle, ctypes.byref(info), self._sp()))
        R = _result_to_host(self.ctx, info)
        return R.birth, R.length, R.off, R.xy, R.solve_stats

    def finish_device(self, ratio, width):
        """the own trajectories stay in HBM (psfm_result_device); returns (track info, their order keys as a device tensor)"""
        import torch
        if self._pending:
            raise RuntimeError("HipShardEngine.finish_device: solves enqueued since the last checkpoint()")
        info = _hip.TrackInfo()
        _hip.check(_hip.lib().psfm_shard_finish(self.ctx.handle, ctypes.byref(info), self._sp()))
        keys = torch.empty(int(info.n_traj), dtype=torch.int64, device=self.device)
        _hip.check(_hip.lib().psfm_result_keys(self.ctx.handle, int(ratio), int(width), _hip.ptr(keys), self._sp()))
        return info, keys


def flow_check_slice(f, b, thres):
    """check_fn for psfm_dist.flow_check_sharded on the GPU: (k,H,W) uint8 maps of a slice of frame pairs."""
    import torch
    from .utils import flow_check_device
    if f.shape[0] == 0:
        return torch.zeros((0,) + tuple(f.shape[1:3]), dtype=torch.uint8, device=f.device)
    _, occ = flow_check_device(f.contiguous(), b.contiguous(), thres)
    return occ.to(torch.uint8)
