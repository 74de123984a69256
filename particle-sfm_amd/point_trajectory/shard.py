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
        # ONE rank (psfm_dist.connect_sharded at world size 1 -- the windowed engine for one long sequence on one GPU): the band is the
        # whole grid, nothing of a solve is exchanged, so solves whose steps get rejected take the one-GPU call's forms
        # (psfm_shard_solve_local / _redo_local: the resident solve, one launch per solve) instead of export -> exchange -> control once
        # per trust-region iteration.  mode: 0 fused solves, 1 a window whose solves are expected to reject steps (psfm_connect's rule)
        self.local = False
        self._loc = False
        self.mode = 0
        self.unroll = 4
        self._own_budget = False
        # SEVERAL ranks (round 6): windows whose solves reject steps run every solve as ONE resident launch per rank, the second hop of
        # its all-reduce crossing the ranks through peer-mapped granule rows (psfm_shard_solve_peer; connect_peers() below sets it up) --
        # no export / all-gather / control launch per trust-region iteration, no host poll.  _epoch: the tag of a solve's granules, advanced
        # by every rank for every such launch.  PSFM_SHARD_PEER=0 keeps the exchange form (A/B runs, tests of the exchange form).
        self._peer = False
        self._comm = None
        self._epoch = 0
        self._peer_maps = {}               # (pid, area pointer of that process) -> the area as this process addresses it
        self.peer_refused = None           # why the last connect_peers() kept the exchange form (a peer's area could not be mapped), if it did

    def set_local(self, on):
        self.local = bool(on) and os.environ.get("PSFM_SHARD_LOCAL", "1") != "0"

    def abort(self):
        """a run that ended in an exception: nothing of it is handed to the next one"""
        self._pending = []
        self._release_budget()

    def connect_peers(self, comm):
        """After begin(), on every rank of `comm`: agree on whether the resident solve can span the ranks (every rank's launch fits its
        share of its device) and, if so, exchange the granule areas -- raw pointers between ranks that are threads of one process, IPC
        handles between processes (same GPU or peers over xGMI)."""
        import socket
        import torch
        self._peer, self._comm = False, comm
        if comm.world < 2 or comm.world > 8 or not self.optimize or os.environ.get("PSFM_SHARD_PEER", "1") == "0":
            return False
        L, h = _hip.lib(), self.ctx.handle
        # ranks that share a device share its co-resident block slots
        where = (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"), int(self.ctx.device))
        wheres = comm.all_gather_object(where)
        sharing = sum(1 for w in wheres if w == where)
        if self.ctx.resident_budget == 0 or self._own_budget:
            self.ctx.set_resident_budget(max(self.ctx.resident_capacity() // sharing, 1))
            self._own_budget = True
        # (whatever fails on ONE rank -- the allocation, the IPC handle -- is shared with the others before anybody decides: a rank that
        # raised here would leave the rest waiting in the next collective)
        try:
            area, handle = ctypes.c_void_p(0), (ctypes.c_ubyte * 64)()
            _hip.check(L.psfm_shard_peer_area(h, ctypes.byref(area), handle, self._sp()))
            nb = ctypes.c_int32(0)
            _hip.check(L.psfm_shard_solve_blocks(h, ctypes.byref(nb)))
            # the area belongs to the CONTEXT and outlives this engine object: go on from the last epoch its granules were tagged with
            last = ctypes.c_uint32(0)
            _hip.check(L.psfm_shard_peer_epoch(h, ctypes.byref(last), 0, self._sp()))
            self._epoch = max(int(self._epoch), int(last.value))
            mine = {"pid": os.getpid(), "area": int(area.value), "handle": bytes(handle), "blocks": int(nb.value), "epoch": int(self._epoch)}
        except Exception as e:                  # noqa: BLE001
            mine = {"pid": os.getpid(), "area": 0, "handle": b"", "blocks": 0, "epoch": int(self._epoch), "error": "%s: %s" % (type(e).__name__, e)}
        infos = comm.all_gather_object(mine)
        if any(q["blocks"] < 1 for q in infos):
            self.peer_refused = next((q["error"] for q in infos if "error" in q), "a rank's resident launch does not fit its share of its device")
            return False                    # (the same answer on every rank: they all keep the exchange form)
        ptrs = (ctypes.c_void_p * comm.world)()
        failure = None
        try:
            for r, q in enumerate(infos):
                if r == comm.rank or q["pid"] == os.getpid():
                    ptrs[r] = q["area"]        # (a thread of this process: its pointer is ours)
                    continue
                key = (q["pid"], q["area"], q["handle"])
                if key not in self._peer_maps:
                    m = ctypes.c_void_p(0)
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(q["handle"])
                    _hip.check(L.psfm_shard_peer_open(h, buf, r, ctypes.byref(m)))
                    self._peer_maps = {k: v for k, v in self._peer_maps.items() if k[0] != q["pid"]}      # (psfm_shard_peer_open closed rank r's old mapping)
                    self._peer_maps[key] = int(m.value)
                ptrs[r] = self._peer_maps[key]
            blocks = (ctypes.c_int32 * comm.world)(*[q["blocks"] for q in infos])
            _hip.check(L.psfm_shard_peer_connect(h, comm.world, comm.rank, ptrs, blocks))
        except Exception as e:                 # noqa: BLE001  (no peer access between two devices, IPC refused by the driver, ...)
            failure = "%s: %s" % (type(e).__name__, e)
        # a rank that could not map a peer must not leave the others waiting for its launches: all of them, or none
        failures = comm.all_gather_object(failure)
        if any(f is not None for f in failures):
            self.peer_refused = [f for f in failures if f is not None][0]
            return False
        self._epoch = max(q["epoch"] for q in infos)      # (a rank whose earlier run was aborted, or whose context is new, catches up)
        if self._epoch > 0xFFFFF - 70000:
            # a run enqueues at most 65 535 solves: before the 20-bit count can wrap inside one, every rank zeroes its area -- nothing is
            # in flight here, and the all-gathers in front of and behind this keep a rank from launching into an area not yet cleared
            last = ctypes.c_uint32(0)
            _hip.check(L.psfm_shard_peer_epoch(h, ctypes.byref(last), 1, self._sp()))
            comm.all_gather_object(0)
            self._epoch = 0
        self._peer = True
        self.counters.update({"peer": 0, "peer_redone": 0})
        return True

    def _solve_peer(self, t, flow_prev, flow_cur, flow2_prev, occ2_prev):
        """several ranks, a window whose solves reject steps: this rank's launch of the solve of frame t over all ranks (enqueued)"""
        self._epoch = (self._epoch % 0xFFFFE) + 1            # 1 .. 2^20 - 1, the same on every rank
        _hip.check(_hip.lib().psfm_shard_solve_peer(self.ctx.handle, _hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev),
                                                    _hip.ptr(occ2_prev), int(t), int(self._epoch), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev), 2))

    def grow_tables(self):
        """psfm_dist.connect_sharded, after some rank reported PSFM_ERR_CAPACITY: the next run of this engine gets twice the lanes and
        four times the trajectory records (the factors run_connect uses for the one-GPU call); kept for the engine's later sequences"""
        self._lane_f, self._traj_f = getattr(self, "_lane_f", 2.0) * 2.0, getattr(self, "_traj_f", 8.0) * 4.0
        self.ctx.set_capacity(self._lane_f, self._traj_f)

    def _release_budget(self):
        if self._own_budget:
            self.ctx.set_resident_budget(0)
            self._own_budget = False

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
        self.counters = {"fused": 0, "fused_redone": 0, "local": 0, "local_redone": 0}
        self.G = ((W + ratio - 1) // ratio) * ((H + ratio - 1) // ratio)
        self.mode, self.unroll = 0, 4
        self._release_budget()
        self._peer = False                 # (connect_peers() decides for this run, on every rank alike)
        self._loc = self.local and bool(optimize) and int(g0) == 0 and int(g1) == self.G      # (this run)
        if self._loc and self.ctx.resident_budget == 0:
            # the resident solves of this engine run inside a budget (the calls of a sharded run come and go under the shared gate: no
            # call holds the device for the sequence) -- all of the device's slots unless the caller has set a share
            self.ctx.set_resident_budget(self.ctx.resident_capacity())
            self._own_budget = True
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
        if self._loc and self.mode == 1:
            return self._solve_local(t, flow_prev, flow_cur, flow2_prev, occ2_prev)
        if self._peer and self.mode == 1:
            return self._solve_peer(t, flow_prev, flow_cur, flow2_prev, occ2_prev)
        p = (_hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev), _hip.ptr(occ2_prev))
        k = max(1, min(K_MAX, self.k))
        mask = [(i % N_SUM) == SUM_GMAX for i in range(k * N_SUM)]
        _hip.check(L.psfm_shard_solve_export(h, *p, int(t), 0, k, _hip.ptr(self.sums), self._sp()))
        reduce(self.sums[:k * N_SUM], mask)
        _hip.check(L.psfm_shard_solve_control_async(h, int(t), k, _hip.ptr(self.sums), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev), 0))    # (keeps the frames alive for a redo)

    def _solve_local(self, t, flow_prev, flow_cur, flow2_prev, occ2_prev):
        """one rank, a window whose solves reject steps: the solve of frame t enqueued the way the one-GPU call does it (the resident
        solve -- iteration 0, every trust-region round and the write-back in ONE launch -- or `unroll` launches of one iteration)"""
        _hip.check(_hip.lib().psfm_shard_solve_local(self.ctx.handle, _hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev),
                                                     _hip.ptr(occ2_prev), int(t), int(self.unroll), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev), 1))

    def frame(self, t, flow_prev, flow_cur, occ, flow2_prev, occ2_prev, reduce_first, reduce):
        """step(t) and the fused export of solve(t) as ONE launch (psfm_shard_frame): the solve of frame t needs this rank's own
        tracks only, so it does not wait for the exchange of the frame's marks.  reduce_first(x): the exchange of the marks (the
        driver's all-reduce(max)), issued right behind the launch; then the sums over the ranks and the control step, enqueued
        like solve()'s."""
        L, h = _hip.lib(), self.ctx.handle
        assert flow_cur.is_cuda and flow_cur.is_contiguous() and occ.is_contiguous()
        if self._loc and self.mode == 1:       # (two launches: the chain step, then the solve of the frame's tracks)
            reduce_first(self.step(t, flow_cur, occ))
            return self._solve_local(t, flow_prev, flow_cur, flow2_prev, occ2_prev)
        if self._peer and self.mode == 1:      # (the chain step, the exchange of its marks, this rank's launch of the cross-rank solve)
            reduce_first(self.step(t, flow_cur, occ))
            return self._solve_peer(t, flow_prev, flow_cur, flow2_prev, occ2_prev)
        k = max(1, min(K_MAX, self.k))
        if self._loc and os.environ.get("PSFM_SHARD_LOCAL_CONTROL", "1") != "0":
            # one rank: the launch runs the control step on its own totals -- no export, no exchange, no control launch per frame
            _hip.check(L.psfm_shard_frame(h, _hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev), _hip.ptr(occ),
                                          _hip.ptr(occ2_prev), int(t), k, None, self._sp()))
            o = (int(t) & 1) * self.pitch
            reduce_first(self.maps[o:o + self.G + 1])
            self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev), 0))
            return
        mask = [(i % N_SUM) == SUM_GMAX for i in range(k * N_SUM)]
        _hip.check(L.psfm_shard_frame(h, _hip.ptr(flow_prev), _hip.ptr(flow_cur), _hip.ptr(flow2_prev), _hip.ptr(occ),
                                      _hip.ptr(occ2_prev), int(t), k, _hip.ptr(self.sums), self._sp()))
        o = (int(t) & 1) * self.pitch
        reduce_first(self.maps[o:o + self.G + 1])
        reduce(self.sums[:k * N_SUM], mask)
        _hip.check(L.psfm_shard_solve_control_async(h, int(t), k, _hip.ptr(self.sums), self._sp()))
        self._pending.append((int(t), (flow_prev, flow_cur, flow2_prev, occ2_prev), 0))

    def window_full(self):
        """one rank: a window of solves enqueued as resident launches ends after 16 frames, like the one-GPU call's -- should the flows
        have turned clean, the next window goes back to fused solves (a third of the time per solve)"""
        return (self._loc or self._peer) and self.mode == 1 and len(self._pending) >= 16

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
        if self._peer and any(how == 2 for _, _, how in self._pending):
            # every rank runs the same control step on the same totals, and a rank that gives a cross-rank solve up poisons the round for all
            # of them -- so they stall on the same frame.  Checked, not assumed: ranks that disagreed here would wait for each other for ever
            import torch
            big = 1 << 30
            v = torch.tensor([-(fs if fs >= 0 else big), fs], dtype=torch.int64, device=self.device)
            self._comm.all_reduce_max_(v)
            lo, hi = -int(v[0]), int(v[1])
            if (lo if lo < big else -1) != hi:
                raise RuntimeError("track-sharded run: the ranks disagree on the stalled solve (frames %d / %d): a cross-rank solve ended on "
                                   "one rank and was given up on another" % (lo if lo < big else -1, hi))
        last_ok = f_hi if fs < 0 else fs - 1
        seen = []                      # statistics of the window's completed solves (the redone one included)
        for t, _, how in self._pending:
            if t > last_ok:
                break
            st = stats[t - f_lo]
            _hip.check(L.psfm_shard_solve_record(h, ctypes.byref(st)))
            self.counters[("fused", "local", "peer")[how]] += 1
            self._adapt(st)
            seen.append(st)
        redo = None
        if fs >= 0 and self._loc:
            frames = {t: (x, how) for t, x, how in self._pending}
            x, how = frames[fs]
            st = _hip.SolveStats()
            self.counters["local_redone" if how else "fused_redone"] += 1
            _hip.check(L.psfm_shard_solve_redo_local(h, *(_hip.ptr(q) for q in x), fs, int(how), ctypes.byref(st), self._sp()))
            self._adapt(st)
            seen.append(st)
            redo = fs
            if os.environ.get("PSFM_SHARD_TRACE"):
                import sys
                import time
                import torch
                torch.cuda.synchronize()
                print("[shard] window %d..%d: solve %d (%s) redone locally: %d iterations, launches %s, t=%.3f" %
                      (f_lo, f_hi, fs, "resident" if how else "fused", st.iterations, self.ctx.solver_counters(), time.perf_counter()),
                      file=sys.stderr)
        elif fs >= 0:
            frames = {t: x for t, x, _ in self._pending}
            hows = {t: how for t, _, how in self._pending}
            p = tuple(_hip.ptr(x) for x in frames[fs])
            self.counters["peer_redone" if hows[fs] == 2 else "fused_redone"] += 1
            if hows[fs] == 2 and self.counters["peer_redone"] >= 2:
                # two cross-rank launches of this run gave up (the ranks' launches were not running at the same time: ranks that share a
                # device and its hardware queues, a device busy with other work): every give-up costs its spin limit, so the rest of the
                # run keeps the exchange form.  Every rank counts the same stalls, so they all decide this here.
                self._peer = False
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
            seen.append(st)
            redo = fs
        if self._loc or self._peer:
            # psfm_connect's rule (csrc/psfm_api.hip): a window with more than one solve in eight off the Gauss-Newton path sends the
            # next window to the resident solves, a clean one brings the fused solves back; launches per solve without a budget:
            # what the slowest solve of the window needed, within [4, 64]
            solved = [q for q in seen if q.termination >= 0]
            if solved:
                self.mode = 1 if 8 * sum(0 if self._clean(q) else 1 for q in solved) > len(solved) else 0
                want = min(64, max(4, max(q.iterations for q in solved) + 1))
                self.unroll = want if want > self.unroll else self.unroll - (self.unroll - want + 1) // 2
            if os.environ.get("PSFM_SHARD_TRACE") and self._loc:
                import sys
                import time
                print("[shard] window %d..%d checked: %d solves, next mode %d, k %d, t=%.3f" % (f_lo, f_hi, len(solved), self.mode, self.k,
                                                                                           time.perf_counter()), file=sys.stderr)
        self._pending = []
        return redo

    @staticmethod
    def _clean(st):
        return st.dogleg_nonGN == 0 and st.termination != 5 and (st.iterations == st.successful_steps + 1 or
                                                                 (st.termination == 2 and st.iterations == st.successful_steps))

    def _adapt(self, st):
        # the next solves speculate what the clean ones of the last 16 frames needed (same statistics, same choice on every rank)
        clean = self._clean(st)
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
        self._release_budget()
        R = _result_to_host(self.ctx, info)
        return R.birth, R.length, R.off, R.xy, R.solve_stats

    def finish_device(self, ratio, width):
        """the own trajectories stay in HBM (psfm_result_device); returns (track info, their order keys as a device tensor)"""
        import torch
        if self._pending:
            raise RuntimeError("HipShardEngine.finish_device: solves enqueued since the last checkpoint()")
        info = _hip.TrackInfo()
        _hip.check(_hip.lib().psfm_shard_finish(self.ctx.handle, ctypes.byref(info), self._sp()))
        self._release_budget()
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
