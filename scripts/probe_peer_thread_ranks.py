#!/usr/bin/env python3
"""ONE 1080p sequence over N thread-ranks that share the box's single GPU, every rank with its own psfm context and HIP stream inside
ONE process (so their kernels really run side by side; two PROCESSES on one device are time-sliced against each other -- a resident
launch that waits for its peer then waits for a context switch: 495 ms per hard sequence, profiles/r06).  The collectives of the
driver (marks of a frame, the exchange form's sums) are done ON THE DEVICE through events -- no stream is ever synchronised with the
host, like RCCL on real ranks.  Times: the cross-rank resident solve (psfm_shard_solve_peer), the exchange form (PSFM_SHARD_PEER=0),
ONE psfm_connect call on the same tensors.  One JSON line.

    python scripts/probe_peer_thread_ranks.py [world=2] [frames=101] [dist=hard|clean] [forms=peer,exchange]

All ranks share one device's CUs, L2s and HBM: the figure prices the hand-off protocol, it is not a multi-GPU speed-up."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

H, W, RATIO, THRES = 1080, 1920, 2, 1.0


class EventComm:
    """collectives of psfm_dist.connect_sharded between threads of one process, ordered by HIP events instead of host synchronisation"""

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared["world"]

    @staticmethod
    def make_shared(world):
        return {"world": world, "bar": threading.Barrier(world), "a": [None] * world, "b": [None] * world}

    def _publish(self, key, value):
        ev = torch.cuda.Event()
        ev.record()
        self.s[key][self.rank] = (value, ev)
        self.s["bar"].wait()
        vals = list(self.s[key])
        st = torch.cuda.current_stream()
        for q, (_, e) in enumerate(vals):
            if q != self.rank:
                st.wait_event(e)
        return [v for v, _ in vals]

    def _done_reading(self):
        # the others' streams must not overwrite what this rank's stream is still reading
        self._publish("b", None)

    def all_reduce_max_(self, t):
        # in place and without a second rendezvous: a rank that reads another's buffer while that one is already folding the maximum in
        # sees, element by element, either the old value or the maximum -- the same result either way; the engine's maps are double-buffered
        # by frame parity and every frame's exchange orders the streams, so a buffer is not rewritten before everybody has read it
        vals = self._publish("a", t)
        for q, v in enumerate(vals):
            if q != self.rank:
                torch.maximum(t, v, out=t)
        return t

    def all_gather_flat(self, t):
        vals = self._publish("a", t.reshape(-1))
        out = torch.cat([v for v in vals])
        self._done_reading()
        return out

    def all_gather_object(self, obj):
        self.s["a"][self.rank] = (obj, None)
        self.s["bar"].wait()
        vals = [v for v, _ in self.s["a"]]
        self.s["bar"].wait()
        return vals

    def broadcast_(self, t, src, async_op=False):
        vals = self._publish("a", t)
        if self.rank != src:
            t.copy_(vals[src])
        self._done_reading()

        class Done:
            def wait(self):
                return True
        return Done()


def run(world=2, frames=101, dist_name="hard", forms=("peer", "exchange"), reps=3, seed=6):
    import psfm_dist
    import psfm_synth
    from point_trajectory import _hip
    from point_trajectory.shard import HipShardEngine, flow_check_slice
    from point_trajectory.trajectory import run_connect
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.cuda.set_device(dev)
    kw = {"hard": psfm_synth.HARD, "clean": dict(sigma=0.05, n_occluders=2)}[dist_name]
    d = psfm_synth.synth_sequence_torch(frames, H, W, seed=seed, stride2=True, device=dev, **kw)
    torch.cuda.synchronize()
    out = {}
    saved = os.environ.get("PSFM_SHARD_PEER")
    try:
        for form in forms:
            os.environ["PSFM_SHARD_PEER"] = "1" if form == "peer" else "0"
            shared = EventComm.make_shared(world)
            res, err = [None] * world, []

            def rank_fn(r):
                try:
                    torch.cuda.set_device(dev)
                    comm = EventComm(shared, r)
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        eng = HipShardEngine(_hip.Context(dev.index or 0))
                        ms = []
                        for rep in range(reps):
                            torch.cuda.current_stream().synchronize(); shared["bar"].wait()
                            t0 = time.perf_counter()
                            part = psfm_dist.connect_sharded(eng, d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO,
                                                             flow_check_slice, comm=comm, keep_on_device=True)
                            torch.cuda.current_stream().synchronize(); shared["bar"].wait()
                            ms.append(1e3 * (time.perf_counter() - t0))
                        res[r] = {"ms": ms, "counters": dict(eng.counters), "n_traj": int(part["n_traj"]), "iterations": int(part["solver_iterations"])}
                except BaseException as e:      # noqa: BLE001
                    err.append(e)
                    shared["bar"].abort()
                finally:
                    _hip.release_thread_contexts()
            ths = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
            [t.start() for t in ths]; [t.join() for t in ths]
            if err:
                raise err[0]
            out[form] = res[0]
    finally:
        if saved is None:
            os.environ.pop("PSFM_SHARD_PEER", None)
        else:
            os.environ["PSFM_SHARD_PEER"] = saved
    ms = []
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = run_connect(d["flows_f"], d["flows_b"], d["flows_f2"], d["flows_b2"], THRES, RATIO, return_device=True)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    best = lambda k: min(out[k]["ms"])
    rec = {"frames": frames, "flows": dist_name, "world": world, "ranks": "threads of one process on ONE GPU, device-side collectives (HIP events)",
           "psfm_connect_ms": min(ms), "all_ms": {"psfm_connect": ms},
           "note": "all ranks share one device: this prices the hand-off protocol, it is not a multi-GPU speed-up; no xGMI link is crossed"}
    for form in forms:
        rec[form + "_ms"] = best(form)
        rec[form + "_over_psfm_connect"] = best(form) / min(ms)
        rec["counters_" + form] = out[form]["counters"]
        rec["all_ms"][form] = out[form]["ms"]
    rec["same_counts"] = all(out[f]["n_traj"] == int(info.n_traj) and out[f]["iterations"] == int(info.solver_iterations) for f in forms)
    return rec


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 101,
                         sys.argv[3] if len(sys.argv) > 3 else "hard",
                         forms=tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("peer", "exchange"))))
