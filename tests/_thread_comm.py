"""Several "ranks" of psfm_dist.connect_sharded inside ONE process: one thread per rank, collectives through barriers.
Lets the GPU tests run the track-sharded mode with 2-3 ranks on the single MI355X of a test box (every thread has its own
psfm context and HIP stream; RCCL itself needs one device per rank)."""
import threading

import torch


class ThreadComm:
    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared["world"]

    @staticmethod
    def make_shared(world):
        return {"world": world, "bar": threading.Barrier(world), "slots": [None] * world}

    def _exchange(self, value):
        s = self.shared
        if value is not None and torch.is_tensor(value) and value.is_cuda:
            torch.cuda.current_stream(value.device).synchronize()      # the other threads read it on their own streams
        s["slots"][self.rank] = value
        s["bar"].wait()
        vals = list(s["slots"])
        s["bar"].wait()
        return vals

    def all_reduce_max_(self, t):
        vals = self._exchange(t.clone())
        m = vals[0].to(t.device)
        for v in vals[1:]:
            m = torch.maximum(m, v.to(t.device))
        t.copy_(m)
        return t

    def all_gather_flat(self, t):
        vals = self._exchange(t.reshape(-1).clone())
        return torch.cat([v.to(t.device) for v in vals])

    def all_gather_object(self, obj):
        return self._exchange(obj)

    def broadcast_(self, t, src, async_op=False):
        vals = self._exchange(t.clone() if self.rank == src else None)
        if self.rank != src:
            t.copy_(vals[src].to(t.device))

        class Done:
            def wait(self):
                return True
        return Done()


def run_ranks(world, fn):
    """fn(comm) on `world` threads; returns the list of results (re-raises the first exception)."""
    shared = ThreadComm.make_shared(world)
    out, err = [None] * world, []

    def worker(r):
        try:
            out[r] = fn(ThreadComm(shared, r))
        except BaseException as e:        # noqa: BLE001
            err.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return out
