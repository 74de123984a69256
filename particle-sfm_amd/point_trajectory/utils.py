"""Mirror of point_trajectory/utils.py: .flo reader (reference :26-56) and flow_check (:58-105)."""
import glob
import os

import numpy as np

from . import _hip

TAG_FLOAT = 202021.25


_FLO_HEADER = np.dtype([("magic", "<f4"), ("w", "<i4"), ("h", "<i4")])


def read_flo(file):
    """One Middlebury .flo file -> (h, w, 2) float32 (what utils.py:43-56 of the reference returns): a 12-byte header
    {f32 magic 202021.25, i32 width, i32 height} followed by h * w interleaved (u, v) pairs.  Like the reference it refuses
    (AssertionError) a path that is not a string, does not exist, is not named *.flo, or does not start with the magic number."""
    if not isinstance(file, str):
        raise AssertionError("read_flo: path must be a str, got %s" % type(file).__name__)
    if not os.path.isfile(file):
        raise AssertionError("read_flo: no such file: %s" % file)
    if not file.endswith(".flo"):
        raise AssertionError("read_flo: not a .flo file: %s" % file)
    with open(file, "rb") as f:
        head = np.fromfile(f, _FLO_HEADER, count=1)
        if len(head) != 1 or head["magic"][0] != np.float32(TAG_FLOAT):
            raise AssertionError("read_flo: %s does not start with the .flo magic number %r" % (file, TAG_FLOAT))
        w, h = int(head["w"][0]), int(head["h"][0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return np.resize(data, (h, w, 2))       # (np.resize, as the reference: a truncated file repeats its data instead of failing)


def _flo_shape(f, name):
    """(h, w) from the 12-byte header of an open .flo file (the same header dtype read_flo parses); AssertionError when the
    file does not start with the magic number."""
    head = np.fromfile(f, _FLO_HEADER, count=1)
    if len(head) != 1 or head["magic"][0] != np.float32(TAG_FLOAT):
        raise AssertionError("%s does not start with the .flo magic number %r" % (name, TAG_FLOAT))
    return int(head["h"][0]), int(head["w"][0])


def write_flo(file, flow):
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    with open(file, 'wb') as f:
        np.array([TAG_FLOAT], np.float32).tofile(f)
        np.array([flow.shape[1], flow.shape[0]], np.int32).tofile(f)
        flow.tofile(f)


def load_flows(dir):
    """utils.py:26-32"""
    return [read_flo(name) for name in sorted(glob.glob(dir + "/*.flo"))]


def load_flows_device_slice(dir, rank, world, device=None, **kw):
    """This rank's frame-pair slice of a stack (psfm_dist.shard_range over the sorted .flo names): what a rank of the
    track-sharded mode owns (psfm_dist.connect_sharded(..., n_flows_total=n)).  Returns (tensor (k,H,W,2), n_total) --
    1 / world of the ingest per rank instead of the whole stack on every rank."""
    import psfm_dist
    names = sorted(glob.glob(dir + "/*.flo"))
    lo, hi = psfm_dist.shard_range(len(names), int(rank), int(world))
    return load_flows_device(dir, device=device, _names=names[lo:hi], _probe=names[:1], **kw), len(names)


def load_flows_device(dir, device=None, n_staging=32, n_readers=16, _names=None, _probe=None):
    """`load_flows` (utils.py:26-32) straight into HBM: the .flo files are read by a few reader threads into pinned
    host buffers (owned by the context, reused across calls) and copied to their slot of one (n,H,W,2) device tensor with
    asynchronous H2D copies on a side stream, so disk / page-cache reads and PCIe transfers overlap (at
    cfg 4 the stacks are 26.5 GB, ingest bounds the end-to-end time once the kernels are fast).  The pipeline runs inside the
    library (psfm_load_flo_stack: no interpreter work per file -- what a stack of SMALL frames is made of); PSFM_FLO_NATIVE=0
    keeps the Python implementation of the same pipeline below (A/B runs).  A missing / foreign / truncated file raises.
    Returns a float32 device tensor (empty (0,0,0,2) if no files)."""
    import ctypes
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n_staging = int(os.environ.get("PSFM_FLO_STAGING", n_staging))      # (measurement knobs)
    n_readers = int(os.environ.get("PSFM_FLO_READERS", n_readers))
    names = sorted(glob.glob(dir + "/*.flo")) if _names is None else list(_names)
    ctx = _hip.context(device)
    dev = torch.device("cuda", ctx.device)
    if not names:
        if _probe:          # an empty slice of a non-empty stack keeps the frame shape
            with open(_probe[0], 'rb') as f:
                h, w = _flo_shape(f, _probe[0])
            return torch.zeros((0, h, w, 2), dtype=torch.float32, device=dev)
        return torch.zeros((0, 0, 0, 2), dtype=torch.float32, device=dev)

    with open(names[0], 'rb') as f:
        h, w = _flo_shape(f, names[0])
    out = torch.empty((len(names), h, w, 2), dtype=torch.float32, device=dev)
    if os.environ.get("PSFM_FLO_NATIVE", "1") != "0":
        arr = (ctypes.c_char_p * len(names))(*[os.fsencode(nm) for nm in names])
        _hip.check(_hip.lib().psfm_load_flo_stack(ctx.handle, arr, len(names), h, w, _hip.ptr(out), int(n_readers),
                                                  _hip.current_stream_ptr(ctx.device)))
        return out
    n_staging = max(2, min(int(n_staging), len(names)))
    key = (h, w, n_staging)
    cache = getattr(ctx, "_flo_staging", None)
    if cache is None or cache[0] != key:
        cache = (key, [torch.empty((h, w, 2), dtype=torch.float32).pin_memory() for _ in range(n_staging)])
        ctx._flo_staging = cache
    staging = cache[1]

    def read(i, k):
        name = names[i]
        with open(name, 'rb') as f:
            if _flo_shape(f, name) != (h, w):
                raise AssertionError("%s: frame size differs from the first file's %d x %d" % (name, w, h))
            buf = staging[k].numpy().reshape(-1)
            got = f.readinto(memoryview(buf).cast('B'))
            if got != buf.nbytes:
                raise AssertionError("%s: truncated (%d of %d data bytes)" % (name, got, buf.nbytes))
        return i

    done = [None] * n_staging          # H2D copy out of staging buffer k
    copy_stream = torch.cuda.Stream(device=dev)
    with ThreadPoolExecutor(max_workers=max(1, int(n_readers))) as pool:
        pending = {}
        nxt = 0

        def submit_upto(limit):
            nonlocal nxt
            while nxt < len(names) and nxt < limit:
                k = nxt % n_staging
                if done[k] is not None:
                    done[k].synchronize()        # the previous copy out of this staging buffer has finished
                    done[k] = None
                pending[nxt] = pool.submit(read, nxt, k)
                nxt += 1

        submit_upto(n_staging)
        for i in range(len(names)):
            pending.pop(i).result()
            k = i % n_staging
            with torch.cuda.stream(copy_stream):
                out[i].copy_(staging[k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
                done[k] = ev
            submit_upto(i + 1 + n_staging)
    torch.cuda.current_stream(dev).wait_stream(copy_stream)
    copy_stream.synchronize()            # the staging buffers belong to the context: nothing may still read them
    return out


def flow_check_device(flows, flows_b, thres, want_error=False, out=None):
    """flow_check on device tensors: (n,H,W,2) float32 stacks -> (err (n,H,W) f32 | None, occ (n,H,W) uint8; `out` if given)."""
    import torch
    ctx = _hip.context()
    n, H, W = int(flows.shape[0]), int(flows.shape[1]), int(flows.shape[2])
    if out is not None and (tuple(out.shape) != (n, H, W) or out.dtype != torch.uint8 or not out.is_contiguous()):
        raise ValueError("flow_check_device: out must be a contiguous (n,H,W) uint8 tensor")
    occ = out if out is not None else torch.empty((n, H, W), dtype=torch.uint8, device=flows.device)
    err = torch.empty((n, H, W), dtype=torch.float32, device=flows.device) if want_error else None
    _hip.check(_hip.lib().psfm_flow_check(ctx.handle, _hip.ptr(flows), _hip.ptr(flows_b), n, H, W, float(thres),
                                          _hip.ptr(occ), _hip.ptr(err), _hip.current_stream_ptr(ctx.device)))
    return err, occ


def flow_check(flows, flows_b, thres):
    """utils.py:94-105: forward/backward consistency.  Returns (error_maps, occ_maps): lists of (H,W) float32 /
    bool arrays, bit-identical to the reference's torch-CPU result."""
    from .trajectory import _as_device_stack
    import torch
    n = min(len(flows), len(flows_b))   # zip() semantics of the reference loop
    if n == 0:
        return [], []
    f = _as_device_stack(flows[:n], torch.float32, (1, 1, 2))
    b = _as_device_stack(flows_b[:n], torch.float32, (1, 1, 2))
    err, occ = flow_check_device(f, b, thres, want_error=True)
    err, occ = err.cpu().numpy(), occ.cpu().numpy().astype(bool)
    return [err[i] for i in range(n)], [occ[i] for i in range(n)]
