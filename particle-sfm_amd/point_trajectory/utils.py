"""Mirror of point_trajectory/utils.py: .flo reader (reference :26-56) and flow_check (:58-105)."""
import glob
import os

import numpy as np

from . import _hip

TAG_FLOAT = 202021.25


def read_flo(file):
    """utils.py:43-56 (Middlebury .flo: f32 magic, i32 w, i32 h, h*w*2 f32 interleaved)."""
    assert type(file) is str, "file is not str %r" % str(file)
    assert os.path.isfile(file) is True, "file does not exist %r" % str(file)
    assert file[-4:] == '.flo', "file ending is not .flo %r" % file[-4:]
    with open(file, 'rb') as f:
        flo_number = np.fromfile(f, np.float32, count=1)[0]
        assert flo_number == TAG_FLOAT, 'Flow number %r incorrect. Invalid .flo file' % flo_number
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return np.resize(data, (h, w, 2))


def write_flo(file, flow):
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    with open(file, 'wb') as f:
        np.array([TAG_FLOAT], np.float32).tofile(f)
        np.array([flow.shape[1], flow.shape[0]], np.int32).tofile(f)
        flow.tofile(f)


def load_flows(dir):
    """utils.py:26-32"""
    return [read_flo(name) for name in sorted(glob.glob(dir + "/*.flo"))]


def load_flows_device(dir, device=None, n_staging=3):
    """`load_flows` (utils.py:26-32) straight into HBM: every .flo is read into a pinned host buffer and copied
    to its slot of one (n,H,W,2) device tensor with an asynchronous H2D copy on a side stream, so the disk read of
    file i+1 overlaps the PCIe transfer of file i (at cfg 4 the stacks are 26.5 GB, ingest bounds the
    end-to-end time once the kernels are fast).  Returns a float32 device tensor (empty (0,0,0,2) if no files)."""
    import torch
    names = sorted(glob.glob(dir + "/*.flo"))
    ctx = _hip.context(device)
    dev = torch.device("cuda", ctx.device)
    if not names:
        return torch.zeros((0, 0, 0, 2), dtype=torch.float32, device=dev)

    def header(name):
        with open(name, 'rb') as f:
            tag = np.fromfile(f, np.float32, count=1)[0]
            assert tag == TAG_FLOAT, 'Flow number %r incorrect. Invalid .flo file' % tag
            w = int(np.fromfile(f, np.int32, count=1)[0])
            h = int(np.fromfile(f, np.int32, count=1)[0])
        return h, w

    h, w = header(names[0])
    out = torch.empty((len(names), h, w, 2), dtype=torch.float32, device=dev)
    staging = [torch.empty((h, w, 2), dtype=torch.float32).pin_memory() for _ in range(max(2, int(n_staging)))]
    done = [None] * len(staging)
    copy_stream = torch.cuda.Stream(device=dev)
    for i, name in enumerate(names):
        k = i % len(staging)
        if done[k] is not None:
            done[k].synchronize()            # the previous copy out of this staging buffer has finished
        hh, ww = header(name)
        assert (hh, ww) == (h, w), "flow size mismatch in %r" % name
        with open(name, 'rb') as f:
            f.seek(12)
            buf = staging[k].numpy().reshape(-1)
            got = f.readinto(memoryview(buf).cast('B'))
            assert got == buf.nbytes, "truncated .flo file %r" % name
        with torch.cuda.stream(copy_stream):
            out[i].copy_(staging[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
            done[k] = ev
    torch.cuda.current_stream(dev).wait_stream(copy_stream)
    return out


def flow_check_device(flows, flows_b, thres, want_error=False):
    """flow_check on device tensors: (n,H,W,2) float32 stacks -> (err (n,H,W) f32 | None, occ (n,H,W) uint8)."""
    import torch
    ctx = _hip.context()
    n, H, W = int(flows.shape[0]), int(flows.shape[1]), int(flows.shape[2])
    occ = torch.empty((n, H, W), dtype=torch.uint8, device=flows.device)
    err = torch.empty((n, H, W), dtype=torch.float32, device=flows.device) if want_error else None
    _hip.check(_hip.lib().psfm_flow_check(ctx.handle, _hip.ptr(flows), _hip.ptr(flows_b), n, H, W, float(thres),
                                          _hip.ptr(occ), _hip.ptr(err), _hip.current_stream_ptr()))
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
