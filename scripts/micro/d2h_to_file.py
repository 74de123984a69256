#!/usr/bin/env python3
"""0.83 GB of device memory into a tmpfs file (what save_track_npy does with the point array), three ways:
    pinned+write   D2H into a pinned host buffer, then one write()                        (what ships)
    memmap-copy    ftruncate + np.memmap(MAP_SHARED) and tensor.copy_ from the device INTO the mapping (pageable D2H, staged by the runtime)
    registered     the mapping pinned with hipHostRegister (cudart name through torch), then an asynchronous D2H straight into the page cache
Prints seconds, best of 3."""
import ctypes
import mmap
import os
import sys
import time

import numpy as np
import torch

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 0.83
n = int(GB * (1 << 30)) // 16 * 16
path = "/dev/shm/psfm_d2h_probe.bin"
src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()


def best(fn):
    b = 1e9
    for _ in range(3):
        if os.path.exists(path):
            os.unlink(path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t0)
    return b


def pinned_write():
    pinned.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    with open(path, "wb") as f:
        f.write(memoryview(pinned.numpy()))


def memmap_copy():
    fd = os.open(path, os.O_CREAT | os.O_RDWR)
    os.ftruncate(fd, n)
    m = np.memmap(path, dtype=np.uint8, mode="r+", shape=(n,))
    torch.from_numpy(m).copy_(src)
    torch.cuda.synchronize()
    m.flush()
    del m
    os.close(fd)


def registered():
    fd = os.open(path, os.O_CREAT | os.O_RDWR)
    os.ftruncate(fd, n)
    mm = mmap.mmap(fd, n, mmap.MAP_SHARED | mmap.MAP_POPULATE, mmap.PROT_READ | mmap.PROT_WRITE)
    arr = np.frombuffer(mm, np.uint8)
    rt = torch.cuda.cudart()
    ptr = arr.ctypes.data
    rc = rt.cudaHostRegister(ptr, n, 0)
    if int(rc) != 0:
        raise RuntimeError("hipHostRegister -> %s" % rc)
    try:
        torch.from_numpy(arr).copy_(src, non_blocking=True)
        torch.cuda.synchronize()
    finally:
        rt.cudaHostUnregister(ptr)
    del arr
    mm.close()
    os.close(fd)


for name, fn in (("pinned+write", pinned_write), ("memmap-copy", memmap_copy), ("registered", registered)):
    try:
        print("%-13s %.3f s" % (name, best(fn)))
    except Exception as e:      # noqa: BLE001
        print("%-13s failed: %s: %s" % (name, type(e).__name__, str(e)[:200]))
if os.path.exists(path):
    os.unlink(path)
