#!/usr/bin/env python3
"""How fast can ~1 GB get into a tmpfs file from host memory on this box?  (The track.npy writer: 0.83 GB of points, one write() =
4.3 GB/s in round 5; parallel pwrite()s to the one file were no faster -- they serialise on the inode lock.)
    write          one write() of the whole buffer
    pwrite-N       N threads, disjoint ranges, os.pwrite
    mmap-N         ftruncate + mmap(MAP_SHARED), N threads np.copyto into disjoint ranges of the mapping (page faults instead of the
                   inode lock)
    files-N        N separate files (what a writer of several sequences at once gets)
Prints GB/s per variant, best of 3."""
import mmap
import os
import sys
import threading
import time

import numpy as np

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
base = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
n = int(GB * (1 << 30)) // 8 * 8
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
path = os.path.join(base, "psfm_write_probe.bin")


def timed(fn):
    best = 1e9
    for _ in range(3):
        for p in [path] + [path + ".%d" % k for k in range(16)]:
            if os.path.exists(p):
                os.unlink(p)
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return n / best / 1e9


def w_write():
    with open(path, "wb") as f:
        f.write(memoryview(src))


def w_pwrite(k):
    def run():
        fd = os.open(path, os.O_CREAT | os.O_WRONLY)
        os.ftruncate(fd, n)
        step = (n + k - 1) // k

        def part(i):
            lo, hi = i * step, min(n, (i + 1) * step)
            os.pwrite(fd, memoryview(src[lo:hi]), lo)
        ths = [threading.Thread(target=part, args=(i,)) for i in range(k)]
        [t.start() for t in ths]; [t.join() for t in ths]
        os.close(fd)
    return run


def w_mmap(k):
    def run():
        fd = os.open(path, os.O_CREAT | os.O_RDWR)
        os.ftruncate(fd, n)
        m = mmap.mmap(fd, n, mmap.MAP_SHARED, mmap.PROT_WRITE | mmap.PROT_READ)
        dst = np.frombuffer(m, np.uint8)
        step = ((n + k - 1) // k + 4095) // 4096 * 4096

        def part(i):
            lo, hi = i * step, min(n, (i + 1) * step)
            if lo < hi:
                np.copyto(dst[lo:hi], src[lo:hi])
        ths = [threading.Thread(target=part, args=(i,)) for i in range(k)]
        [t.start() for t in ths]; [t.join() for t in ths]
        del dst
        m.close()
        os.close(fd)
    return run


def w_files(k):
    def run():
        step = (n + k - 1) // k

        def part(i):
            lo, hi = i * step, min(n, (i + 1) * step)
            with open(path + ".%d" % i, "wb") as f:
                f.write(memoryview(src[lo:hi]))
        ths = [threading.Thread(target=part, args=(i,)) for i in range(k)]
        [t.start() for t in ths]; [t.join() for t in ths]
    return run


print("buffer %.2f GB -> %s (%d cores)" % (n / 1e9, base, os.cpu_count()))
print("write      %.2f GB/s" % timed(w_write))
for k in (2, 4, 8):
    print("pwrite-%d   %.2f GB/s" % (k, timed(w_pwrite(k))))
for k in (1, 2, 4, 8, 16):
    print("mmap-%-2d    %.2f GB/s" % (k, timed(w_mmap(k))))
for k in (2, 4, 8):
    print("files-%d    %.2f GB/s" % (k, timed(w_files(k))))
for p in [path] + [path + ".%d" % k for k in range(16)]:
    if os.path.exists(p):
        os.unlink(p)
