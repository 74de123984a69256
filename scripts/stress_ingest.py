#!/usr/bin/env python3
"""Randomised stress of the native .flo ingest (psfm_load_flo_stack: reader threads -> pinned ring -> async H2D): stacks of 1-70 random
frames of random small sizes on tmpfs, 1-16 reader threads, back to back on one context (the ring is re-sized between frame sizes) and
from two host threads with their own contexts at the same time; every stack compared byte for byte with the files.  Every few rounds
one file is damaged (truncated / foreign magic / other frame size) and the call must refuse it by name and leave the next call intact.

    python scripts/stress_ingest.py [rounds=200] [seed=1]"""
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from point_trajectory import _hip
from point_trajectory.utils import write_flo, load_flows_device

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base = tempfile.mkdtemp(prefix="psfm_stress_ingest_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t0 = time.time()
stats = {"stacks": 0, "files": 0, "bytes": 0, "refused": 0}
errs = []


def one_thread(tid, rounds, seed):
    rng = np.random.default_rng(seed)
    try:
        torch.cuda.set_device(0)
        for r in range(rounds):
            d = os.path.join(base, "t%d_r%d" % (tid, r))
            os.makedirs(d)
            n = int(rng.integers(1, 71))
            H, W = int(rng.integers(2, 90)), int(rng.integers(2, 130))
            frames = rng.standard_normal((n, H, W, 2)).astype(np.float32)
            for i in range(n):
                write_flo(os.path.join(d, "%05d.flo" % i), frames[i])
            readers = int(rng.integers(1, 17))
            damage = rng.random() < 0.15 and n >= 2
            if damage:
                kind = int(rng.integers(0, 3))
                # (a frame of another size: not the first file -- the stack's size IS the first file's, the refusal would name the second)
                victim = os.path.join(d, "%05d.flo" % int(rng.integers(1 if kind == 2 else 0, n)))
                raw = open(victim, "rb").read()
                if kind == 0:
                    open(victim, "wb").write(raw[:max(12, len(raw) // 2)])
                elif kind == 1:
                    open(victim, "wb").write(b"NOPE" + raw[4:])
                else:
                    write_flo(victim, rng.standard_normal((H + 1, W, 2)).astype(np.float32))
                try:
                    load_flows_device(d, n_readers=readers)
                    errs.append("thread %d round %d: a damaged file (%s, kind %d) was accepted" % (tid, r, victim, kind))
                    return
                except Exception as e:      # noqa: BLE001
                    if os.path.basename(victim) not in str(e):
                        errs.append("thread %d round %d: the refusal does not name the file: %s" % (tid, r, e))
                        return
                    stats["refused"] += 1
            else:
                got = load_flows_device(d, n_readers=readers)
                if got.shape != (n, H, W, 2) or not np.array_equal(got.cpu().numpy(), frames):
                    errs.append("thread %d round %d: %d frames of %dx%d with %d readers differ from the files" % (tid, r, n, H, W, readers))
                    return
                stats["stacks"] += 1
                stats["files"] += n
                stats["bytes"] += frames.nbytes
            shutil.rmtree(d)
    except Exception as e:      # noqa: BLE001
        errs.append("thread %d: %r" % (tid, e))
    finally:
        if tid != 0:
            _hip.release_thread_contexts()


one_thread(0, n_rounds // 2, seed)                      # back to back on one context
ths = [threading.Thread(target=one_thread, args=(t, n_rounds // 4, seed + t)) for t in (1, 2)]      # two contexts at the same time
for t in ths:
    t.start()
for t in ths:
    t.join()
shutil.rmtree(base, ignore_errors=True)
if errs:
    print("FAILED:", errs[:3])
    sys.exit(1)
print("stress_ingest: %s: every stack byte for byte the files, every damaged file refused by name, %.0f s" % (stats, time.time() - t0))
