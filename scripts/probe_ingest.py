#!/usr/bin/env python3
"""psfm_load_flo_stack on a stack of 1080p .flo files in /dev/shm: GB/s of .flo -> HBM for the reader / staging / copy-stream counts the
environment names (PSFM_FLO_READERS, PSFM_FLO_STAGING, PSFM_FLO_COPY_STREAMS).   python scripts/probe_ingest.py [files=100] [keep]"""
import os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
from point_trajectory.utils import write_flo, load_flows_device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
work = "/dev/shm/psfm_ingest_probe"
H, W = 1080, 1920
if not os.path.isdir(work) or len(os.listdir(work)) != n:
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    rng = np.random.default_rng(0)
    a = rng.normal(0, 1, (H, W, 2)).astype(np.float32)
    for i in range(n):
        write_flo(os.path.join(work, "%05d.flo" % i), a + np.float32(i))
gb = n * H * W * 8 / 1e9
best = 1e9
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = load_flows_device(work)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
    assert float(t[n - 1, 0, 0, 0]) != 0.0 and t.shape == (n, H, W, 2)
    del t
print("readers %s staging %s copy-streams %s: %.1f GB/s (%.1f ms for %.2f GB)" % (os.environ.get("PSFM_FLO_READERS", "8"), os.environ.get("PSFM_FLO_STAGING", "16"),
                                                                         os.environ.get("PSFM_FLO_COPY_STREAMS", "2"), gb / best, 1e3 * best, gb))
if len(sys.argv) <= 2:
    shutil.rmtree(work, ignore_errors=True)
