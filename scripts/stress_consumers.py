#!/usr/bin/env python3
"""Randomised stress of traj_to_matches (sfm/matches_from_flow.py:51-118) on the device against its host form: psfm_result_filter ->
psfm_traj_to_matches -> psfm_matches_copy against psfm_sfm.matches_from_flow.match_tables_host (pinned by the reference's own function in
tests/test_consumers_golden.py) on the same trajectories -- random sequences (3-40 frames, random small shapes, sample ratios, noise),
random dynamic labels (none / sparse / dense / all), traj_min_len 1-6, sample_k 2-15, image lists longer than the sequence: the six
tables element for element.

    python scripts/stress_consumers.py [cases=60] [seed=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, _result_to_host
from psfm_sfm import matches_from_flow as mff

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
tot = {"cases": 0, "keypoints": 0, "matches": 0, "pairs": 0}
NAMES = ("kp_off", "kp_xy", "pair_key", "pair_off", "pair_first", "rows")
for case in range(n_cases):
    H, W = int(rng.integers(24, 120)), int(rng.integers(24, 160))
    T = int(rng.choice([3, 4, 5, 8, 13, 21, 40]))
    r = int(rng.choice([1, 2, 3, 4, 6]))
    sigma = float(rng.choice([0.03, 0.2, 0.5, 1.0]))
    d = psfm_synth.synth_sequence(T, H, W, seed=int(rng.integers(0, 1 << 30)), sigma=sigma, n_occluders=int(rng.integers(0, 4)), stride2=False)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    ctx = _hip.context()
    info = run_connect(ff, fb, None, None, 1.0, r, return_device=True)
    R = _result_to_host(ctx, info)
    min_len = int(rng.integers(1, 7))
    sample_k = int(rng.integers(2, 16))
    keep = np.flatnonzero(R.length >= min_len)
    off = np.zeros(len(keep) + 1, np.int64)
    np.cumsum(R.length[keep], out=off[1:])
    frames = (np.concatenate([np.arange(R.birth[i], R.birth[i] + R.length[i]) for i in keep]).astype(np.int64)
              if len(keep) else np.zeros(0, np.int64))
    xy = np.concatenate([R.xy[R.off[i]:R.off[i + 1]] for i in keep], 0) if len(keep) else np.zeros((0, 2))
    mode = int(rng.integers(0, 4))
    labels = (np.zeros(len(frames), bool) if mode == 0 else rng.random(len(frames)) < (0.05 if mode == 1 else 0.6)
              if mode < 3 else np.ones(len(frames), bool))
    n_img = T + int(rng.integers(0, 3))
    want = mff.match_tables_host(off, frames, xy, labels, n_img, remove_dynamic=True, sample_k=sample_k)
    lab_dev = torch.from_numpy(labels.astype(np.uint8)).cuda() if mode != 0 or rng.random() < 0.5 else None
    got = mff.match_tables_device(ctx, n_img, traj_min_len=min_len, sample_k=sample_k, labels=lab_dev)
    for nm, a, b in zip(NAMES, want, got):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or not np.array_equal(a, b):
            print("DIFFERENT: case %d %dx%d T=%d r=%d sigma=%.2f min_len=%d sample_k=%d labels mode %d: table %s (%s vs %s)"
                  % (case, H, W, T, r, sigma, min_len, sample_k, mode, nm, a.shape, b.shape))
            sys.exit(1)
    tot["cases"] += 1
    tot["keypoints"] += len(want[1])
    tot["matches"] += len(want[5])
    tot["pairs"] += len(want[2])
print("stress_consumers: %s: the device's match tables equal the host's element for element, %.0f s" % (tot, time.time() - t0))
