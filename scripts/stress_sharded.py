#!/usr/bin/env python3
"""Randomised stress of the exact single-sequence mode: psfm_dist.connect_sharded with the HIP engine at world size 1 (the one-rank
forms: control step inside the frame launch, resident solves for windows that reject steps) and with 2 / 3 thread-ranks on the one GPU
(tests/_thread_comm.py: the exchange form) against ONE psfm_connect call on the same tensors -- random small shapes, lengths (two-flow
sequences, lengths just behind a checkpoint window), sample ratios, clean / noisy / realistic / spliced (noisy then clean) flows.
Ids, lengths, per-solve iterations / accepted steps / terminations equal; positions bit for bit without path consistency, <= 1e-9 px
with it (the order of the solver's sums follows the bands).

    python scripts/stress_sharded.py [cases=60] [seed=1]
Exit code 1 on the first difference."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import psfm_dist
import psfm_synth
from _thread_comm import run_ranks
from point_trajectory import _hip
from point_trajectory.shard import HipShardEngine, flow_check_slice
from point_trajectory.trajectory import run_connect

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
t0 = time.time()
worst = 0.0
forms = {"local": 0, "local_redone": 0, "fused": 0, "fused_redone": 0, "peer": 0, "peer_redone": 0}
for case in range(n_cases):
    H, W = int(rng.integers(30, 120)), int(rng.integers(30, 150))
    if os.environ.get("PSFM_STRESS_BIG"):
        H, W = int(rng.integers(200, 480)), int(rng.integers(300, 860))
    r = int(rng.choice([1, 2, 2, 3, 4]))
    T = int(rng.choice([3, 3, 4, 6, 9, 13, 18, 19, 26, 34, 40]))
    opt = bool(rng.random() < 0.75)
    kind = str(rng.choice(["clean", "mild", "noisy", "realistic", "spliced"]))
    seed = int(rng.integers(0, 1 << 30))
    world = int(rng.choice([1, 1, 2, 3]))
    if kind == "realistic":
        d = psfm_synth.synth_realistic(T, H, W, seed=seed, stride2=True, **psfm_synth.REALISTIC)
    elif kind == "spliced":
        a = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.4, n_occluders=2, stride2=True)
        b = psfm_synth.synth_sequence(T, H, W, seed=seed + 1, sigma=0.02, n_occluders=0, stride2=True)
        cut = int(rng.integers(1, T))
        d = {k: [a[k][t] if t < cut else b[k][t] for t in range(len(a[k]))] for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    else:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma={"clean": 0.03, "mild": 0.15, "noisy": 0.4}[kind],
                                      n_occluders=int(rng.integers(0, 3)), stride2=True)
    stack = {k: torch.from_numpy(np.stack(d[k])).to(dev) for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
    torch.cuda.synchronize()
    R = run_connect(stack["flows_f"], stack["flows_b"], stack["flows_f2"] if opt else None, stack["flows_b2"] if opt else None, 1.0, r)

    def rank_fn(comm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            try:
                eng = HipShardEngine()
                eng.ctx.set_capacity(2.0, 24.0)
                part = psfm_dist.connect_sharded(eng, stack["flows_f"], stack["flows_b"], stack["flows_f2"] if opt else None,
                                                 stack["flows_b2"] if opt else None, 1.0, r, flow_check_slice,
                                                 comm=comm)
                full = psfm_dist.gather_result(part, comm=comm)
                return part, full, dict(eng.counters)
            finally:
                _hip.release_thread_contexts()

    res = run_ranks(world, rank_fn)
    part, (birth, length, off, xy), cnt = res[0]
    ok = np.array_equal(birth, R.birth) and np.array_equal(length, R.length)
    if ok and opt:
        ok = [(s["iterations"], s["successful_steps"], s["termination"]) for s in part["solve_stats"]] == \
             [(s["iterations"], s["successful_steps"], s["termination"]) for s in R.solve_stats]
    if ok and len(xy):
        err = float(np.abs(xy - R.xy).max())
        ok = err == 0.0 if not opt else err <= 1e-9
        worst = max(worst, err)
    if not ok:
        print("DIFFERENT: case %d %dx%d T=%d r=%d %s %s world=%d seed=%d: trajectories %d vs %d, points %d vs %d, engine %s"
              % (case, H, W, T, r, "optimize" if opt else "track", kind, world, seed, len(birth), len(R.birth), len(xy), len(R.xy), cnt))
        sys.exit(1)
    for k in forms:
        forms[k] += cnt.get(k, 0)
    print("case %3d ok: %3dx%-3d T=%2d r=%d %-8s %-9s world=%d  %s" % (case, H, W, T, r, "optimize" if opt else "track", kind, world, cnt),
          flush=True)
print("stress_sharded: %d cases equal to their psfm_connect runs (positions bit for bit without path consistency, max |dxy| %.3g px "
      "with it); solves by form %s, %.0f s" % (n_cases, worst, forms, time.time() - t0))
