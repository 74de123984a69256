#!/usr/bin/env python3
"""Randomised stress of SEVERAL HOST THREADS on one GPU: every thread has its own context and runs its own list of random sequences
(track / track_optimize, clean / noisy / realistic flows, 3-30 frames, random small shapes) at the same time as the others, in one of the
ways a host can set its workers up --
    default         contexts as created: whoever finds the device gate free runs its rejecting solves as resident launches, the
                    others with one launch per trust-region iteration; holders yield at their checkpoints
    budgets         every context a share of the resident block slots (psfm_ctx_set_resident_budget): resident solves side by side
    over-budgets    shares that add up to TWICE the capacity: launches that do not become co-resident give up and are redone
    chain-mode-1    psfm_ctx_set_chain_mode(ctx, 1): launches only
    mixed           thread 0 runs psfm_connect_batch on groups of its sequences while the others run psfm_connect
-- and every result is compared with the same sequence run alone on an idle device: ids, lengths, per-solve iterations / accepted
steps / terminations equal, positions bit for bit without path consistency, <= 1e-9 px with it.

    python scripts/stress_threads.py [rounds=10] [seed=1]
Exit code 1 on the first difference or exception."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, run_connect_batch, _result_to_host

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_start = time.time()
MODES = ["default", "budgets", "over-budgets", "chain-mode-1", "mixed"]
totals = {"sequences": 0, "resident_launches": 0, "resident_giveups": 0, "iteration_launches": 0}
worst = 0.0


def make(H, W, T, kind, seed, n_occ):
    if kind == "realistic":
        d = psfm_synth.synth_realistic(T, H, W, seed=seed, stride2=True, **psfm_synth.REALISTIC)
    else:
        d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma={"clean": 0.03, "mild": 0.15, "noisy": 0.4}[kind], n_occluders=n_occ, stride2=True)
    return {k: torch.from_numpy(np.stack(d[k])).cuda() for k in ("flows_f", "flows_b", "flows_f2", "flows_b2")}


def summary(R, opt):
    return (R.birth.copy(), R.length.copy(), R.xy.copy(),
            [(s["iterations"], s["successful_steps"], s["termination"]) for s in R.solve_stats] if opt else None)


def same(a, b, opt):
    global worst
    if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3]):
        return False
    if len(a[2]) == 0:
        return True
    err = float(np.abs(a[2] - b[2]).max())
    worst = max(worst, err) if opt else worst
    return err <= 1e-9 if opt else err == 0.0


for rnd in range(n_rounds):
    mode = MODES[rnd % len(MODES)]
    n_thr = int(rng.integers(2, 5))
    opt = bool(rng.random() < 0.8)
    r = int(rng.choice([1, 2, 2, 3, 4]))
    work = []
    for t in range(n_thr):
        H, W = (int(rng.integers(40, 160)), int(rng.integers(40, 220)))
        seqs = []
        for _ in range(int(rng.integers(2, 6))):
            T = int(rng.choice([3, 4, 7, 12, 18, 19, 30]))
            kind = str(rng.choice(["clean", "mild", "noisy", "realistic"]))
            dd = make(H, W, T, kind, int(rng.integers(0, 1 << 30)), int(rng.integers(0, 3)))
            seqs.append((kind + str(T), (dd["flows_f"], dd["flows_b"], dd["flows_f2"] if opt else None, dd["flows_b2"] if opt else None)))
        work.append(seqs)
    torch.cuda.synchronize()
    # every sequence alone on an idle device
    want = [[summary(run_connect(*s, 1.0, r), opt) for _, s in seqs] for seqs in work]
    got = [[None] * len(seqs) for seqs in work]
    cnts, errs = [None] * n_thr, []
    start = threading.Barrier(n_thr)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            ctx = _hip.context()
            cap = ctx.resident_capacity()
            if mode == "budgets":
                ctx.set_resident_budget(max(1, cap // n_thr))
            elif mode == "over-budgets":
                ctx.set_resident_budget(max(1, 2 * cap // n_thr))
            elif mode == "chain-mode-1":
                ctx.set_chain_mode(1)
            start.wait()
            acc = {"resident_launches": 0, "resident_giveups": 0, "iteration_launches": 0}
            with torch.cuda.stream(torch.cuda.Stream()):
                for rep in range(2):
                    if mode == "mixed" and t == 0:
                        ctxs, infos = run_connect_batch([s for _, s in work[t]], 1.0, r)
                        for k, (c, i) in enumerate(zip(ctxs, infos)):
                            got[t][k] = summary(_result_to_host(c, i), opt)
                    else:
                        for k, (_, s) in enumerate(work[t]):
                            got[t][k] = summary(run_connect(*s, 1.0, r), opt)
                            c = ctx.solver_counters()
                            for q in acc:
                                acc[q] += c[q]
            cnts[t] = acc
        except Exception as e:      # noqa: BLE001
            errs.append((t, repr(e)))
            try:
                start.abort()
            except Exception:       # noqa: BLE001
                pass
        finally:
            _hip.release_thread_contexts()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errs:
        print("EXCEPTION in round %d (%s, %d threads): %s" % (rnd, mode, n_thr, errs))
        sys.exit(1)
    for t in range(n_thr):
        for k in range(len(work[t])):
            if not same(want[t][k], got[t][k], opt):
                print("DIFFERENT: round %d (%s, %d threads, r=%d, %s) thread %d sequence %d (%s): trajectories %d vs %d, points %d vs %d"
                      % (rnd, mode, n_thr, r, "optimize" if opt else "track", t, k, work[t][k][0], len(want[t][k][0]), len(got[t][k][0]),
                         len(want[t][k][2]), len(got[t][k][2])))
                sys.exit(1)
            totals["sequences"] += 1
        for q in ("resident_launches", "resident_giveups", "iteration_launches"):
            totals[q] += cnts[t][q]
    print("round %2d ok: %-13s %d threads r=%d %-8s %s   %s" % (rnd, mode, n_thr, r, "optimize" if opt else "track",
                                                              " | ".join(" ".join(n for n, _ in seqs) for seqs in work), cnts), flush=True)
print("stress_threads: %d rounds, %s: every result equal to the same sequence run alone (positions bit for bit without path consistency, "
      "max |dxy| %.3g px with it), %.0f s" % (n_rounds, totals, worst, time.time() - t_start))
