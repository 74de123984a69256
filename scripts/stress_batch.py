#!/usr/bin/env python3
"""Randomised stress of psfm_connect_batch: batches of random small shapes, lengths, sample ratios, flow distributions (clean, noisy,
realistic, mixed within one batch), batch sizes 1..12 -- every sequence of every batch compared with its own psfm_connect run: ids,
lengths, per-solve iterations / accepted steps / terminations equal; positions BIT FOR BIT without path consistency, and to <= 1e-9 px
with it: a dogleg step's coefficients come from sums over the tracks, whose order follows the lanes the tracks were born on (popped from
shared stacks in atomic order) and the blocks of the launch (a sequence that left the batch runs on a SHARE of the resident block
slots) -- two runs of the SAME call differ in the last bits of a position near zero now and then (seed 5, batch 8: one point of 58 751,
2e-19 px).  Same decisions always.
No oracle: the single-sequence call is pinned against it elsewhere (tests/test_gpu_*.py); this looks for anything the batch form does
differently.

    python scripts/stress_batch.py [batches=40] [seed=1] [only=<batch index>]
Prints one line per batch and a summary; exit code 1 on the first difference.  only: run that batch alone (the random stream is consumed
as in the full run) and print every sequence's counts from the batch, from psfm_connect and from the CPU oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np
import torch
import psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, run_connect_batch, _result_to_host

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
t_start = time.time()
n_seq_total = n_left = 0
worst = 0.0
for b in range(n_batches):
    H, W = int(rng.integers(24, 140)), int(rng.integers(24, 180))
    if os.environ.get("PSFM_STRESS_BIG"):      # (DAVIS / Sintel-sized frames: launches of several hundred blocks, trimmed grids)
        H, W = int(rng.integers(200, 480)), int(rng.integers(300, 860))
    r = int(rng.choice([1, 2, 2, 3, 4]))
    opt = bool(rng.random() < 0.65)
    B = int(rng.integers(1, 13))
    if os.environ.get("PSFM_STRESS_B"):      # (e.g. 64: the largest batch the entry point takes)
        B = int(os.environ["PSFM_STRESS_B"])
    thres = float(rng.choice([1.0, 1.0, 3.0]))
    kinds, seqs, data = [], [], []
    for k in range(B):
        T = int(rng.integers(3, 40))
        kind = rng.choice(["clean", "clean", "noisy", "realistic", "mild"])
        seed = int(rng.integers(0, 1 << 30))
        n_occ = int(rng.integers(0, 3)) if kind != "realistic" else 0
        if only >= 0 and b != only:
            continue
        if kind == "realistic":
            d = psfm_synth.synth_realistic(T, H, W, seed=seed, stride2=True, **psfm_synth.REALISTIC)
        else:
            sigma = {"clean": 0.03, "mild": 0.15, "noisy": 0.4}[kind]
            d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=n_occ, stride2=True)
        dd = {k2: torch.from_numpy(np.stack(d[k2])).cuda() if len(d[k2]) else torch.zeros((0, H, W, 2), dtype=torch.float32, device="cuda")
              for k2 in ("flows_f", "flows_b", "flows_f2", "flows_b2")}
        data.append(dd)
        kinds.append("%s%d" % (kind[0], T))
        seqs.append((dd["flows_f"], dd["flows_b"], dd["flows_f2"] if opt else None, dd["flows_b2"] if opt else None))
    if only >= 0 and b != only:
        continue
    ctxs, infos = run_connect_batch(seqs, thres, r)
    got = [_result_to_host(c, i) for c, i in zip(ctxs, infos)]
    modes = [int(i.chain_mode) for i in infos]
    if only >= 0:
        from oracle import oracle as orc
        for k in range(B):
            dn = {k2: [x.cpu().numpy() for x in data[k][k2]] for k2 in data[k]}
            _, occ = orc.flow_check(dn["flows_f"], dn["flows_b"], thres)
            if opt:
                _, occ2 = orc.flow_check(dn["flows_f2"], dn["flows_b2"], thres) if len(dn["flows_f2"]) else (None, [])
                O = orc.track_optimize(dn["flows_f"], dn["flows_f2"], occ, occ2, r)
            else:
                O = orc.track(dn["flows_f"], occ, r)
            R = run_connect(*seqs[k], thres, r)
            print("  sequence %d (%s, mode %d): trajectories / points  batch %d / %d   psfm_connect %d / %d   oracle %d / %d" %
                  (k, kinds[k], modes[k], len(got[k].birth), len(got[k].xy), len(R.birth), len(R.xy), O.n_traj, len(O.xy)), flush=True)
            if os.environ.get("PSFM_STRESS_FORMS") and opt:      # the single-sequence call in each of its solver forms against the batch's bits
                ctx = _hip.context()
                for label, env, solver in (("adaptive", {}, (0, 0)), ("launch chain / resident", {}, (1, 0)), ("fused", {}, (2, 0)),
                                           ("launch chain, launches only", {"PSFM_PC_PERSIST": "0"}, (1, 0)),
                                           ("resident, iteration 0 apart", {"PSFM_PC_INIT_INSIDE": "0"}, (1, 0)),
                                           ("adaptive, launches only", {"PSFM_PC_PERSIST": "0"}, (0, 0))):
                    for kk, vv in env.items():
                        os.environ[kk] = vv
                    ctx.set_solver(*solver)
                    R2 = run_connect(*seqs[k], thres, r)
                    ctx.set_solver(0, 0)
                    for kk in env:
                        del os.environ[kk]
                    nd = int((R2.xy != got[k].xy).any(1).sum()) if R2.xy.shape == got[k].xy.shape else -1
                    print("      %-30s vs batch: %d points differ, max |dxy| %.3g; vs oracle %.3g   %s" %
                          (label, nd, float(np.abs(R2.xy - got[k].xy).max()) if nd >= 0 else -1, float(np.abs(R2.xy - O.xy).max()),
                           ctx.solver_counters()), flush=True)
            if len(R.birth) != O.n_traj:      # the single-sequence call under its switches: which form of it is off?
                ctx = _hip.context()
                for label, env, solver in (("default again", {}, (0, 0)), ("launch chain", {}, (1, 0)), ("fused K=4", {}, (2, 4)),
                                           ("host-paced frames", {"PSFM_SEQ": "0"}, (0, 0)), ("two launches", {"PSFM_MERGE_FRAME": "0"}, (0, 0)),
                                           ("no resident", {"PSFM_PC_PERSIST": "0"}, (1, 0)), ("chain mode 1", {}, (0, 0))):
                    for kk, vv in env.items():
                        os.environ[kk] = vv
                    ctx.set_solver(*solver)
                    if label == "chain mode 1":
                        ctx.set_chain_mode(1)
                    R2 = run_connect(*seqs[k], thres, r)
                    print("      %-18s %d / %d   %s" % (label, len(R2.birth), len(R2.xy), ctx.solver_counters()), flush=True)
                    ctx.set_solver(0, 0)
                    ctx.set_chain_mode(0)
                    for kk in env:
                        del os.environ[kk]
                for prev in range(B):         # which predecessor on the same context sets it off?
                    R0 = run_connect(*seqs[prev], thres, r)
                    c0 = ctx.solver_counters()
                    os.environ["PSFM_TRACE"] = "1"
                    R2 = run_connect(*seqs[k], thres, r)
                    del os.environ["PSFM_TRACE"]
                    print("      behind sequence %d (%s; its counters %s): %d / %d   %s" % (prev, kinds[prev], c0, len(R2.birth), len(R2.xy),
                                                                                         ctx.solver_counters()), flush=True)
    for k in range(B):
        try:
            R = run_connect(*seqs[k], thres, r)
        except Exception as e:      # noqa: BLE001  (say which sequence, and whether the call fails again / in its other forms)
            print("EXCEPTION: batch %d sequence %d (%s) %dx%d r=%d opt=%s thres=%.1f kinds=%s modes=%s: %s" % (b, k, kinds[k], H, W, r, opt, thres, kinds, modes, e))
            ctx = _hip.context()
            print("   counters", ctx.solver_counters())
            for label, env in (("again", {}), ("host-paced frames", {"PSFM_SEQ": "0"}), ("again", {}), ("window of 4 frames", {"PSFM_CHECK_FRAMES": "4"})):
                os.environ.update(env)
                try:
                    R2 = run_connect(*seqs[k], thres, r)
                    print("   %-20s ok: %d / %d %s, equal to the batch's: %s" % (label, len(R2.birth), len(R2.xy), ctx.solver_counters(),
                                                                               np.array_equal(R2.birth, got[k].birth) and np.array_equal(R2.length, got[k].length)))
                except Exception as e2:      # noqa: BLE001
                    print("   %-20s fails: %s" % (label, e2))
                for q in env:
                    del os.environ[q]
            sys.exit(1)
        G = got[k]
        ok = np.array_equal(R.birth, G.birth) and np.array_equal(R.length, G.length) and np.array_equal(R.off, G.off)
        if ok and not opt:
            ok = np.array_equal(R.xy, G.xy)
        elif ok and len(R.xy):
            worst = max(worst, float(np.abs(R.xy - G.xy).max()))
            ok = float(np.abs(R.xy - G.xy).max()) <= 1e-9
        if ok and opt:
            ok = [(s["iterations"], s["successful_steps"], s["termination"]) for s in R.solve_stats] == \
                 [(s["iterations"], s["successful_steps"], s["termination"]) for s in G.solve_stats]
        if not ok:
            print("DIFFERENT: batch %d sequence %d (%s) %dx%d r=%d opt=%s thres=%.1f kinds=%s modes=%s: traj %d vs %d, points %d vs %d"
                  % (b, k, kinds[k], H, W, r, opt, thres, kinds, modes, len(R.birth), len(G.birth), len(R.xy), len(G.xy)))
            if len(R.xy) == len(G.xy) and len(R.xy):
                print("  max |dxy| = %g" % float(np.abs(R.xy - G.xy).max()))
                bad = np.nonzero((R.xy != G.xy).any(1))[0]
                print("  %d points differ; solver counters of the psfm_connect run %s" % (len(bad), _hip.context().solver_counters()))
                for p in bad[:12]:
                    tr = int(np.searchsorted(R.off, p, side="right") - 1)
                    print("    point %d = trajectory %d (birth %d, length %d) at time %d: psfm_connect %r batch %r" %
                          (p, tr, R.birth[tr], R.length[tr], R.birth[tr] + p - R.off[tr], R.xy[p].tolist(), G.xy[p].tolist()))
                if opt:
                    print("  solves (iterations, accepted): %s" % [(q["iterations"], q["successful_steps"]) for q in R.solve_stats])
            sys.exit(1)
    n_seq_total += B
    n_left += sum(1 for m in modes if m != 3)
    print("batch %3d ok: %3dx%-3d r=%d %s thres=%.0f B=%2d  %s  left the batch: %d" %
          (b, H, W, r, "optimize" if opt else "track   ", thres, B, " ".join(kinds), sum(1 for m in modes if m != 3)), flush=True)
print("stress_batch: %d batches, %d sequences (%d ran outside their batch): ids / lengths / solver decisions equal to their psfm_connect "
      "runs, positions bit-identical without path consistency, max |dxy| %.3g px with it, %.0f s"
      % (n_batches, n_seq_total, n_left, worst, time.time() - t_start))
