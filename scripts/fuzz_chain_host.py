#!/usr/bin/env python3
"""The device's launch chain compiled for the HOST (tests/host/pc_chain_host.cpp: psfm_pc_core.h + psfm_pc_control.h) against the
CPU oracle on many random optimize_location batches -- no GPU.  Counts solves whose trust-region DECISIONS differ (iterations,
successful steps, termination, dogleg cases) and the largest position difference among those that agree.

    python scripts/fuzz_chain_host.py [n_batches=3000] [seed=0]        -> one JSON line
"""
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from _common import solver_batch          # noqa: E402
from oracle import oracle as orc           # noqa: E402
import test_pc_chain_host as T             # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = os.path.join(tempfile.mkdtemp(), "libpc_chain_host.so")
    subprocess.run(["g++", "-O2", "-mfma", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "particle-sfm_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "pc_chain_host.cpp"), "-o", out], check=True)
    L = ctypes.CDLL(out)
    dp, fp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.pc_host_chain_solve.argtypes = [ctypes.c_long, dp, dp, dp, dp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip, dp]
    L.pc_host_chain_solve.restype = ctypes.c_int
    L.pc_host_fused_solve.argtypes = L.pc_host_chain_solve.argtypes
    L.pc_host_fused_solve.restype = ctypes.c_int
    rng = np.random.default_rng(seed)
    shapes = [(24, 31), (40, 56), (9, 120), (64, 64), (120, 160), (33, 47)]
    sigmas = [0.0, 0.02, 0.05, 0.1, 0.3, 0.6, 1.0, 2.0]
    bad, worst, its, rej, tracks, terms = [], 0.0, 0, 0, 0, {}
    fused = {"finished": 0, "handed_over": 0, "split_differs_from_oracle_statistics": 0, "differs_from_chain": 0}
    t0 = time.time()
    for b in range(n_batches):
        H, W = shapes[int(rng.integers(len(shapes)))]
        n = int(rng.integers(1, 2000)) if rng.random() < 0.8 else int(rng.integers(1, 6))
        sigma = sigmas[int(rng.integers(len(sigmas)))]
        kink = bool(rng.random() < 0.4)
        s = int(rng.integers(0, 2**31 - 1))
        uv, ref1, ref2, scale, flow12 = solver_batch(H, W, n, s, sigma, kink)
        if rng.random() < 0.2:          # far-off start values: long solves, radius changes
            uv = uv + rng.normal(0, 3.0, uv.shape)
        elif rng.random() < 0.5:        # start values near the references: the clean solves the fused form is made for
            uv = np.concatenate([ref1, ref2], 1) + rng.normal(0, 0.05, uv.shape)
        want, so = orc.optimize_location(uv, ref1, ref2, scale, flow12, return_stats=True)
        got, sg, rc = T._solve(L, uv, ref1, ref2, scale, flow12)
        same = all(sg[k] == so[k] for k in ("iterations", "successful_steps", "termination", "dogleg_nonGN")) and rc == 0
        # the speculated form (K = 3 + continuation launches): finishes exactly the clean solves, with the chain's bits
        gf, sf, rf = T._solve(L, uv, ref1, ref2, scale, flow12, fused_k=3)
        fused["finished" if rf == 0 else "handed_over"] += 1
        if so["iterations"] <= 7 and (rf == 0) != T._clean(so):
            fused["split_differs_from_oracle_statistics"] += 1
        if rf == 0 and not (np.array_equal(gf, got) and all(sf[k] == sg[k] for k in ("iterations", "successful_steps", "termination"))):
            fused["differs_from_chain"] += 1
        its += so["iterations"]; rej += so["iterations"] - so["successful_steps"]; tracks += n
        terms[so["termination"]] = terms.get(so["termination"], 0) + 1
        if same:
            worst = max(worst, float(np.abs(got - want).max()))
        else:
            bad.append({"batch": b, "shape": [H, W], "n": n, "seed": s, "sigma": sigma, "kink": kink, "oracle": so,
                        "chain": {k: sg[k] for k in ("iterations", "successful_steps", "termination", "dogleg_nonGN")},
                        "max_abs_dx": float(np.abs(got - want).max())})
    print(json.dumps({"batches": n_batches, "seed": seed, "tracks": tracks, "oracle_iterations": its, "oracle_rejected": rej,
                      "terminations": {str(k): v for k, v in sorted(terms.items())}, "decision_mismatches": len(bad), "fused_solve": fused,
                      "max_abs_dx_px_where_decisions_agree": worst, "mismatches": bad[:10], "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
