"""Shared helpers for the parity tests."""
import hashlib
import os

import numpy as np

import psfm_synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def input_hash(d):
    h = hashlib.sha256()
    for k in sorted(d):
        for a in d[k]:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def regen_inputs(g, stride2):
    """Re-synthesise a fixture's inputs from its seed and check they are the bytes it was made from."""
    if "realistic" in g:
        d = psfm_synth.synth_realistic(int(g["T"]), int(g["H"]), int(g["W"]), seed=int(g["seed"]), stride2=stride2, **psfm_synth.REALISTIC)
        assert input_hash(d) == str(g["input_hash"]), "psfm_synth.synth_realistic no longer reproduces this fixture's inputs"
        return d
    d = psfm_synth.synth_sequence(int(g["T"]), int(g["H"]), int(g["W"]), seed=int(g["seed"]),
                                  amp=float(g["amp"]) if "amp" in g else 3.0,
                                  sigma=float(g["sigma"]) if "sigma" in g else 0.05,
                                  n_occluders=int(g["n_occluders"]) if "n_occluders" in g else 0,
                                  stride2=stride2,
                                  drift=tuple(float(x) for x in g["drift"]) if "drift" in g else (0.0, 0.0),
                                  warp_b=bool(int(g["warp_b"])) if "warp_b" in g else False)
    assert input_hash(d) == str(g["input_hash"]), "psfm_synth no longer reproduces this fixture's inputs"
    return d


def assert_csr_equal(birth, length, xy, g, tol=0.0):
    """ids / lengths bit-exact; positions within tol px (0 -> bit-exact)."""
    assert birth.shape[0] == g["birth"].shape[0], (birth.shape[0], g["birth"].shape[0])
    assert np.array_equal(birth, g["birth"])
    assert np.array_equal(length, g["length"])
    assert xy.shape == g["xy"].shape
    if tol == 0.0:
        assert np.array_equal(xy, g["xy"]), float(np.abs(xy - g["xy"]).max())
    else:
        assert float(np.abs(xy - g["xy"]).max()) <= tol


def solver_batch(H, W, n, seed, sigma, kink=False):
    """A batch for optimize_location: (uv12 (n,4), ref1, ref2, scale (n,1), flow12 (H,W,2) f32).  kink: points outside
    the image too (Grid2D clamping); 20 % of the scales are exactly 0."""
    rng = np.random.default_rng(seed)
    d = psfm_synth.synth_sequence(3, H, W, seed=seed, sigma=sigma, stride2=True)
    flow12 = d["flows_f"][1]
    p0 = rng.uniform([-2, -2], [W + 1, H + 1], size=(n, 2)) if kink else rng.uniform([2, 2], [W - 3, H - 3], size=(n, 2))
    ref1 = p0 + rng.normal(0, 1.0, size=(n, 2))
    ref2 = ref1 + rng.normal(0, 1.5, size=(n, 2))
    uv = np.concatenate([ref1 + rng.normal(0, 0.5, (n, 2)), ref2 + rng.normal(0, 0.5, (n, 2))], 1)
    scale = rng.uniform(0, 1, size=(n, 1)).astype(np.float32).astype(np.float64)
    scale[rng.uniform(size=n) < 0.2] = 0.0
    return uv, ref1, ref2, scale, flow12


# reference-python fixtures with ~10 px of drift per frame: stride-2 flows on both sides of the 20 px gate of trajectory.py:179,
# fractional occ02 weights, tracks leaving the image (tests/golden/make_golden.py::make_large_motion)
LARGE_MOTION = ["opt_largemotion_96x128_r2", "opt_largemotion_90x140_r3"]

# reference-python fixtures on psfm_synth.REALISTIC (tests/golden/make_golden.py::make_realistic): depth-ordered layers with true
# (dis)occlusion, correlated flow error, outlier blobs -- solves that reject steps at the motion boundaries
REALISTIC_OPT = ["opt_realistic_96x160_r2", "opt_realistic_84x132_r1"]
REALISTIC_TRACK = ["track_realistic_100x150_r2"]

# the batches every solver test runs (GPU vs oracle, oracle vs the second restatement, real Ceres when available)
SOLVER_BATCHES = [
    (60, 80, 3000, 1, 0.02, False),
    (60, 80, 3000, 2, 0.5, False),      # noisy flow: rejected steps, dogleg interpolation
    (45, 70, 5000, 3, 0.3, True),       # points outside the image: Grid2D clamping
    (270, 480, 100000, 4, 0.05, False),
    (33, 47, 1, 5, 0.1, False),         # a single track
]
