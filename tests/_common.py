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
    d = psfm_synth.synth_sequence(int(g["T"]), int(g["H"]), int(g["W"]), seed=int(g["seed"]),
                                  sigma=float(g["sigma"]) if "sigma" in g else 0.05,
                                  n_occluders=int(g["n_occluders"]) if "n_occluders" in g else 0,
                                  stride2=stride2)
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
