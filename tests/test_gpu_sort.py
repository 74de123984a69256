"""psfm_sort_records -- the record sort of the id assignment (csrc/psfm_sort.hip) on its own: equal to numpy's stable sort on sizes around
the tile and group boundaries, on every pass count, and on the digit distributions that stress its pieces (one digit only: every lane
of a wave on one counter; already sorted / reversed; a handful of distinct keys)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TILE = 4096
GROUP = 8 * TILE


def _sort(keys, vals, end_bit):
    import torch
    from point_trajectory import _hip
    ctx = _hip.context(0)
    k = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    v = torch.from_numpy(vals.copy()).cuda()
    _hip.check(_hip.lib().psfm_sort_records(ctx.handle, _hip.ptr(k), _hip.ptr(v), int(keys.size), int(end_bit), _hip.current_stream_ptr()))
    torch.cuda.synchronize()
    return k.cpu().numpy().view(np.uint32), v.cpu().numpy()


def _check(keys, end_bit):
    vals = np.arange(keys.size, dtype=np.int32)
    k, v = _sort(keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])          # equal keys keep their order: the sort is stable


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, TILE - 1, TILE, TILE + 1, GROUP - 1, GROUP, GROUP + 1, 3 * GROUP + 17, 300_001])
def test_sizes_around_tiles_and_groups(n):
    rng = np.random.default_rng(n)
    _check(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32), 32)


@pytest.mark.parametrize("end_bit", [1, 7, 8, 9, 13, 16, 17, 24, 25, 31, 32])
def test_every_pass_count_and_partial_last_digit(end_bit):
    rng = np.random.default_rng(end_bit)
    _check(rng.integers(0, 1 << end_bit, 70_001, dtype=np.uint64).astype(np.uint32), end_bit)


@pytest.mark.parametrize("kind", ["all-equal", "sorted", "reversed", "three-values", "one-digit-varies", "top-digit-only"])
def test_digit_distributions(kind):
    n = 2 * GROUP + 1234
    rng = np.random.default_rng(7)
    if kind == "all-equal":
        keys = np.full(n, 0xDEADBEEF, np.uint32)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    elif kind == "three-values":
        keys = rng.choice(np.array([5, 0x00FF00FF, 0xFFFFFFFF], np.uint32), n)
    elif kind == "one-digit-varies":
        keys = (rng.integers(0, 256, n, dtype=np.uint64).astype(np.uint32) << 8) | np.uint32(0xAB0000CD)
    else:
        keys = rng.integers(0, 256, n, dtype=np.uint64).astype(np.uint32) << 24
    _check(keys, 32)


def test_headline_size_key_layout():
    """2.07 M records with the headline key layout (13 bits of (death, birth) group over 19 bits of grid index), in the order the
    frame loop leaves them: segments of ~2 000 records, each ordered by death step."""
    rng = np.random.default_rng(11)
    n = 2_073_277
    tri = np.sort(rng.integers(0, 5253, n).reshape(-1, 1)[: (n // 2023) * 2023].reshape(-1, 2023), axis=1).reshape(-1)
    tri = np.concatenate([tri, rng.integers(0, 5253, n - tri.size)])
    keys = ((tri.astype(np.uint64) << 19) | rng.integers(0, 518_400, n).astype(np.uint64)).astype(np.uint32)
    _check(keys, 32)


def test_bad_arguments_are_refused():
    import torch
    from point_trajectory import _hip
    ctx = _hip.context(0)
    k = torch.zeros(8, dtype=torch.int32, device="cuda")
    for n, end_bit, kp, vp in [(-1, 32, k, k), (8, 0, k, k), (8, 33, k, k), (8, 32, None, k), (8, 32, k, None)]:
        st = _hip.lib().psfm_sort_records(ctx.handle, _hip.ptr(kp) if kp is not None else None, _hip.ptr(vp) if vp is not None else None,
                                          n, end_bit, _hip.current_stream_ptr())
        assert st == _hip.PSFM_ERR_ARG
    assert _hip.lib().psfm_sort_records(ctx.handle, None, None, 0, 32, _hip.current_stream_ptr()) == 0      # nothing to sort


def test_finalize_forms_kept_for_long_sequences_and_64_bit_keys_still_agree():
    """Sequences with more (death, birth) groups than the plan block holds (n_flows > 178) keep decode + scan, 64-bit keys keep rocPRIM's
    device sort; the headline shapes no longer reach either.  PSFM_FIN_PLAN=0 PSFM_FIN_SORT=0 (read once per process) sends every shape
    down those forms: the parity file, in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSFM_FIN_PLAN="0", PSFM_FIN_SORT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " passed" in r.stdout
