"""psfm_connect_batch: B same-shape sequences through ONE launch per frame (blockIdx.y = sequence), one checkpoint per window of
frames and one segmented finalize for the whole batch -- the loop of the reference's driver over a directory of sequences
(run_particlesfm.py:168-176).  Every sequence's result must be what psfm_connect / the oracle give for it alone: ids and lengths
bit-exact, positions bit-exact in track mode and within 1e-4 px with path consistency (in practice ~1e-12: same control flow,
another summation order), every solve's iterations and termination equal."""
import numpy as np
import pytest

import psfm_synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def pt():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from point_trajectory import trajectory, _hip
    _hip.context()
    class NS: pass
    ns = NS()
    ns.trajectory, ns.hip, ns.torch = trajectory, _hip, torch
    return ns


def _dev(pt, d, opt):
    t = pt.torch
    f = lambda k: t.from_numpy(np.stack(d[k])).cuda() if len(d[k]) else t.zeros((0,) + d["flows_f"][0].shape, dtype=t.float32, device="cuda")
    return (f("flows_f"), f("flows_b"), f("flows_f2") if opt else None, f("flows_b2") if opt else None)


def _oracle(d, thres, r, opt):
    from oracle import oracle as orc
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], thres)
    if not opt:
        return orc.track(d["flows_f"], occ, r)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], thres)
    return orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)


def _check(pt, ctxs, infos, oracles, opt, exact=True):
    for k, (ctx, info, O) in enumerate(zip(ctxs, infos, oracles)):
        R = pt.trajectory._result_to_host(ctx, info)
        assert len(R) == O.n_traj, (k, len(R), O.n_traj)
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length), k
        assert np.array_equal(R.off, np.concatenate([[0], np.cumsum(O.length)])), k
        if exact and not opt:
            assert np.array_equal(R.xy, O.xy), (k, float(np.abs(R.xy - O.xy).max()))
        else:
            assert float(np.abs(R.xy - O.xy).max()) <= TOL, k
        if opt:
            assert [s["iterations"] for s in R.solve_stats] == [s["iterations"] for s in O.solves], k
            assert [s["termination"] for s in R.solve_stats] == [s["termination"] for s in O.solves], k
            assert [s["successful_steps"] for s in R.solve_stats] == [s["successful_steps"] for s in O.solves], k


# (H, W, sample_ratio, [(frames, seed, sigma, occluders) per sequence])
TRACK_BATCHES = [
    (120, 200, 2, [(9, 1, 0.05, 1), (6, 2, 0.3, 3), (12, 3, 0.1, 2), (2, 4, 0.05, 0), (9, 5, 0.6, 2)]),
    (96, 130, 1, [(8, 11, 0.05, 1), (8, 12, 0.35, 2), (5, 13, 0.1, 0)]),
    (150, 210, 4, [(9, 21, 0.05, 2)] * 1 + [(7, 22, 0.4, 3)]),
    (135, 240, 3, [(10, 31 + k, 0.05 + 0.05 * k, k % 3) for k in range(9)]),
    (480, 854, 4, [(7, 41, 0.05, 2), (5, 42, 0.3, 3), (7, 43, 0.05, 2), (7, 44, 0.15, 1)]),      # configs[0] shape
]


@pytest.mark.parametrize("H,W,r,seqs", TRACK_BATCHES)
def test_track_batch_vs_oracle(pt, H, W, r, seqs):
    """--skip_path_consistency: flow_check + track for every sequence of a batch -- different seeds, noise levels, lengths (a
    sequence that ends early turns its blocks into no-ops) -- bit-exact against the oracle, sequence by sequence."""
    data = [psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sg, n_occluders=no, stride2=False) for (T, seed, sg, no) in seqs]
    oracles = [_oracle(d, 1.0, r, False) for d in data]
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, False) for d in data], 1.0, r)
    assert all(int(i.chain_mode) == 3 for i in infos)
    _check(pt, ctxs, infos, oracles, False)
    # ... and again on the warm workspaces, in another order (the contexts keep their buffers; nothing may leak from the run before)
    order = list(range(len(data)))[::-1]
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, data[k], False) for k in order], 1.0, r)
    _check(pt, ctxs, infos, [oracles[k] for k in order], False)


def test_track_batch_all_tracks_die(pt):
    """SciPy's empty-map corner (trajectory.py:150-152: no survivor at all -> every grid point but (0,0) respawns) inside a batch:
    one sequence whose second flow throws every track out of the image, beside an ordinary one."""
    H, W, r = 60, 80, 2
    a = psfm_synth.synth_sequence(6, H, W, seed=51, sigma=0.05, n_occluders=1, stride2=False)
    b = psfm_synth.synth_sequence(6, H, W, seed=52, sigma=0.05, n_occluders=1, stride2=False)
    b["flows_f"][1][:] = 500.0
    data = [a, b, a]
    oracles = [_oracle(d, 1.0, r, False) for d in data]
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, False) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, False)


OPT_BATCHES = [
    (120, 200, 2, [(9, 61, 0.05, 2), (7, 62, 0.03, 1), (12, 63, 0.05, 2), (2, 64, 0.05, 0), (3, 65, 0.05, 1)]),
    (64, 96, 1, [(20, 71, 0.05, 1), (14, 72, 0.08, 1)]),
    (436, 1024, 2, [(8, 81, 0.05, 2), (6, 82, 0.05, 2), (8, 83, 0.04, 1)]),                      # configs[2] shape
]


@pytest.mark.parametrize("H,W,r,seqs", OPT_BATCHES)
def test_track_optimize_batch_vs_oracle(pt, H, W, r, seqs):
    """flow_check x2 + track_optimize for a batch: every sequence with its own device-side program counter and K (lengths differ:
    one sequence is through while others go on; a two-flow sequence has a single solve, a one-flow sequence none)."""
    data = [psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sg, n_occluders=no, stride2=True) for (T, seed, sg, no) in seqs]
    oracles = [_oracle(d, 1.0, r, True) for d in data]
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, True) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, True)
    if r != 1:       # (clean flows on a coarse grid: the sequences stay in the batch; the dense 64 x 96 ones reject steps and leave it)
        assert sum(int(i.chain_mode) == 3 for i in infos) >= len(data) - 1, [int(i.chain_mode) for i in infos]


def test_track_optimize_batch_with_sequences_that_reject_steps(pt):
    """A batch that mixes clean sequences with ones whose solves reject steps (sigma 0.3, occluders: what the launch chain is for):
    the hard ones have their first stalled solve redone at the checkpoint and then leave the batch to run alone behind it; the
    clean ones stay.  Every sequence equals the oracle's."""
    H, W, r = 120, 200, 2
    spec = [(9, 91, dict(sigma=0.05, n_occluders=1)), (8, 92, psfm_synth.HARD), (24, 93, dict(sigma=0.04, n_occluders=0)),
            (7, 94, psfm_synth.HARD), (20, 95, dict(sigma=0.05, n_occluders=0))]
    data = [psfm_synth.synth_sequence(T, H, W, seed=seed, stride2=True, **kw) for (T, seed, kw) in spec]
    oracles = [_oracle(d, 1.0, r, True) for d in data]
    assert sum(s["iterations"] - s["successful_steps"] for s in oracles[1].solves) > 0
    for c in pt.hip.batch_contexts(len(data)):
        c.set_solver(0, 0)
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, True) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, True)
    modes = [int(i.chain_mode) for i in infos]
    assert modes[2] == 3 and modes[4] == 3, modes          # (smooth flows without occluder boxes: nothing to reject)
    assert modes[1] != 3 and modes[3] != 3, modes          # (the hard sequences ran alone)


def test_batch_of_one_and_consumers_on_a_member(pt):
    """A batch of one sequence is psfm_connect; and a batch member's context serves the consumers like any other: the saved set
    (main_connect_point_trajectories.py:56-61, psfm_result_filter) of member 1 equals the host-side filter of its result."""
    H, W, r = 90, 140, 2
    data = [psfm_synth.synth_sequence(8, H, W, seed=101 + k, sigma=0.2, n_occluders=2, stride2=False) for k in range(3)]
    oracles = [_oracle(d, 1.0, r, False) for d in data]
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, data[0], False)], 1.0, r)
    _check(pt, ctxs, infos, oracles[:1], False)
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, False) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, False)
    R = pt.trajectory._result_to_host(ctxs[1], infos[1])
    want = R.to_trajectory_set(3)
    got = pt.trajectory.result_to_trajectory_set(ctxs[1], infos[1], 3)
    for x, y in zip(want._to_csr()[:4], got._to_csr()[:4]):       # ids, offsets, frame of every point, positions
        assert np.array_equal(np.asarray(x), np.asarray(y))


@pytest.mark.parametrize("opt", [False, True])
def test_batch_launches_that_cover_too_few_lanes_are_run_again(pt, monkeypatch, opt):
    """The batched frame launches cover the lanes a sequence can be expected to use (grid + 1/8), not its whole lane table; a launch
    that finds more lanes in use than it covers raises overflow bit 16 and the batch runs again with launches that cover the tables.
    PSFM_BATCH_GRID_LANES=512 makes every launch too small for these grids (6000 points): same results as ever."""
    H, W, r = 120, 200, 2
    data = [psfm_synth.synth_sequence(7 + k, H, W, seed=111 + k, sigma=0.05, n_occluders=1 - k % 2, stride2=opt) for k in range(3)]
    oracles = [_oracle(d, 1.0, r, opt) for d in data]
    monkeypatch.setenv("PSFM_BATCH_GRID_LANES", "512")
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, opt) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, opt)


def test_batch_grows_its_tables_when_a_sequence_needs_more(pt):
    """PSFM_ERR_CAPACITY from one sequence of the batch (here: trajectory-record tables sized 1 x the grid for sequences that kill and
    respawn half of their tracks every frame): the mirror grows the tables of all contexts and runs the batch again."""
    H, W, r = 150, 210, 1
    data = [psfm_synth.synth_sequence(13, H, W, seed=121 + k, sigma=0.9, n_occluders=3, stride2=False) for k in range(3)]
    oracles = [_oracle(d, 1.0, r, False) for d in data]
    assert oracles[0].n_traj > 2.5 * H * W + 70000        # (more records than tables of 1.5 x + 1 x the grid + their head-room hold)
    ctxs = pt.hip.batch_contexts(3)
    ctxs[0]._capacity = {("batch", H, W, r, False): (1.0, 1.0)}
    ctxs, infos = pt.trajectory.run_connect_batch([_dev(pt, d, False) for d in data], 1.0, r)
    _check(pt, ctxs, infos, oracles, False)
    assert ctxs[0]._capacity[("batch", H, W, r, False)][1] > 1.0


@pytest.mark.parametrize("opt", [False, True])
def test_connect_sequences_in_batches_writes_the_same_files(pt, tmp_path, opt):
    """point_trajectory.batch.connect_sequences(batch=N): the reference driver's loop over a directory of sequences
    (run_particlesfm.py:168-176) disk to disk -- .flo files in, one track.npy per sequence out -- through psfm_connect_batch on two
    worker threads: every file holds what main_connect_point_trajectories writes for that sequence alone."""
    import os
    from point_trajectory.batch import connect_sequences
    from point_trajectory.main_connect_point_trajectories import main_connect_point_trajectories
    from point_trajectory.trajectory import load_track_npy
    from point_trajectory.utils import write_flo
    H, W, r = 60, 80, 2
    fdirs, tdirs, sdirs = [], [], []
    for k in range(5):
        d = psfm_synth.synth_sequence(6 + k % 3, H, W, seed=131 + k, sigma=0.1, n_occluders=1, stride2=opt)
        fd = tmp_path / ("seq%d" % k) / "flows"
        for name, key in (("flow_f", "flows_f"), ("flow_b", "flows_b"), ("flow_f2", "flows_f2"), ("flow_b2", "flows_b2")):
            if key not in d:
                continue
            os.makedirs(fd / name)
            for i, a in enumerate(d[key]):
                write_flo(str(fd / name / ("%05d.flo" % i)), a)
        fdirs.append(str(fd)); tdirs.append(str(tmp_path / ("seq%d" % k) / "traj")); sdirs.append(str(tmp_path / ("seq%d" % k) / "single"))
    connect_sequences(fdirs, tdirs, sample_ratio=r, skip_path_consistency=not opt, concurrency=2, rank=0, world=1, layout="reference", batch=3)
    for fd, td, sd in zip(fdirs, tdirs, sdirs):
        main_connect_point_trajectories(fd, sd, sample_ratio=r, skip_path_consistency=not opt)
        a, b = load_track_npy(os.path.join(td, "track.npy")), load_track_npy(os.path.join(sd, "track.npy"))
        for x, y in zip(a._to_csr()[:4], b._to_csr()[:4]):
            if np.asarray(x).dtype == np.float64 and opt:
                assert float(np.abs(np.asarray(x) - np.asarray(y)).max()) <= 1e-9
            else:
                assert np.array_equal(np.asarray(x), np.asarray(y))


def test_batch_rejects_bad_arguments(pt):
    import ctypes
    H, W = 40, 56
    d = psfm_synth.synth_sequence(4, H, W, seed=1, stride2=False)
    s = _dev(pt, d, False)
    ctxs = pt.hip.batch_contexts(2)
    L = pt.hip.lib()
    vp = ctypes.c_void_p
    h2 = (vp * 2)(ctxs[0].handle, ctxs[0].handle)        # the same context twice
    ff, fb = (vp * 2)(s[0].data_ptr(), s[0].data_ptr()), (vp * 2)(s[1].data_ptr(), s[1].data_ptr())
    nf = (ctypes.c_int * 2)(3, 3)
    st = L.psfm_connect_batch(h2, 2, ff, fb, None, None, nf, H, W, 1.0, 2, None, pt.hip.current_stream_ptr())
    assert st == pt.hip.PSFM_ERR_ARG and b"twice" in L.psfm_last_error()
    h2 = (vp * 2)(ctxs[0].handle, ctxs[1].handle)
    nf = (ctypes.c_int * 2)(3, 0)
    assert L.psfm_connect_batch(h2, 2, ff, fb, None, None, nf, H, W, 1.0, 2, None, pt.hip.current_stream_ptr()) == pt.hip.PSFM_ERR_ARG
    assert L.psfm_connect_batch(h2, 65, ff, fb, None, None, nf, H, W, 1.0, 2, None, pt.hip.current_stream_ptr()) == pt.hip.PSFM_ERR_ARG
