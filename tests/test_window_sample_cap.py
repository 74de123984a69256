"""TrajectorySet::sample_inside_window ABOVE max_num_tracks (optimize/src/trajectory_base.cpp:150-153: `std::random_shuffle` of the
eligible ids + `resize(max_num_tracks)`, unseeded in the reference -- so no fixture can pin WHICH subset comes out).  What the branch
guarantees, and what is checked here on the host class (CPU suite) and on psfm_window_sample (`-m gpu`):
  * exactly max_num_tracks rows;
  * their ids are distinct and a subset of the eligible ids (= the ids of the uncapped call);
  * every row equals the uncapped call's row of the same id -- locations, masks / absence masks, normalised coordinates;
  * the order is a shuffle (not ascending), another seed picks another subset, the same seed the same one (device: seeded)."""
import numpy as np
import pytest

import psfm_synth


def _sequence(T=14, H=40, W=56, seed=3):
    return psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.3, n_occluders=2, stride2=False)


def test_host_class_above_the_cap():
    from oracle import oracle as orc
    from point_trajectory.optimize.build.particlesfm import Trajectory, TrajectorySet
    d = _sequence()
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = orc.track(d["flows_f"], occ, 2)
    ts = TrajectorySet()
    for i in range(R.n_traj):
        xy = R.xy[R.off[i]:R.off[i + 1]]
        ts.insert(i, Trajectory(times=list(range(int(R.birth[i]), int(R.birth[i]) + len(xy))), xys=[tuple(p) for p in xy]))
    ts.build_invert_indexes()
    frames = list(range(3, 11))
    full = ts.sample_inside_window(frames, 3, 10 ** 9)
    K_all = len(full["traj_ids"])
    cap = K_all // 3
    assert cap > 50
    row_of = {t: k for k, t in enumerate(full["traj_ids"])}
    np.random.seed(1)
    a = ts.sample_inside_window(frames, 3, cap)
    np.random.seed(2)
    b = ts.sample_inside_window(frames, 3, cap)
    for s in (a, b):
        ids = s["traj_ids"]
        assert len(ids) == cap == len(set(ids)) and set(ids) <= set(full["traj_ids"])
        rows = [row_of[t] for t in ids]
        assert np.array_equal(s["locations"][0], full["locations"][0][rows]) and np.array_equal(s["locations"][1], full["locations"][1][rows])
        assert np.array_equal(s["masks"], full["masks"][rows]) and s["masks"].shape == (cap, len(frames))
        assert ids != sorted(ids)
    assert a["traj_ids"] != b["traj_ids"]
    exact = ts.sample_inside_window(frames, 3, K_all)          # at the cap, not above it: untouched, ascending
    assert exact["traj_ids"] == full["traj_ids"]


@pytest.mark.gpu
def test_device_window_sample_above_the_cap():
    import torch
    from point_trajectory import _hip, trajectory
    from psfm_motion_seg.load_cut_seq import sample_window_device
    d = _sequence(T=18, H=60, W=84, seed=5)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    trajectory.run_connect(ff, fb, None, None, 1.0, 2, return_device=True)
    ctx = _hip.context()
    f0, n, raw_hw, inp = 4, 9, (60, 84), (32, 48)
    ids_all, raw_all, nor_all, mask_all = [t.cpu().numpy() for t in sample_window_device(ctx, f0, n, raw_hw, inp, 10 ** 9)]
    K_all = len(ids_all)
    assert K_all > 300 and np.array_equal(ids_all, np.sort(ids_all))
    row_of = {int(t): k for k, t in enumerate(ids_all)}
    picked = {}
    for cap in (K_all // 4, K_all - 1, 1):
        for seed in (0, 1, 1):
            ids, raw, nor, mask = [t.cpu().numpy() for t in sample_window_device(ctx, f0, n, raw_hw, inp, cap, seed=seed)]
            assert ids.shape == (cap,) and raw.shape == (cap, n, 2) and nor.shape == (cap, n, 2) and mask.shape == (cap, n, 1)
            assert len(set(ids.tolist())) == cap and set(ids.tolist()) <= set(row_of)
            rows = [row_of[int(t)] for t in ids]
            assert np.array_equal(raw, raw_all[rows]) and np.array_equal(nor, nor_all[rows]) and np.array_equal(mask, mask_all[rows])
            if (cap, seed) in picked:
                assert np.array_equal(picked[(cap, seed)], ids)          # seeded: the same subset in the same order again
            picked[(cap, seed)] = ids
    big = K_all // 4
    assert not np.array_equal(picked[(big, 0)], np.sort(picked[(big, 0)]))     # a shuffle, not the first rows
    assert set(picked[(big, 0)].tolist()) != set(picked[(big, 1)].tolist())     # another seed, another subset
    ids, _, _, _ = sample_window_device(ctx, f0, n, raw_hw, inp, K_all)          # at the cap: untouched, ascending
    assert np.array_equal(ids.cpu().numpy(), ids_all)
