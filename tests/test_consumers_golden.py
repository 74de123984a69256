"""The consumers of track.npy (SURVEY f-3 traj_to_matches, f-4 motion-seg window tensors) against golden vectors that the
REFERENCE's own functions produced (tests/golden/make_consumer_golden.py: sfm/matches_from_flow.py:51-118 and
motion_seg/load_cut_seq.py:25-89 + core/dataset/data_utils.py:74-89, imported unmodified in the build container).

CPU part (runs anywhere): the host tables of psfm_sfm.matches_from_flow from the oracle's trajectories.
GPU part (-m gpu): psfm_traj_to_matches and psfm_window_sample straight from the result the HIP path left in HBM.
"""
import numpy as np
import pytest

from _common import golden, regen_inputs

MATCH_CASES = ["matches_40x56_t12", "matches_24x32_t27_dyn"]


def _pairs_in_dict_order(datas, names):
    """[(src image, tgt image, rows)] in the order the reference's dicts hold them (image order, then first use)."""
    out = []
    for i, n in enumerate(names):
        for key, rows in datas[n].match_pairs.items():
            a, b = key.split("-")
            out.append((names.index(a), names.index(b), np.asarray(rows, np.int32).reshape(-1, 2)))
    return out


def _check_against_fixture(datas, names, g):
    kp = [np.asarray(datas[n].keypoints, np.float64).reshape(-1, 2) for n in names]
    assert np.array_equal(np.cumsum([0] + [len(k) for k in kp]), g["kp_off"])
    assert np.array_equal(np.concatenate(kp, 0), g["kp_xy"])
    pairs = _pairs_in_dict_order(datas, names)
    assert [p[0] for p in pairs] == g["pair_src"].tolist() and [p[1] for p in pairs] == g["pair_tgt"].tolist()
    assert np.array_equal(np.cumsum([0] + [len(p[2]) for p in pairs]), g["pair_off"])
    assert np.array_equal(np.concatenate([p[2] for p in pairs], 0), g["rows"])


@pytest.mark.parametrize("name", MATCH_CASES)
def test_host_match_tables_equal_reference_fixture(name, tmp_path):
    from oracle import oracle as orc
    from psfm_sfm import matches_from_flow as mff
    g = golden(name)
    d = regen_inputs(g, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = orc.track(d["flows_f"], occ, int(g["ratio"]))
    keep = np.flatnonzero(R.length >= 3)
    assert len(keep) == int(g["n_saved"])
    off = np.zeros(len(keep) + 1, np.int64)
    np.cumsum(R.length[keep], out=off[1:])
    frames = np.concatenate([np.arange(R.birth[i], R.birth[i] + R.length[i]) for i in keep]).astype(np.int64)
    xy = np.concatenate([R.traj(int(i))[1] for i in keep], 0)
    labels = np.unpackbits(g["labels"])[:int(g["n_points"])].astype(bool)
    T = int(g["T"])
    names = ["%05d.png" % i for i in range(T)]
    tables = mff.match_tables_host(off, frames, xy, labels, T, remove_dynamic=True)
    _check_against_fixture(mff.assemble(names, tables, str(tmp_path / "pairs.txt")), names, g)


def test_host_match_tables_reject_frames_outside_the_image_list():
    """The reference indexes image_names[frame] (sfm/matches_from_flow.py:79) and raises IndexError on a frame beyond the image
    list; the device path returns PSFM_ERR_ARG.  The host tables must not silently drop or misfile such points."""
    from psfm_sfm import matches_from_flow as mff
    off = np.array([0, 3], np.int64)
    xy = np.zeros((3, 2))
    labels = np.zeros(3, bool)
    for frames in (np.array([0, 1, 4], np.int64), np.array([-1, 0, 1], np.int64)):
        with pytest.raises(IndexError):
            mff.match_tables_host(off, frames, xy, labels, 4, remove_dynamic=True)
    mff.match_tables_host(off, np.array([1, 2, 3], np.int64), xy, labels, 4, remove_dynamic=True)


@pytest.fixture(scope="module")
def pt():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from point_trajectory import utils, trajectory, _hip
    _hip.context()
    class NS: pass
    ns = NS()
    ns.utils, ns.trajectory, ns.hip = utils, trajectory, _hip
    return ns


@pytest.mark.gpu
@pytest.mark.parametrize("name", MATCH_CASES)
def test_device_traj_to_matches_equals_reference_fixture(pt, name, tmp_path):
    """psfm_result_filter -> psfm_traj_to_matches -> psfm_matches_copy: keypoints per image, match rows per image pair and
    the dict order of the pairs, element for element what the reference's traj_to_matches returned."""
    import torch
    from psfm_sfm import matches_from_flow as mff
    g = golden(name)
    d = regen_inputs(g, stride2=False)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    ctx = pt.hip.context()
    info = pt.trajectory.run_connect(ff, fb, None, None, 1.0, int(g["ratio"]), return_device=True)
    assert int(info.n_traj) >= int(g["n_saved"])
    labels = np.unpackbits(g["labels"])[:int(g["n_points"])].astype(np.uint8)
    T = int(g["T"])
    names = ["%05d.png" % i for i in range(T)]
    for lab in ((torch.from_numpy(labels).cuda(),) if labels.any() else (None, torch.from_numpy(labels).cuda())):
        datas = mff.traj_to_matches_device(ctx, names, str(tmp_path / "pairs.txt"), traj_min_len=3, labels=lab)
        _check_against_fixture(datas, names, g)
    import hashlib
    assert hashlib.sha256(open(str(tmp_path / "pairs.txt")).read().encode()).hexdigest() == str(g["pair_file_hash"])
    # frames beyond the image list are an argument error, not a silent truncation
    with pytest.raises(pt.hip.PsfmError):
        mff.match_tables_device(ctx, T - 2)


@pytest.mark.gpu
def test_device_window_tensors_equal_reference_fixture(pt):
    """psfm_window_sample per window = the reference's load_cut_seq (window cutting, min_length 3, sample_inside_window)
    + resize_point_traj / normalize_point_traj: raw and normalised coordinates, absence masks, frame and trajectory ids."""
    import torch
    from psfm_motion_seg.load_cut_seq import cut_trajectory_windows
    g = golden("windows_48x64_t23")
    d = regen_inputs(g, stride2=False)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    pt.trajectory.run_connect(ff, fb, None, None, 1.0, int(g["ratio"]), return_device=True)
    T, H, W = int(g["T"]), int(g["H"]), int(g["W"])
    input_size = tuple(int(x) for x in g["input_size"])
    for tag, win in (("w", int(g["window"])), ("full", T + 5)):
        raw_b, nor_b, mask_b, time_b, idx_b = cut_trajectory_windows(T, win, (H, W), input_size, traj_max_num=10 ** 9, as_numpy=True)
        assert len(raw_b) == int(g[tag + "_n"])
        for w in range(len(raw_b)):
            assert np.array_equal(idx_b[w].astype(np.int64), g["%s%d_ids" % (tag, w)])
            assert np.array_equal(time_b[w], g["%s%d_time" % (tag, w)])
            assert np.array_equal(raw_b[w], g["%s%d_raw" % (tag, w)])
            assert np.array_equal(nor_b[w], g["%s%d_nor" % (tag, w)])
            assert np.array_equal(mask_b[w], g["%s%d_mask" % (tag, w)])
