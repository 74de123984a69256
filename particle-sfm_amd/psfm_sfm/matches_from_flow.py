"""Trajectories -> per-image keypoints and sampled pair matches for the COLMAP database: the consumer
sfm/matches_from_flow.py:51-118 of the reference, as array work instead of its per-trajectory /
per-point Python loops (O(sum(len) * K) interpreter steps).

Two producers feed ONE assembler:
  * `traj_to_matches(img_dir, traj_dir, match_list_file, remove_dynamic=True)` -- the reference's signature; reads
    track.npy and does the index arithmetic in NumPy on the host;
  * `traj_to_matches_device(ctx, image_names, match_list_file)` -- the same tables computed by
    psfm_traj_to_matches (csrc/psfm_matches.hip) straight from the saved set that psfm_result_filter left in HBM; only
    the finished tables cross PCIe.
Both return what the reference returns: {image name: object with .keypoints and .match_pairs} in image order, and write
the pair list file.  Element for element equal to the reference's own function (tests/test_reference_consumers.py,
tests/golden/matches_*.npz).

What the reference computes, as tables:
  keypoints   image i lists the points observed in frame i in trajectory (id) order (:67-81); a point's keypoint index
              is its rank in that list
  matches     point j of a trajectory with n kept points pairs with every other point when n <= K = 20, otherwise with
              the K points at k * (n // K), itself skipped (:83-101); a match is filed under the ordered image pair
              (frame of j, frame of the target) as the row [keypoint index of j, keypoint index of the target], rows in
              the order the loops produce them (trajectory, j, k); the pairs of an image appear in order of first use.
"""
import os

import numpy as np

SAMPLE_K = 20     # sfm/matches_from_flow.py:52


class ImageMatches:
    """What sfm/import_feature_matches.py:76-104 reads per image: `.keypoints` ((n,2) coordinates) and `.match_pairs`
    ({"<image name>-<image name>": rows of [own keypoint index, other keypoint index]})."""
    __slots__ = ("image_id", "keypoints", "match_pairs")

    def __init__(self, image_id, keypoints=None):
        self.image_id = int(image_id)
        self.keypoints = [] if keypoints is None else keypoints
        self.match_pairs = {}


imageMatchData = ImageMatches   # the reference's name for this container (sfm/matches_from_flow.py:21)


def _flatten(trajectories):
    """TrajectorySet (CSR or map backed) or plain dict -> (off, frames, xy, labels) in iteration (id) order."""
    if hasattr(trajectories, "_to_csr"):
        ids, off, frames, xy, labels = trajectories._to_csr()
        if labels is None:
            labels = np.zeros(len(frames), bool)
        return off, frames, xy, labels
    cnt, fr, loc, lab = [], [], [], []
    for key in trajectories:                      # dict written by motion_seg (main_motion_segmentation.py:122-129)
        t = trajectories[key]
        f = np.asarray(t["frame_ids"], np.int64)
        cnt.append(len(f))
        fr.append(f)
        loc.append(np.asarray(t["locations"], np.float64).reshape(-1, 2))
        lab.append(np.asarray(t["labels"]).astype(bool))
    off = np.zeros(len(cnt) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    cat = lambda a, shape, dt: np.concatenate(a) if a else np.zeros(shape, dt)
    return off, cat(fr, (0,), np.int64), cat(loc, (0, 2), np.float64), cat(lab, (0,), bool)


def match_tables_host(off, frames, xy, labels, n_img, remove_dynamic=True, sample_k=SAMPLE_K):
    """The tables of the module docstring from flat trajectory arrays (NumPy):
    kp_off (n_img+1), kp_xy (n_kept,2), pair_key (n_pairs) = src_image * n_img + tgt_image in ascending key order,
    pair_off (n_pairs+1), pair_first (n_pairs) = position of the pair's first match in the loop order, rows (n_matches,2)."""
    n_traj = len(off) - 1
    if len(frames) and (int(np.min(frames)) < 0 or int(np.max(frames)) >= n_img):
        # the reference indexes image_names[frame] (matches_from_flow.py:79) and raises; the device path returns PSFM_ERR_ARG
        raise IndexError("traj_to_matches: frame ids span [%d, %d], %d images" % (int(np.min(frames)), int(np.max(frames)), n_img))
    owner = np.repeat(np.arange(n_traj), np.diff(off))
    keep = ~labels if remove_dynamic else np.ones(len(frames), bool)     # :71-74
    frames, xy, owner = frames[keep], xy[keep], owner[keep]
    n_pts = len(frames)
    cnt = np.bincount(owner, minlength=n_traj).astype(np.int64)          # kept points per trajectory
    toff = np.zeros(n_traj + 1, np.int64)
    np.cumsum(cnt, out=toff[1:])

    # keypoint index = running count per image in trajectory order (:78-81): one stable sort by frame
    order = np.argsort(frames, kind="stable")
    fsorted = frames[order]
    kp_off = np.searchsorted(fsorted, np.arange(n_img + 1)).astype(np.int64)
    kp_ind = np.empty(n_pts, np.int64)
    kp_ind[order] = np.arange(n_pts) - kp_off[fsorted]
    kp_xy = xy[order]

    # matches (:83-101)
    n_of = cnt[owner]
    j_loc = np.arange(n_pts) - toff[owner]
    reps = np.where(n_of <= sample_k, n_of, sample_k)
    src = np.repeat(np.arange(n_pts), reps)
    r = np.arange(len(src)) - np.repeat(np.cumsum(reps) - reps, reps)     # 0..reps-1 inside each source point
    n_src = n_of[src]
    stride = np.where(n_src <= sample_k, 1, n_src // sample_k)
    tgt_loc = r * stride
    ok = tgt_loc != j_loc[src]
    src, tgt = src[ok], (toff[owner[src]] + tgt_loc)[ok]
    key = frames[src] * n_img + frames[tgt]
    po = np.argsort(key, kind="stable")             # inside a pair the loop order (trajectory, j, k) survives
    pk = key[po]
    rows = np.stack([kp_ind[src[po]], kp_ind[tgt[po]]], 1).astype(np.int32)
    bounds = np.flatnonzero(np.r_[True, pk[1:] != pk[:-1], True]) if len(pk) else np.zeros(1, np.int64)
    pair_key = pk[bounds[:-1]] if len(pk) else np.zeros(0, np.int64)
    pair_first = po[bounds[:-1]] if len(pk) else np.zeros(0, np.int64)
    return kp_off, kp_xy, pair_key.astype(np.int64), bounds.astype(np.int64), pair_first.astype(np.int64), rows


def assemble(image_names, tables, match_list_file, as_arrays=False):
    """Tables -> {image name: ImageMatches} (:103-108) + the pair list file (:110-117)."""
    kp_off, kp_xy, pair_key, pair_off, pair_first, rows = tables
    n_img = len(image_names)
    datas = [ImageMatches(i, kp_xy[kp_off[i]:kp_off[i + 1]] if as_arrays else kp_xy[kp_off[i]:kp_off[i + 1]].tolist())
             for i in range(n_img)]
    for g in np.argsort(pair_first, kind="stable"):       # dict order of the reference = order of first use
        si, ti = divmod(int(pair_key[g]), n_img)
        block = rows[pair_off[g]:pair_off[g + 1]]
        datas[si].match_pairs[image_names[si] + "-" + image_names[ti]] = block if as_arrays else block.tolist()
    out = {name: datas[i] for i, name in enumerate(image_names)}
    with open(match_list_file, "w") as fp:
        for data in out.values():
            for pair in data.match_pairs:
                a, b = pair.split("-")
                fp.write(a + " " + b + "\n")
    return out


def traj_to_matches(img_dir, traj_dir, match_list_file, remove_dynamic=True, sample_k=SAMPLE_K, as_arrays=False):
    """sfm/matches_from_flow.py:51-118, same arguments.  as_arrays=True keeps `.keypoints` / `.match_pairs[...]` as
    (n,2) ndarrays instead of nested lists -- the only consumer (sfm/import_feature_matches.py:82,96) wraps them in
    np.array() anyway, and building ~1e7 two-element lists is what dominates the list form."""
    from point_trajectory.trajectory import load_track_npy
    trajectories = load_track_npy(os.path.join(traj_dir, "track.npy"))      # (either layout; this package's files at array speed)
    image_names = sorted(os.listdir(img_dir))
    off, frames, xy, labels = _flatten(trajectories)
    tables = match_tables_host(off, frames, xy, labels, len(image_names), remove_dynamic, sample_k)
    return assemble(image_names, tables, match_list_file, as_arrays)


def match_tables_device(ctx, n_img, traj_min_len=3, sample_k=SAMPLE_K, labels=None):
    """The same tables from the result the last psfm_track / psfm_connect of `ctx` left in HBM: the saved set
    (length >= traj_min_len, psfm_result_filter) -> psfm_traj_to_matches -> one copy of the finished tables.
    labels: optional (n_points,) uint8 device tensor over the saved set's points (1 = dynamic, dropped)."""
    import ctypes
    from point_trajectory import _hip
    L = _hip.lib()
    sp = _hip.current_stream_ptr(ctx.device)
    k, npt = ctypes.c_int64(0), ctypes.c_int64(0)
    _hip.check(L.psfm_result_filter(ctx.handle, int(traj_min_len), ctypes.byref(k), ctypes.byref(npt), sp))
    n_kp, n_m, n_p = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _hip.check(L.psfm_traj_to_matches(ctx.handle, int(n_img), int(sample_k), _hip.ptr(labels), ctypes.byref(n_kp),
                                      ctypes.byref(n_m), ctypes.byref(n_p), sp))
    n_kp, n_m, n_p = int(n_kp.value), int(n_m.value), int(n_p.value)
    kp_off = np.zeros(n_img + 1, np.int64)
    kp_xy = np.empty((n_kp, 2), np.float64)
    pair_key = np.empty(n_p, np.int64)
    pair_off = np.zeros(n_p + 1, np.int64)
    pair_first = np.empty(n_p, np.int64)
    rows = np.empty((n_m, 2), np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _hip.check(L.psfm_matches_copy(ctx.handle, vp(kp_off), vp(kp_xy), vp(pair_key), vp(pair_off), vp(pair_first), vp(rows), sp))
    return kp_off, kp_xy, pair_key, pair_off, pair_first, rows


def traj_to_matches_device(ctx, image_names, match_list_file, traj_min_len=3, sample_k=SAMPLE_K, labels=None, as_arrays=True):
    """traj_to_matches without the track.npy round trip (e.g. --assume_static, where the trajectory stage feeds SfM
    directly, run_particlesfm.py:114): tables from HBM, assembled like the host path."""
    tables = match_tables_device(ctx, len(image_names), traj_min_len, sample_k, labels)
    return assemble(list(image_names), tables, match_list_file, as_arrays)
