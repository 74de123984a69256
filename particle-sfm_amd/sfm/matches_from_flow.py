"""Array-speed drop-in for the reference's sfm/matches_from_flow.py (a "next" row: the consumer
that turns trajectories into per-image keypoints and sampled pair matches for the COLMAP database).

Same API and the same output objects as the reference (`imageMatchData` with `.keypoints` and `.match_pairs`,
`traj_to_matches(img_dir, traj_dir, match_list_file, remove_dynamic=True)`, reference :21-118), but the per-trajectory /
per-point Python loops (:66-101, O(sum(len) * K)) are replaced by NumPy index arithmetic on the CSR-backed
TrajectorySet: keypoint indices are a grouped running count, match lists are built from flat (source, target)
index arrays with one stable sort by image pair.  Outputs are identical element for element (tested against the
reference's own function in the build container).
"""
import os

import numpy as np


class imageMatchData():
    """sfm/matches_from_flow.py:21-49"""

    def __init__(self, image_id):
        self.image_id = image_id
        self.keypoints = []
        self.match_pairs = {}

    def insert_keypoint(self, kp1):
        self.keypoints.append(kp1)

    def insert_match(self, tgt_img_id, kp_ind1, kp_ind2):
        match_key = str(self.image_id) + '-' + str(tgt_img_id)
        if match_key in self.match_pairs.keys():
            self.match_pairs[match_key].append([kp_ind1, kp_ind2])
        else:
            self.match_pairs[match_key] = [[kp_ind1, kp_ind2]]

    def rename_matches(self, image_names):
        old_keys = np.copy(list(self.match_pairs.keys()))
        for key in old_keys:
            self_id, tgt_img_id = key.split('-')
            self_id, tgt_img_id = int(self_id), int(tgt_img_id)
            assert self_id == self.image_id
            new_key = image_names[self_id] + '-' + image_names[tgt_img_id]
            self.match_pairs[new_key] = self.match_pairs.pop(key)


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


def traj_to_matches(img_dir, traj_dir, match_list_file, remove_dynamic=True, sample_k=20, as_arrays=False):
    """sfm/matches_from_flow.py:51-118.  as_arrays=True keeps `.keypoints` / `.match_pairs[...]` as (n,2) ndarrays instead of
    nested Python lists -- the only consumer (sfm/import_feature_matches.py:82,96) wraps them in np.array() anyway, and
    building ~1e7 two-element lists is what dominates the run time of the list form."""
    trajectories = np.load(os.path.join(traj_dir, "track.npy"), allow_pickle=True).item()
    image_names = sorted(os.listdir(img_dir))
    n_img = len(image_names)
    image_datas = [imageMatchData(image_id=i) for i in range(n_img)]

    off, frames, xy, labels = _flatten(trajectories)
    n_traj = len(off) - 1
    owner = np.repeat(np.arange(n_traj), np.diff(off))
    keep = ~labels if remove_dynamic else np.ones(len(frames), bool)     # :71-74
    frames, xy, owner = frames[keep], xy[keep], owner[keep]
    n_pts = len(frames)
    # trajectory lengths / offsets after dropping dynamic points
    cnt = np.bincount(owner, minlength=n_traj).astype(np.int64)
    toff = np.zeros(n_traj + 1, np.int64)
    np.cumsum(cnt, out=toff[1:])

    # keypoint index of every point inside its image: running count per image in trajectory order (:78-81)
    order = np.argsort(frames, kind="stable")
    fsorted = frames[order]
    first = np.searchsorted(fsorted, np.arange(n_img))
    kp_ind = np.empty(n_pts, np.int64)
    kp_ind[order] = np.arange(n_pts) - first[fsorted]
    for i in range(n_img):
        sel = order[first[i]:(first[i + 1] if i + 1 < n_img else n_pts)]
        image_datas[i].keypoints = xy[sel] if as_arrays else xy[sel].tolist()

    # matches (:83-101): point j of a trajectory of length n pairs with every other point (n <= K) or with the K
    # points k*stride, stride = n // K (skipping itself)
    n_of = cnt[owner]
    j_loc = np.arange(n_pts) - toff[owner]
    small = n_of <= sample_k
    reps = np.where(small, n_of, sample_k)
    src = np.repeat(np.arange(n_pts), reps)
    r = np.arange(len(src)) - np.repeat(np.cumsum(reps) - reps, reps)     # 0..reps-1 inside each source point
    n_src = n_of[src]
    stride = np.where(n_src <= sample_k, 1, n_src // sample_k)
    tgt_loc = r * stride
    ok = tgt_loc != j_loc[src]
    src, tgt = src[ok], (toff[owner[src]] + tgt_loc)[ok]
    pair_key = frames[src] * n_img + frames[tgt]
    po = np.argsort(pair_key, kind="stable")        # per pair: trajectory order, then j, then k -- as the loops append
    pk = pair_key[po]
    rows = np.stack([kp_ind[src[po]], kp_ind[tgt[po]]], 1)
    bounds = np.flatnonzero(np.r_[True, pk[1:] != pk[:-1], True]) if len(pk) else np.array([0])
    # dict insertion order of the reference = order of first appearance of each pair in the loops
    first_seen = np.full(len(bounds) - 1, 0, np.int64)
    if len(pk):
        first_seen = po[bounds[:-1]]
    for g in np.argsort(first_seen, kind="stable"):
        a, b = bounds[g], bounds[g + 1]
        si, ti = divmod(int(pk[a]), n_img)
        image_datas[si].match_pairs["%d-%d" % (si, ti)] = rows[a:b] if as_arrays else rows[a:b].tolist()

    colmap_datas = {}
    for i, img_name in enumerate(image_names):      # :103-108
        image_datas[i].rename_matches(image_names)
        colmap_datas[img_name] = image_datas[i]
    with open(match_list_file, 'w') as pair_txt:    # :110-117
        for img_name, data in colmap_datas.items():
            for m in data.match_pairs.keys():
                name0, name1 = m.split('-')
                pair_txt.write(name0 + ' ' + name1 + '\n')
    return colmap_datas
