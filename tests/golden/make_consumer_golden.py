#!/usr/bin/env python3
"""Golden vectors for the consumers of track.npy (SURVEY f-3, f-4), produced by the REFERENCE's own functions, imported
unmodified from /root/reference through oracle/ref_shim.load_consumers():

  matches_<case>.npz   sfm/matches_from_flow.py:51-118  traj_to_matches  -> per-image keypoints, per-pair match rows in the
                       reference's dict order
  windows_<case>.npz   motion_seg/load_cut_seq.py:25-89 load_cut_seq (window cutting, min_length 3) +
                       core/dataset/data_utils.py:74-89 resize_point_traj / normalize_point_traj -> raw / normalised window
                       tensors, masks, frame and trajectory ids per window.  TrajectorySet::sample_inside_window is C++
                       (trajectory_base.cpp:127-185, unbuildable here): the shim's restatement supplies it.

The trajectories come from the CPU oracle's track() on a seeded psfm_synth sequence (bit-exactly what the HIP path
produces -- tests/test_gpu_parity.py), saved set = length >= 3.  Run in the build container:
    python tests/golden/make_consumer_golden.py
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import psfm_synth                     # noqa: E402
from oracle import oracle as orc     # noqa: E402
from oracle import ref_shim          # noqa: E402

# name, T, H, W, ratio, seed, sigma, occluders, flow amplitude (px), fraction of dynamic points (labels == 1)
MATCH_CASES = [("matches_40x56_t12", 12, 40, 56, 2, 301, 0.3, 2, 3.0, 0.0),
               # slow flow: trajectories longer than K = 20 points (the strided-sampling branch, :92-101) + dynamic labels
               ("matches_24x32_t27_dyn", 27, 24, 32, 2, 302, 0.05, 1, 0.5, 0.12)]
WINDOW_CASES = [("windows_48x64_t23", 23, 48, 64, 2, 311, 0.3, 2, 3.0, 10, (30, 50))]


def input_hash(d):
    h = hashlib.sha256()
    for k in sorted(d):
        for a in d[k]:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def saved_set(T, H, W, r, seed, sigma, nocc, amp, min_len=3):
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, amp=amp, sigma=sigma, n_occluders=nocc, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = orc.track(d["flows_f"], occ, r)
    keep = np.flatnonzero(R.length >= min_len)
    return d, R, keep


def dynamic_labels(n_points, frac, seed):
    return (np.random.default_rng(seed).uniform(size=n_points) < frac) if frac > 0 else np.zeros(n_points, bool)


def main():
    ref = ref_shim.load_consumers()
    for name, T, H, W, r, seed, sigma, nocc, amp, frac in MATCH_CASES:
        d, R, keep = saved_set(T, H, W, r, seed, sigma, nocc, amp)
        n_pts = int(R.length[keep].sum())
        labels = dynamic_labels(n_pts, frac, seed + 1000)          # over the saved set's points, in id order
        trajs, o = {}, 0
        for i in keep:
            b, xy = R.traj(int(i))
            n = len(xy)
            trajs[int(i)] = {"frame_ids": list(range(int(b), int(b) + n)), "locations": [p.copy() for p in xy],
                             "labels": labels[o:o + n].tolist()}
            o += n
        with tempfile.TemporaryDirectory() as tmp:
            img_dir, traj_dir = os.path.join(tmp, "images"), os.path.join(tmp, "traj")
            os.makedirs(img_dir); os.makedirs(traj_dir)
            names = ["%05d.png" % i for i in range(T)]
            for nme in names:
                open(os.path.join(img_dir, nme), "w").close()
            np.save(os.path.join(traj_dir, "track.npy"), trajs, allow_pickle=True)   # the plain-dict form motion_seg writes
            data = ref.traj_to_matches(img_dir, traj_dir, os.path.join(tmp, "pairs.txt"), remove_dynamic=True)
            pair_lines = open(os.path.join(tmp, "pairs.txt")).read().split("\n")
        kp_off = np.zeros(T + 1, np.int64)
        kp_xy, p_src, p_tgt, p_off, rows = [], [], [], [0], []
        for i, nme in enumerate(names):
            kp = np.asarray(data[nme].keypoints, np.float64).reshape(-1, 2)
            kp_off[i + 1] = kp_off[i] + len(kp)
            kp_xy.append(kp)
            for key, m in data[nme].match_pairs.items():          # dict order = order of first use
                a, b = key.split("-")
                p_src.append(names.index(a)); p_tgt.append(names.index(b))
                rows.append(np.asarray(m, np.int32).reshape(-1, 2))
                p_off.append(p_off[-1] + len(m))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, sigma=sigma, n_occluders=nocc,
                            amp=amp, dyn_frac=frac, input_hash=input_hash(d), n_saved=len(keep), n_points=n_pts,
                            labels=np.packbits(labels), kp_off=kp_off, kp_xy=np.concatenate(kp_xy, 0),
                            pair_src=np.asarray(p_src, np.int32), pair_tgt=np.asarray(p_tgt, np.int32),
                            pair_off=np.asarray(p_off, np.int64), rows=np.concatenate(rows, 0),
                            pair_file_hash=hashlib.sha256("\n".join(pair_lines).encode()).hexdigest())
        print(name, len(keep), "trajectories,", n_pts, "points,", int(labels.sum()), "dynamic,", int(kp_off[-1]), "keypoints,",
              len(p_src), "pairs,", int(p_off[-1]), "matches; longest", int(R.length.max()))

    for name, T, H, W, r, seed, sigma, nocc, amp, window, input_size in WINDOW_CASES:
        d, R, keep = saved_set(T, H, W, r, seed, sigma, nocc, amp)
        ts = ref_shim.TrajectorySet({int(i): ref_shim.Trajectory({"frame_ids": list(range(int(R.birth[i]), int(R.birth[i]) + int(R.length[i]))),
                                                                   "locations": list(R.traj(int(i))[1]), "labels": [False] * int(R.length[i])})
                                     for i in keep})
        ref.cv2.imread = lambda nme, flag=1, H=H, W=W: np.zeros((H, W, 3), np.uint8) if flag != -1 else np.zeros((H, W), np.float64)
        out = {}
        with tempfile.TemporaryDirectory() as tmp:
            img_dir, depth_dir, traj_dir = (os.path.join(tmp, x) for x in ("images", "depths", "traj"))
            for p in (img_dir, depth_dir, traj_dir):
                os.makedirs(p)
            for i in range(T):
                open(os.path.join(img_dir, "%05d.png" % i), "w").close()
                open(os.path.join(depth_dir, "%05d.png" % i), "w").close()
            np.save(os.path.join(traj_dir, "track.npy"), ts, allow_pickle=True)
            for tag, win in (("w", window), ("full", T + 5)):      # cut into windows / one window over everything (:50-58)
                _, _, raw_b, nor_b, mask_b, time_b, idx_b = ref.load_cut_seq(img_dir, depth_dir, traj_dir, win, input_size, 10 ** 9)
                out[tag + "_n"] = len(raw_b)
                for w in range(len(raw_b)):
                    out["%s%d_raw" % (tag, w)] = np.asarray(raw_b[w], np.float64)
                    out["%s%d_nor" % (tag, w)] = np.asarray(nor_b[w], np.float64)
                    out["%s%d_mask" % (tag, w)] = np.asarray(mask_b[w], np.float64)
                    out["%s%d_time" % (tag, w)] = np.asarray(time_b[w], np.int64)
                    out["%s%d_ids" % (tag, w)] = np.asarray(idx_b[w], np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), T=T, H=H, W=W, ratio=r, seed=seed, sigma=sigma, n_occluders=nocc,
                            amp=amp, window=window, input_size=np.asarray(input_size), input_hash=input_hash(d), n_saved=len(keep), **out)
        print(name, len(keep), "trajectories,", out["w_n"], "windows of", window, "frames:", [int(out["w%d_ids" % w].shape[0]) for w in range(out["w_n"])],
              "tracks; full window", int(out["full0_ids"].shape[0]))


if __name__ == "__main__":
    main()
