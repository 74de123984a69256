"""The trajectory side of motion_seg/load_cut_seq.py:60-89 (reference) on the GPU.

The reference reloads track.npy, builds the inverted index and, per window, calls
TrajectorySet.sample_inside_window, then resizes / normalises the padded coordinate arrays
(motion_seg/core/dataset/data_utils.py:74-89).  `cut_trajectory_windows` produces the same five lists --
raw_traj_batchs, traj_batchs, mask_batchs, time_idx_batchs, sample_idx_batchs -- directly from the result that
psfm_track / psfm_connect left in HBM (one psfm_window_sample call per window), as torch tensors on the device or
NumPy arrays.  Images and depth maps (cv2 I/O in the reference) are not part of the hot path and stay with the caller.
"""
import ctypes

import numpy as np

from point_trajectory import _hip


def window_ranges(length, window):
    """The frame ranges load_cut_seq.py:50-73 cuts a sequence into: [(first, n_frames), ...]."""
    length, window = int(length), int(window)
    if window >= length:
        return [(0, length)]
    num = int(np.ceil(1.0 * length / window))
    return [(i * window, window) if i != num - 1 else (length - window, window) for i in range(num)]


def sample_window_device(ctx, frame0, n_frames, raw_hw, input_size, traj_max_num=100000, min_length=3, traj_min_len=3,
                         seed=0, normalise=True):
    """One window: (ids (K,) i32, raw (K,L,2) f64, normalised (K,L,2) f64 | None, mask_absent (K,L,1) f64) device tensors."""
    import torch
    L = _hip.lib()
    k = ctypes.c_int64(0)
    sp = _hip.current_stream_ptr(ctx.device)
    args = (int(frame0), int(n_frames), int(traj_min_len), int(min_length), int(traj_max_num), int(seed),
            int(raw_hw[0]), int(raw_hw[1]), int(input_size[0]), int(input_size[1]))
    _hip.check(L.psfm_window_sample(ctx.handle, *args, 0, None, None, None, None, ctypes.byref(k), sp))   # count
    K = int(k.value)
    dev = torch.device("cuda", ctx.device)
    ids = torch.empty((K,), dtype=torch.int32, device=dev)
    raw = torch.empty((K, int(n_frames), 2), dtype=torch.float64, device=dev)
    nor = torch.empty((K, int(n_frames), 2), dtype=torch.float64, device=dev) if normalise else None
    mask = torch.empty((K, int(n_frames), 1), dtype=torch.float64, device=dev)
    if K:
        _hip.check(L.psfm_window_sample(ctx.handle, *args, K, _hip.ptr(ids), _hip.ptr(raw), _hip.ptr(nor), _hip.ptr(mask),
                                        ctypes.byref(k), sp))
        assert int(k.value) == K
    return ids, raw, nor, mask


def cut_trajectory_windows(length, window, raw_hw, input_size, traj_max_num, seed=0, traj_min_len=3, as_numpy=False, ctx=None):
    """load_cut_seq.py:46-89 for the trajectories: five lists (one entry per window) built from the device-resident
    result of the calling thread's psfm context (run psfm_track / psfm_connect / main_connect first)."""
    ctx = ctx or _hip.context()
    raw_b, nor_b, mask_b, time_b, idx_b = [], [], [], [], []
    for w, (f0, n) in enumerate(window_ranges(length, window)):
        ids, raw, nor, mask = sample_window_device(ctx, f0, n, raw_hw, input_size, traj_max_num, 3, traj_min_len, seed + w)
        if as_numpy:
            ids, raw, nor, mask = ids.cpu().numpy(), raw.cpu().numpy(), nor.cpu().numpy(), mask.cpu().numpy()
        raw_b.append(raw); nor_b.append(nor); mask_b.append(mask)
        time_b.append(np.arange(f0, f0 + n)); idx_b.append(ids)
    return raw_b, nor_b, mask_b, time_b, idx_b
