"""Mirror of point_trajectory/track.py (reference :24-50)."""
from .trajectory import run_track


def track(flows, occ_maps, sample_ratio):
    """Sequentially track point trajectories (track.py:24-50).

    flows: list of (H,W,2) float arrays (or an (n,H,W,2) array / device tensor); occ_maps: list of (H,W) bool.
    Returns the trajectories in full_trajs order as a list-like of Trajectory (see TrajectoryList).
    The whole frame loop runs on the MI355X (csrc/psfm_track.hip)."""
    return run_track(flows, occ_maps, None, None, sample_ratio)
