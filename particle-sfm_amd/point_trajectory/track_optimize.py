"""Mirror of point_trajectory/track_optimize.py (reference :24-53)."""
from .trajectory import run_track


def track_optimize(flows, flows_f2, occ_maps, occ_maps_s2, sample_ratio):
    """Sequentially track and optimize point trajectories (track_optimize.py:24-53): the chain step of
    track() plus, from the third frame on, the path-consistency solve over every track with a full
    3-deep buffer (trajectory.py:161-194 -> csrc/psfm_solver.hip)."""
    return run_track(flows, occ_maps, flows_f2, occ_maps_s2, sample_ratio)
