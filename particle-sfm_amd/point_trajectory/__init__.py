"""point_trajectory -- MI355X-native drop-in for bytedance/particle-sfm's `point_trajectory` package.

Put `particle-sfm_amd/` on sys.path ahead of the reference checkout and the pipeline driver's
`from point_trajectory import main_connect_point_trajectories` (run_particlesfm.py:21) resolves here.
"""
from .main_connect_point_trajectories import main_connect_point_trajectories  # noqa: F401
