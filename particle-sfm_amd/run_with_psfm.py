#!/usr/bin/env python3
"""Run a script of an unmodified bytedance/particle-sfm checkout with this package's `point_trajectory` in place of
the checkout's own:

    python /path/to/particle-sfm_amd/run_with_psfm.py [--ref /path/to/particle-sfm] run_particlesfm.py <its arguments>

Why a launcher: `python run_particlesfm.py` puts the checkout's directory at sys.path[0], AHEAD of PYTHONPATH, so
`from point_trajectory import main_connect_point_trajectories` (run_particlesfm.py:21) would keep resolving to the
checkout's package whatever PYTHONPATH says.  Here the order is made explicit: this directory first (it provides
`point_trajectory` only -- the helpers for the other stages are named psfm_sfm / psfm_motion_seg so that the
checkout's `sfm` and `motion_seg` packages, imported at run_particlesfm.py:61,78, stay the checkout's), then the
checkout.  The script then runs as __main__ exactly as if started from the checkout.
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def configure_paths(ref_root):
    """sys.path = [this package dir, the checkout, ...rest] (duplicates removed)."""
    ref_root = os.path.abspath(ref_root)
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, ref_root)]
    sys.path[:] = [HERE, ref_root] + rest
    for name in [m for m in sys.modules if m == "point_trajectory" or m.startswith("point_trajectory.")]:
        del sys.modules[name]     # anything imported before the switch must not shadow it
    return sys.path


def main(argv):
    ref = os.environ.get("PSFM_REFERENCE_ROOT", os.getcwd())
    if len(argv) >= 2 and argv[0] == "--ref":
        ref, argv = argv[1], argv[2:]
    if not argv:
        sys.exit(__doc__)
    script = argv[0] if os.path.isabs(argv[0]) else os.path.join(ref, argv[0])
    configure_paths(ref)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main(sys.argv[1:])
