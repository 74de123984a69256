"""Drop-in proof on the consumer side: track.npy files written by this implementation ON THE GPU
(scripts/make_track_fixture.py -> tests/golden/track_gpu_60x80*.npy) are fed to the REFERENCE's own, unmodified
sfm/matches_from_flow.py::traj_to_matches.  Needs the reference tree, so it only runs in the build container
(skipped elsewhere); the oracle provides the expected trajectories."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from _common import GOLDEN
import psfm_synth

REF = os.environ.get("PSFM_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sfm")), reason="reference tree not present")


def _load_reference_consumer():
    spec = importlib.util.spec_from_file_location("psfm_ref_matches_from_flow", os.path.join(REF, "sfm", "matches_from_flow.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.tqdm = lambda it, *a, **k: it
    return mod


@pytest.mark.parametrize("fname", ["track_gpu_60x80.npy", "track_gpu_60x80_legacy.npy"])
def test_reference_traj_to_matches_consumes_our_track_npy(tmp_path, fname):
    from oracle import oracle as orc
    import point_trajectory  # noqa: F401  (makes point_trajectory.optimize.build.particlesfm resolvable for the unpickler)
    T, H, W, r = 7, 60, 80, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=77, sigma=0.1, n_occluders=1, stride2=True)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    O = orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
    keep = [i for i in range(O.n_traj) if O.length[i] >= 3]

    traj_dir = tmp_path / "trajectories"
    img_dir = tmp_path / "images"
    traj_dir.mkdir(); img_dir.mkdir()
    for i in range(T):
        (img_dir / ("%05d.png" % i)).write_bytes(b"")
    (traj_dir / "track.npy").write_bytes(open(os.path.join(GOLDEN, fname), "rb").read())

    ref = _load_reference_consumer()
    data = ref.traj_to_matches(str(img_dir), str(traj_dir), str(tmp_path / "pairs.txt"), remove_dynamic=True)
    # every kept trajectory point became one keypoint of its frame (sfm/matches_from_flow.py:67-86)
    n_kp = sum(len(v.keypoints) for v in data.values())
    assert n_kp == int(sum(O.length[i] for i in keep))
    # keypoints of frame 0 are the time-0 points of the kept trajectories born at 0, in id order
    exp0 = [O.xy[O.off[i]] for i in keep if O.birth[i] == 0]
    kp0 = np.array(data["00000.png"].keypoints)
    assert kp0.shape == (len(exp0), 2) and np.abs(kp0 - np.array(exp0)).max() <= 1e-4
    assert os.path.getsize(str(tmp_path / "pairs.txt")) > 0
    # and the object the consumers unpickle is this package's class at the reference's dotted path
    ts = np.load(str(traj_dir / "track.npy"), allow_pickle=True).item()
    assert type(ts).__module__ == "point_trajectory.optimize.build.particlesfm"
    assert sorted(ts.as_dict()) == keep
