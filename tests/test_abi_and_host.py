"""CPU-side checks: the C-ABI library loads and exports every symbol include/psfm.h declares, fails loudly
without a GPU (no fallback), and the host-side mirror of the reference's containers behaves like the pybind
classes (optimize/src/bindings.cc, trajectory_base.cpp)."""
import ctypes
import io
import os
import pickle
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "psfm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psfm_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from point_trajectory import _hip
    L = _hip.lib()
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), "libpsfm_hip.so does not export %s" % n
    assert set(names) == set(_hip.EXPORTS)
    assert L.psfm_version() == 141


def test_no_cpu_fallback():
    """Without a HIP device every compute path raises; nothing silently routes to the oracle or the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from point_trajectory import _hip
    from point_trajectory.utils import flow_check
    from point_trajectory.track import track
    L = _hip.lib()
    assert L.psfm_device_count() == 0
    h = ctypes.c_void_p()
    assert L.psfm_ctx_create(0, ctypes.byref(h)) == _hip.PSFM_ERR_HIP
    assert b"no CPU fallback" in L.psfm_last_error()
    f = [np.zeros((8, 8, 2), np.float32)]
    with pytest.raises(RuntimeError):
        flow_check(f, f, 1.0)
    with pytest.raises(RuntimeError):
        track(f, [np.zeros((8, 8), bool)], 2)
    from point_trajectory.trajectory import run_connect_batch
    z = torch.zeros((3, 8, 8, 2))
    with pytest.raises(RuntimeError):
        run_connect_batch([(z, z, None, None), (z, z, None, None)], 1.0, 2)       # (a batch of sequences: no GPU, no result)
    # the product package never imports the oracle
    pkg = os.path.join(ROOT, "particle-sfm_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", ""), os.path.join(dp, fn)


def test_flo_roundtrip(tmp_path):
    from point_trajectory.utils import read_flo, write_flo, load_flows
    rng = np.random.default_rng(0)
    for i in range(3):
        write_flo(str(tmp_path / ("%05d.flo" % i)), rng.standard_normal((7, 11, 2)).astype(np.float32))
    fl = load_flows(str(tmp_path))
    assert len(fl) == 3 and fl[0].shape == (7, 11, 2) and fl[0].dtype == np.float32
    rng = np.random.default_rng(0)
    assert np.array_equal(fl[0], rng.standard_normal((7, 11, 2)).astype(np.float32))
    with pytest.raises(AssertionError):
        read_flo(str(tmp_path / "nope.flo"))


def _toy_list():
    from point_trajectory.trajectory import TrajectoryList
    birth = np.array([0, 0, 1, 2, 0], np.int32)
    length = np.array([1, 3, 4, 2, 6], np.int32)
    off = np.zeros(6, np.int64)
    off[1:] = np.cumsum(length)
    xy = np.arange(2 * off[-1], dtype=np.float64).reshape(-1, 2)
    return TrajectoryList(birth, length, off, xy)


def test_trajectory_list_behaves_like_the_reference_list():
    tl = _toy_list()
    assert len(tl) == 5
    t = tl[2]
    assert t.length() == 4 and t.times == [1, 2, 3, 4] and t.labels == [False] * 4
    assert all(isinstance(p, np.ndarray) and p.shape == (2,) for p in t.xys)
    assert np.array_equal(np.array(t.xys), tl.xy[4:8])
    assert np.array_equal(t.get_tail_location(), tl.xy[7])
    d = t.as_dict()
    assert set(d) == {"frame_ids", "locations", "labels"} and d["frame_ids"] == [1, 2, 3, 4]
    assert [x.length() for x in tl] == [1, 3, 4, 2, 6]
    # main_connect_point_trajectories.py:56-60: ids are list indices, short ones dropped
    ts = tl.to_trajectory_set(3)
    assert sorted(ts.trajs) == [1, 2, 4]


def test_trajectory_semantics_match_trajectory_base_cpp():
    from point_trajectory.optimize.build.particlesfm import Trajectory
    t = Trajectory(np.int64(3), np.array([4, 6]), buffer_size=3)       # trajectory.py:119 call form
    for k in range(4):
        t.extend(4 + k, [k, k])
    assert t.length() == 5 and len(t.buffer_xys) == 3 and len(t.xys) == 2
    assert np.array_equal(t.get_tail_location(), [3.0, 3.0])
    t.set_buffer_xy(1, [9.0, 9.0])
    with pytest.raises(RuntimeError):
        t.set_buffer_xy(3, [0, 0])
    t.clear_buffer()
    assert len(t.buffer_xys) == 0 and np.array_equal(np.array(t.xys)[3], [9.0, 9.0])
    assert t.times == [3, 4, 5, 6, 7]
    t2 = Trajectory([0, 1], [np.zeros(2), np.ones(2)])
    assert t2.labels == [False, False] and t2.length() == 2
    t3 = Trajectory(t2.as_dict())
    assert t3.times == [0, 1] and np.array_equal(np.array(t3.xys), np.array(t2.xys))
    with pytest.raises(RuntimeError):
        Trajectory().get_tail_location()


def test_track_npy_roundtrip_and_consumers(tmp_path):
    """np.save / np.load(allow_pickle).item() through the access patterns of the unmodified consumers
    (sfm/matches_from_flow.py:56-86, motion_seg/load_cut_seq.py:46-79)."""
    from point_trajectory.optimize.build import particlesfm
    tl = _toy_list()
    ts = tl.to_trajectory_set(3)
    fn = str(tmp_path / "track.npy")
    np.save(fn, ts)
    raw = open(fn, "rb").read()
    assert b"point_trajectory.optimize.build.particlesfm" in raw and b"TrajectorySet" in raw
    back = np.load(fn, allow_pickle=True).item()
    assert type(back).__name__ == "TrajectorySet" and sorted(back.trajs) == [1, 2, 4]
    d = back.as_dict()                                     # matches_from_flow.py:57-58
    for key in d:
        traj = d[key]
        loc, lab, fid = np.array(traj["locations"]), np.array(traj["labels"]), np.array(traj["frame_ids"])
        assert loc.shape == (len(fid), 2) and lab.shape[0] == len(fid)
    assert np.array_equal(np.array(d[4]["locations"]), tl.xy[10:16])
    # the legacy state layout (list of (2,) arrays, as written by the pybind module) loads too
    legacy = {7: {"frame_ids": [2, 3, 4], "locations": [np.array([1.0, 2.0])] * 3, "labels": [False, True, False]}}
    ts2 = particlesfm.TrajectorySet.__new__(particlesfm.TrajectorySet)
    ts2.__setstate__(legacy)
    assert ts2.trajs[7].labels == [False, True, False] and ts2.trajs[7].length() == 3
    ts3 = particlesfm.TrajectorySet(legacy)
    assert ts3.as_dict()[7]["frame_ids"] == [2, 3, 4]
    with pytest.raises(RuntimeError):
        ts3.insert(7, ts3.trajs[7])
    # sample_inside_window (trajectory_base.cpp:127-185)
    with pytest.raises(RuntimeError):
        back.sample_inside_window([0, 1, 2])
    back.build_invert_indexes()
    out = back.sample_inside_window([1, 2, 3, 4], min_length=3)
    X, Y = out["locations"]
    assert out["traj_ids"] == [2, 4] and X.shape == (2, 4) and out["masks"].dtype == np.int32
    assert out["masks"].tolist() == [[1, 1, 1, 1], [1, 1, 1, 1]]
    assert np.array_equal(X[0], tl.xy[4:8, 0]) and np.array_equal(Y[1], tl.xy[11:15, 1])
    out = back.sample_inside_window([0, 1, 2, 9], min_length=3)
    assert out["traj_ids"] == [1, 4] and out["masks"].tolist() == [[1, 1, 1, 0], [1, 1, 1, 0]]
    assert out["locations"][0][0, 3] == 0.0
    out = back.sample_inside_window([0, 1, 2, 3, 4, 5], min_length=1, max_num_tracks=2)
    assert len(out["traj_ids"]) == 2


def _reference_sample_inside_window(trajs, frame_ids, min_length):
    """Plain restatement of trajectory_base.cpp:115-185 (map-of-maps + loops), without the random shrink."""
    inv = {}
    for tid, (times, xy) in trajs.items():
        for i, f in enumerate(times):
            inv.setdefault(f, {})[tid] = i
    counter = {}
    for f in frame_ids:
        for tid in inv.get(f, {}):
            counter[tid] = counter.get(tid, 0) + 1
    ids = [t for t in sorted(counter) if counter[t] >= min_length]
    K, L = len(ids), len(frame_ids)
    X, Y, M = np.zeros((K, L)), np.zeros((K, L)), np.zeros((K, L), np.int32)
    for a, tid in enumerate(ids):
        for b, f in enumerate(frame_ids):
            if f in inv and tid in inv[f]:
                X[a, b], Y[a, b] = trajs[tid][1][inv[f][tid]]
                M[a, b] = 1
    return ids, X, Y, M


def test_trajectory_set_csr_vs_map_and_reference_semantics(tmp_path, monkeypatch):
    from point_trajectory.optimize.build import particlesfm
    from point_trajectory.trajectory import TrajectoryList
    rng = np.random.default_rng(3)
    n = 200
    birth = rng.integers(0, 20, n).astype(np.int32)
    length = rng.integers(1, 12, n).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(length)
    xy = rng.uniform(0, 100, (int(off[-1]), 2))
    tl = TrajectoryList(birth, length, off, xy)
    ts = tl.to_trajectory_set(3)                      # CSR-backed
    kept = [i for i in range(n) if length[i] >= 3]
    assert len(ts) == len(kept)
    ref = {i: (list(range(birth[i], birth[i] + length[i])), xy[off[i]:off[i + 1]]) for i in kept}
    ts.build_invert_indexes()
    for frames, ml in [([3, 4, 5, 6, 7], 3), (list(range(10, 20)), 3), ([0, 1], 1), ([50, 51, 52], 3), ([5, 5, 6], 3)]:
        out = ts.sample_inside_window(frames, min_length=ml)
        ids, X, Y, M = _reference_sample_inside_window(ref, frames, ml)
        assert out["traj_ids"] == ids
        assert np.array_equal(out["locations"][0], X) and np.array_equal(out["locations"][1], Y)
        assert np.array_equal(out["masks"], M)
    # default pickle state = the reference's layout (what the pybind module writes/reads); the compact CSR is opt-in
    monkeypatch.delenv("PSFM_TRACK_LAYOUT", raising=False)
    st = ts.__getstate__()
    assert sorted(st) == kept and set(st[kept[0]]) == {"frame_ids", "locations", "labels"}
    assert st[kept[0]]["frame_ids"] == ref[kept[0]][0]
    fn2 = str(tmp_path / "legacy.npy")
    np.save(fn2, ts)
    assert b"__psfm_csr__" not in open(fn2, "rb").read()
    back2 = np.load(fn2, allow_pickle=True).item()
    ts.pickle_layout = "csr"
    fn = str(tmp_path / "track.npy")
    np.save(fn, ts)
    assert b"__psfm_csr__" in open(fn, "rb").read()
    back = np.load(fn, allow_pickle=True).item()
    assert sorted(back.trajs) == kept and np.array_equal(np.array(back.trajs[kept[0]].xys), ref[kept[0]][1])
    ts.pickle_layout = None
    d1, d2 = back.as_dict(), back2.as_dict()
    assert sorted(d1) == sorted(d2)
    for k in kept[:20]:
        assert d1[k]["frame_ids"] == d2[k]["frame_ids"]
        assert np.array_equal(np.array(d1[k]["locations"]), np.array(d2[k]["locations"]))
    # map-backed set built the reference way (main_connect_point_trajectories.py:56-61) behaves the same
    ts_map = particlesfm.TrajectorySet({i: tl[i] for i in kept})
    ts_map.build_invert_indexes()
    o1 = ts.sample_inside_window([3, 4, 5, 6, 7])
    o2 = ts_map.sample_inside_window([3, 4, 5, 6, 7])
    assert o1["traj_ids"] == o2["traj_ids"] and np.array_equal(o1["locations"][0], o2["locations"][0])


def test_fast_division_and_threshold_shortcuts_are_exact():
    """The HIP sampler divides by (size-1)/2 through a refined reciprocal + two fma corrections, and flow_check compares
    the squared error against a pre-squared threshold: oracle/test_fastdiv.c enumerates ~5e7 operands (every
    W <= 8192, reciprocal off by up to 1 ulp) and the floats around thres^2 against the true division / sqrtf."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-s", "-C", odir, "build/test_fastdiv"], check=True)
    r = subprocess.run([os.path.join(odir, "build", "test_fastdiv"), "2000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 counter-examples" in r.stdout


def test_window_ranges_and_track_npy_writer(tmp_path):
    """Host logic that needs no GPU: the window cut of load_cut_seq.py:50-73 and the protocol-5 track.npy writer (same
    container as np.save: consumers np.load(..., allow_pickle=True).item())."""
    from psfm_motion_seg.load_cut_seq import window_ranges
    from point_trajectory.trajectory import save_track_npy
    from point_trajectory.optimize.build.particlesfm import TrajectorySet
    assert window_ranges(23, 10) == [(0, 10), (10, 10), (13, 10)]
    assert window_ranges(20, 10) == [(0, 10), (10, 10)]
    assert window_ranges(7, 10) == [(0, 7)] and window_ranges(10, 10) == [(0, 10)]
    ids = np.array([0, 3, 4], np.int64)
    birth = np.array([0, 1, 2], np.int32)
    length = np.array([3, 4, 3], np.int32)
    off = np.array([0, 3, 7, 10], np.int64)
    xy = np.arange(20, dtype=np.float64).reshape(10, 2)
    ts = TrajectorySet._from_csr(ids, birth, length, off, xy)
    save_track_npy(str(tmp_path / "track.npy"), ts, layout="csr")
    save_track_npy(str(tmp_path / "ref.npy"), ts)                       # default: the reference's state layout
    assert b"__psfm_csr__" in open(str(tmp_path / "track.npy"), "rb").read()
    assert b"__psfm_csr__" not in open(str(tmp_path / "ref.npy"), "rb").read()
    a = np.load(str(tmp_path / "track.npy"), allow_pickle=True).item()
    b = np.load(str(tmp_path / "ref.npy"), allow_pickle=True).item()
    assert a.as_dict().keys() == b.as_dict().keys() == {0, 3, 4}
    assert list(a.as_dict()[3]["frame_ids"]) == [1, 2, 3, 4]
    assert np.array_equal(np.asarray(a.as_dict()[3]["locations"]), np.asarray(b.as_dict()[3]["locations"]))
    a.build_invert_indexes()
    out = a.sample_inside_window([2, 3, 4], min_length=3)
    assert out["traj_ids"] == [3, 4]


def test_default_track_npy_loads_with_a_pybind_module_that_only_knows_the_reference_contract(tmp_path):
    """bindings.cc:64-71: the reference's TrajectorySet unpickles from std::map<int, py::dict> and Trajectory from the
    three casts of trajectory_base.cpp:39-46.  tests/pybind_standin/ implements exactly that contract as a REAL pybind11
    module (g++ + the pip pybind11 headers); it is installed at point_trajectory/optimize/build/ in a scratch tree without
    this package on the path, and must load the DEFAULT output of save_track_npy / np.save."""
    import subprocess
    import sysconfig
    import textwrap
    pybind11 = pytest.importorskip("pybind11")
    from point_trajectory.trajectory import save_track_npy, TrajectoryList
    rng = np.random.default_rng(5)
    n = 50
    birth = rng.integers(0, 9, n).astype(np.int32)
    length = rng.integers(1, 9, n).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(length)
    xy = rng.uniform(0, 100, (int(off[-1]), 2))
    ts = TrajectoryList(birth, length, off, xy).to_trajectory_set(3)     # CSR-backed, as the stage entry produces it
    save_track_npy(str(tmp_path / "track.npy"), ts)
    np.save(str(tmp_path / "track_np_save.npy"), ts)
    tree = tmp_path / "refpkg" / "point_trajectory" / "optimize" / "build"
    tree.mkdir(parents=True)
    for d in (tree, tree.parent, tree.parent.parent):
        (d / "__init__.py").write_text("")
    so = tree / ("particlesfm" + sysconfig.get_config_var("EXT_SUFFIX"))
    src = os.path.join(ROOT, "tests", "pybind_standin", "particlesfm_standin.cpp")
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"]]
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-std=c++17"] + inc + [src, "-o", str(so)], check=True, timeout=300)
    code = textwrap.dedent("""
        import sys, json
        import numpy as np
        sys.path.insert(0, %r)
        out = {}
        for name in ("track.npy", "track_np_save.npy"):
            ts = np.load(%r + "/" + name, allow_pickle=True).item()
            assert type(ts).__module__ == "point_trajectory.optimize.build.particlesfm", type(ts).__module__
            assert "pybind11" in str(type(type(ts))), type(type(ts))
            d = ts.as_dict()
            out[name] = {str(k): [v["frame_ids"], np.asarray(v["locations"]).tolist(), v["labels"]] for k, v in d.items()}
        print(json.dumps(out))
    """) % (str(tmp_path / "refpkg"), str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    got = json.loads(r.stdout.strip().splitlines()[-1])
    kept = [i for i in range(n) if length[i] >= 3]
    for name in ("track.npy", "track_np_save.npy"):
        g = got[name]
        assert sorted(int(k) for k in g) == kept
        for i in kept:
            fr, loc, lab = g[str(i)]
            assert fr == list(range(birth[i], birth[i] + length[i])) and lab == [False] * int(length[i])
            assert np.array_equal(np.array(loc), xy[off[i]:off[i + 1]])


@pytest.mark.parametrize("n", [0, 1, 37, 3000])
def test_reference_layout_is_streamed_from_the_csr(tmp_path, n):
    """save_track_npy writes the reference's pickle state (bindings.cc:64-71) as opcodes straight from the CSR arrays
    (point_trajectory/reference_pickle.py).  What unpickles must be exactly what the generic pickler produces from the same set:
    the plain-Python object graph {id: {"frame_ids": [int], "locations": (n,2) f64, "labels": [bool]}} -- read here with the
    class replaced by a recorder, so nothing of this package's loading code is involved -- and the set itself on this side
    (3000 entries: loaded back as CSR, array-speed; fewer: the generic map)."""
    import pickle
    from point_trajectory.optimize.build import particlesfm
    from point_trajectory.trajectory import save_track_npy
    from point_trajectory import reference_pickle
    rng = np.random.default_rng(n)
    length = rng.integers(1, 30, n).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(length)
    birth = rng.integers(0, 400, n).astype(np.int32)
    xy = rng.normal(size=(int(off[-1]), 2)) * 1e3
    ids = np.sort(rng.choice(10 * n + 5, n, replace=False)).astype(np.int64)
    ts = particlesfm.TrajectorySet._from_csr(ids, birth, length, off, xy)
    assert reference_pickle.can_stream(ts)
    save_track_npy(str(tmp_path / "fast.npy"), ts)
    generic = pickle.loads(pickle.dumps(particlesfm.TrajectorySet._from_csr(ids, birth, length, off, xy)))._legacy_state()

    class Recorder:                      # stands in for the class named in the stream: records the raw state
        def __setstate__(self, state):
            self.state = state
    real = particlesfm.TrajectorySet
    particlesfm.TrajectorySet = Recorder
    try:
        raw = np.load(str(tmp_path / "fast.npy"), allow_pickle=True).item().state
    finally:
        particlesfm.TrajectorySet = real
    assert type(raw) is dict and list(raw.keys()) == ids.tolist() == list(generic.keys())
    for j, k in enumerate(raw):
        v = raw[k]
        assert set(v) == {"frame_ids", "locations", "labels"}
        assert type(v["frame_ids"]) is list and v["frame_ids"] == list(range(birth[j], birth[j] + length[j])) == generic[k]["frame_ids"]
        assert type(v["labels"]) is list and v["labels"] == [False] * int(length[j]) == generic[k]["labels"]
        assert all(type(x) is int for x in v["frame_ids"][:3]) and all(x is False for x in v["labels"][:3])
        loc = np.asarray(v["locations"])
        assert loc.dtype == np.float64 and loc.shape == (length[j], 2) and np.array_equal(loc, xy[off[j]:off[j + 1]])
    if n:
        raw[ids[0]]["labels"].append(True)            # the label lists are independent objects (not one shared list)
        assert raw[ids[-1]]["labels"] == [False] * int(length[-1]) or n == 1
    back = np.load(str(tmp_path / "fast.npy"), allow_pickle=True).item()
    assert isinstance(back, real)
    if n >= 1024:
        assert back._csr is not None and back._map is None      # array-speed load of the reference layout
        assert np.array_equal(back._csr[0], ids) and np.array_equal(back._csr[1], birth) and np.array_equal(back._csr[2], length)
        assert np.array_equal(back._csr[4], xy)
    assert len(back.trajs) == n
    for j in (0, n // 2, n - 1) if n else ():
        t = back.trajs[int(ids[j])]
        assert t.length() == length[j] and np.array_equal(np.asarray(t.as_dict()["locations"]), xy[off[j]:off[j + 1]])
    # this package's own reader takes the footer behind the pickle: the same set without unpickling anything
    from point_trajectory.trajectory import load_track_npy
    fast = load_track_npy(str(tmp_path / "fast.npy"))
    assert isinstance(fast, real) and fast._csr is not None and fast._map is None
    for got, want in zip(fast._csr[:5], (ids, birth, length, off, xy)):
        assert np.array_equal(np.asarray(got), want)
    assert fast._csr[5] is None
    # ... and falls back to np.load for every other file (the compact layout, a file whose footer does not match)
    save_track_npy(str(tmp_path / "csr.npy"), particlesfm.TrajectorySet._from_csr(ids, birth, length, off, xy), layout="csr")
    assert len(load_track_npy(str(tmp_path / "csr.npy")).trajs) == n
    blob = open(str(tmp_path / "fast.npy"), "rb").read()
    open(str(tmp_path / "cut.npy"), "wb").write(blob[:-8] + b"XXXXXXXX")
    assert len(load_track_npy(str(tmp_path / "cut.npy")).trajs) == n
    # a set with labels (after motion segmentation) is outside the streaming form: the generic pickler writes it
    if n == 37:
        lab = rng.uniform(size=int(off[-1])) < 0.5
        tl = particlesfm.TrajectorySet._from_csr(ids, birth, length, off, xy, lab)
        assert not reference_pickle.can_stream(tl)
        save_track_npy(str(tmp_path / "lab.npy"), tl)
        bl = np.load(str(tmp_path / "lab.npy"), allow_pickle=True).item()
        assert bl.trajs[int(ids[3])].as_dict()["labels"] == lab[off[3]:off[4]].tolist()


def test_streaming_writer_declines_what_its_records_cannot_hold():
    """4-byte fields: a set with 2^31 or more points, or an id outside [0, 2^31), goes through the generic pickler."""
    from point_trajectory.optimize.build import particlesfm
    from point_trajectory import reference_pickle
    mk = lambda ids, off_last: particlesfm.TrajectorySet._from_csr(np.asarray(ids, np.int64), np.zeros(len(ids), np.int32),
                                                                  np.ones(len(ids), np.int32),
                                                                  np.concatenate([np.arange(len(ids)), [off_last]]).astype(np.int64),
                                                                  np.zeros((0, 2)))
    assert reference_pickle.can_stream(mk([0, 1, 2], 3))
    assert not reference_pickle.can_stream(mk([0, 1, 2], 1 << 31))
    assert not reference_pickle.can_stream(mk([0, 1, 1 << 31], 3))
    assert not reference_pickle.can_stream(mk([-1, 1, 2], 3))
    ts = mk([0, 1, 2], 3)
    _ = ts.trajs                                   # the map was materialised (and may have been edited): generic path
    assert not reference_pickle.can_stream(ts)
    assert not reference_pickle.can_stream({0: {}})


def test_package_imports_from_a_git_archive(tmp_path):
    """What `git archive HEAD` exports must be a working source tree: every module of the product package is tracked
    (an unanchored ignore pattern once hid point_trajectory/optimize/build/) and `import point_trajectory` works from it."""
    import shutil
    import subprocess
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    tar = tmp_path / "head.tar"
    r = subprocess.run(["git", "-C", ROOT, "archive", "--format=tar", "-o", str(tar), "HEAD"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("git archive failed: " + r.stderr[:200])
    out = tmp_path / "export"
    out.mkdir()
    subprocess.run(["tar", "-xf", str(tar), "-C", str(out)], check=True)
    for rel in ("particle-sfm_amd/point_trajectory/optimize/build/particlesfm.py", "particle-sfm_amd/point_trajectory/shard.py",
                "particle-sfm_amd/csrc/psfm_chain_step.h", "particle-sfm_amd/csrc/psfm_shard.hip", "particle-sfm_amd/csrc/psfm_matches.hip",
                "include/psfm.h", "oracle/psfm_oracle.c", "tests/golden/matches_40x56_t12.npz"):
        assert (out / rel).exists(), rel
    code = ("import sys; sys.path.insert(0, %r); import point_trajectory; from point_trajectory.optimize.build import particlesfm; "
            "import psfm_dist; from point_trajectory import shard; print('ok')" % str(out / "particle-sfm_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-800:]


def _small_set(n, seed=0):
    from point_trajectory.optimize.build import particlesfm
    rng = np.random.default_rng(seed)
    length = rng.integers(1, 30, n).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum(length)
    birth = rng.integers(0, 400, n).astype(np.int32)
    xy = rng.normal(size=(int(off[-1]), 2)) * 1e3
    ids = np.arange(n, dtype=np.int64) * 3
    return particlesfm.TrajectorySet._from_csr(ids, birth, length, off, xy), (ids, birth, length, off, xy)


@pytest.mark.parametrize("n", [0, 5, 2000])
def test_streamed_reference_layout_loads_with_the_pure_python_unpickler(tmp_path, n):
    """the streamed file must not depend on the C unpickler's tolerance of opcodes outside a declared FRAME --
    `pickle._Unpickler` (PyPy's only unpickler) reads the same object graph.  The point array is written writable."""
    import pickle
    from point_trajectory.trajectory import save_track_npy
    ts, (ids, birth, length, off, xy) = _small_set(n, seed=n)
    path = str(tmp_path / "t.npy")
    save_track_npy(path, ts)
    with open(path, "rb") as fp:
        ver = np.lib.format.read_magic(fp)
        assert ver == (1, 0)
        shape, fortran, dtype = np.lib.format.read_array_header_1_0(fp)
        assert shape == () and dtype == np.dtype(object)
        body = fp.read()
    assert b"\x95" not in body[:64] or True      # (no assumption on the template; the check below is the real one)
    import io
    arr = pickle._Unpickler(io.BytesIO(body)).load()
    back = arr.item() if hasattr(arr, "item") else arr
    assert sorted(back.trajs.keys()) == ids.tolist()
    for j in (0, n // 2, n - 1) if n else ():
        t = back.trajs[int(ids[j])]
        assert np.array_equal(np.asarray(t.as_dict()["locations"]), xy[off[j]:off[j + 1]])
        assert t.as_dict()["frame_ids"] == list(range(birth[j], birth[j] + length[j]))
    # raw state: locations are views of one writable array
    from point_trajectory.optimize.build import particlesfm

    class Recorder:
        def __setstate__(self, state):
            self.state = state
    real = particlesfm.TrajectorySet
    particlesfm.TrajectorySet = Recorder
    try:
        raw = pickle._Unpickler(io.BytesIO(body)).load().item().state
    finally:
        particlesfm.TrajectorySet = real
    if n:
        loc = raw[int(ids[0])]["locations"]
        assert isinstance(loc, np.ndarray) and loc.flags.writeable


def test_save_over_a_loaded_track_file_keeps_the_loaded_set_alive(tmp_path):
    """load_track_npy maps the points of the file; save_track_npy to the SAME path used to truncate the file in
    place and the next touch of the loaded set's points died with SIGBUS.  The writer now writes beside the target and renames."""
    from point_trajectory.trajectory import save_track_npy, load_track_npy
    ts, (ids, birth, length, off, xy) = _small_set(3000, seed=7)
    path = str(tmp_path / "track.npy")
    save_track_npy(path, ts)
    a = load_track_npy(path)
    assert a._csr is not None
    save_track_npy(path, a)                       # in place, from the mapped set itself
    assert np.array_equal(np.asarray(a._csr[4]), xy)          # the old mapping still reads the old bytes
    b = load_track_npy(path)
    assert np.array_equal(np.asarray(b._csr[4]), xy) and np.array_equal(b._csr[0], ids)
    ts2, (_, _, _, _, xy2) = _small_set(3000, seed=8)
    save_track_npy(path, ts2)                     # a different set over the same path while a and b are alive
    assert np.array_equal(np.asarray(a._csr[4]), xy) and np.array_equal(np.asarray(b._csr[4]), xy)
    assert np.array_equal(np.asarray(load_track_npy(path)._csr[4]), xy2)
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp-" in f]


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus N` without a launcher must run N ranks (it re-executes itself under
    torch.distributed.run) and never print a line for another size.  Without GPUs here every rank stops at its device check --
    what is asserted is that BOTH ranks were started, that no JSON line came out, and that the exit code is not 0; and that a
    job whose WORLD_SIZE differs from --gpus refuses to run."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "rank 0 needs cuda:0" in r.stdout and "rank 1 needs cuda:1" in r.stdout, r.stdout[-2000:]
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but the job has 1 rank" in r.stdout and '"metric"' not in r.stdout


def test_saving_over_an_existing_track_npy_replaces_it_and_leaves_nothing_behind(tmp_path):
    """save_track_npy over an existing output: the new file is moved over the old one (readers see one or the other, never a
    truncated file), the old file's pages are given back on a helper thread (a second link keeps the inode alive across the
    rename) -- and once that thread is through, the directory holds the one file."""
    import os
    from point_trajectory.optimize.build import particlesfm
    from point_trajectory.trajectory import save_track_npy, load_track_npy, wait_for_reclaims
    rng = np.random.default_rng(5)

    def make(n):
        length = rng.integers(1, 30, n).astype(np.int32)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(length)
        return particlesfm.TrajectorySet._from_csr(np.arange(n, dtype=np.int64), rng.integers(0, 50, n).astype(np.int32), length, off,
                                                   rng.normal(size=(int(off[-1]), 2)))
    path = str(tmp_path / "track.npy")
    first, second, third = make(2000), make(3100), make(10)
    save_track_npy(path, first)
    old_inode = os.stat(path).st_ino
    mapped = load_track_npy(path)                  # a set read back from the file that is about to be replaced stays readable
    save_track_npy(path, second)
    save_track_npy(path, third)
    wait_for_reclaims()
    assert sorted(os.listdir(tmp_path)) == ["track.npy"] and os.stat(path).st_ino != old_inode
    assert len(load_track_npy(path).trajs) == 10
    assert len(mapped.trajs) == 2000
