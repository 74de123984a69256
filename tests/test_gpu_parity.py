"""GPU parity: the HIP path (through the C ABI, via the point_trajectory mirror) against the CPU oracle and the
golden vectors produced by the reference's own Python.  Bit-exact for everything fp32/integer."""
import numpy as np
import pytest

from _common import golden, regen_inputs, assert_csr_equal, REALISTIC_TRACK
import psfm_synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import point_trajectory
    from point_trajectory import utils, trajectory, track, track_optimize, _hip
    _hip.context()   # fails loudly if libpsfm_hip.so is missing
    class NS: pass
    ns = NS()
    ns.utils, ns.trajectory, ns.track, ns.track_optimize, ns.hip = utils, trajectory, track.track, track_optimize.track_optimize, _hip
    return ns


@pytest.fixture(autouse=True, params=[1, 2], ids=["per-frame-launches", "persistent-loop"])
def chain_mode(request, pt):
    """Every test of this module runs twice: with one chain_step launch per frame, and with the persistent frame loop
    wherever it can run (track mode: ONE resident launch per sequence; mode 0 would pick it by shape).  Results must
    not differ."""
    ctx = pt.hip.context()
    ctx.set_chain_mode(request.param)
    yield request.param
    ctx.set_chain_mode(0)


def test_sampler_golden(pt):
    import torch
    g = golden("sampler")
    rng = np.random.default_rng(int(g["seed"]))
    H, W = int(g["H"]), int(g["W"])
    m2 = rng.standard_normal((H, W, 2)).astype(np.float32)
    m1 = (rng.uniform(size=(H, W)) < 0.3)
    s2 = pt.trajectory.grid_sample(torch.from_numpy(m2).permute(2, 0, 1), g["pts"])
    s1 = pt.trajectory.grid_sample(torch.from_numpy(m1.astype(np.float32)).unsqueeze(0), g["pts"])
    assert np.array_equal(s2.view(np.uint32), g["s2"].view(np.uint32))
    assert np.array_equal(s1.view(np.uint32), g["s1"].view(np.uint32))
    rng.uniform([-3, -3], [W + 2, H + 2], size=(4000, 2))
    mb = rng.standard_normal((int(g["Hb"]), int(g["Wb"]), 2)).astype(np.float32)
    sb = pt.trajectory.grid_sample(torch.from_numpy(mb).permute(2, 0, 1), g["pb"])
    assert np.array_equal(sb.view(np.uint32), g["sb"].view(np.uint32))


def test_flow_check_golden(pt):
    g = golden("flow_check")
    d = psfm_synth.synth_sequence(4, 64, 96, seed=21, sigma=0.4, n_occluders=2, stride2=False)
    for thres in (1.0, 3.0):
        err, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], thres)
        assert np.array_equal(np.stack(err).view(np.uint32), g["fc_err_%g" % thres].view(np.uint32))
        assert np.array_equal(np.packbits(np.stack(occ)), g["fc_occ_%g" % thres])
    dd = psfm_synth.synth_sequence(3, 40, 56, seed=22, amp=9.0, sigma=0.0, stride2=False)
    err, occ = pt.utils.flow_check(dd["flows_f"], dd["flows_b"], 1.0)
    assert np.array_equal(np.stack(err).view(np.uint32), g["big_err"].view(np.uint32))
    assert np.array_equal(np.packbits(np.stack(occ)), g["big_occ"])


def test_flow_check_vs_oracle_odd_sizes(pt):
    from oracle import oracle as orc
    for (H, W, seed) in [(37, 53, 1), (270, 480, 2), (33, 130, 3)]:
        d = psfm_synth.synth_sequence(3, H, W, seed=seed, sigma=0.5, n_occluders=1, stride2=False)
        e_o, o_o = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
        e_g, o_g = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
        for a, b in zip(e_o, e_g):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(o_o, o_g):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["track_48x64_r2", "track_45x70_r1", "track_50x66_r3", "track_52x61_r4",
                                  "track_largemotion_80x120_r2", "track_largemotion_75x110_r1"] + REALISTIC_TRACK)   # (~10 px of drift per frame; layered scene)
def test_track_golden(pt, name):
    g = golden(name)
    d = regen_inputs(g, stride2=False)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = pt.track(d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)


def test_nonfinite_flows_golden(pt):
    """NaN / +-Inf / huge flow components against the fixture made by the reference's torch ops: same error maps (NaN
    bit patterns included), masks and trajectories through every entry point, and no fault on the saturated taps."""
    import ctypes
    import torch
    g = golden("nonfinite_40x56_r2")
    T, H, W, r = int(g["T"]), int(g["H"]), int(g["W"]), int(g["ratio"])
    d = psfm_synth.poison_nonfinite(psfm_synth.synth_sequence(T, H, W, seed=int(g["seed"]), sigma=float(g["sigma"]),
                                                              n_occluders=int(g["n_occluders"]), stride2=False),
                                    seed=int(g["seed"]) + 1)
    err, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    ge, e = g["fc_err"], np.stack(err)
    assert np.array_equal(np.isnan(e), np.isnan(ge))
    assert np.array_equal(e[~np.isnan(ge)].view(np.uint32), ge[~np.isnan(ge)].view(np.uint32))
    assert np.array_equal(np.packbits(np.stack(occ)), g["fc_occ"])
    R = pt.track(d["flows_f"], occ, r)
    assert_csr_equal(R.birth, R.length, R.xy, g)
    hip = pt.hip
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    R = pt.trajectory.run_connect(ff, fb, None, None, 1.0, r)
    assert_csr_equal(R.birth, R.length, R.xy, g)
    # mask-only flow_check (the squared-threshold form) and the maps psfm_connect hands back
    occ_out = torch.full((T - 1, H, W), 9, dtype=torch.uint8, device="cuda")
    info = hip.TrackInfo()
    hip.check(hip.lib().psfm_connect(hip.context().handle, hip.ptr(ff), hip.ptr(fb), None, None, T - 1, H, W, 1.0, r,
                                     hip.ptr(occ_out), None, ctypes.byref(info), hip.current_stream_ptr()))
    assert np.array_equal(np.packbits(occ_out.cpu().numpy().astype(bool)), g["fc_occ"])


def test_track_all_tracks_die(pt):
    g = golden("track_alldie_24x30_r2")
    d = regen_inputs(g, stride2=False)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    occ = [o.copy() for o in occ]
    occ[1][:] = True
    R = pt.track(d["flows_f"], occ, int(g["ratio"]))
    assert_csr_equal(R.birth, R.length, R.xy, g)


@pytest.mark.parametrize("H,W,T,r,seed,sigma,nocc", [
    (270, 480, 12, 2, 41, 0.2, 3),
    (135, 241, 10, 3, 42, 0.35, 2),
    (96, 130, 30, 1, 43, 0.1, 2),
    (200, 300, 300, 4, 44, 0.05, 1),     # > 255 frames: exercises the occupied-stamp wrap
    (64, 90, 1100, 1, 45, 0.1, 1),       # long + dense: (last, birth, idx) no longer fits 32 bits -> 64-bit sort keys
])
def test_track_vs_oracle(pt, H, W, T, r, seed, sigma, nocc):
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=sigma, n_occluders=nocc, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ, r)
    R = pt.track(d["flows_f"], occ, r)
    assert R.birth.shape[0] == O.n_traj
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length)
    assert np.array_equal(R.xy, O.xy)
    assert (O.length < 3).any() and (O.length > 5).any()


def test_track_single_flow(pt):
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(2, 40, 60, seed=9, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ, 2)
    R = pt.track(d["flows_f"], occ, 2)
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)


def test_track_full_size_properties(pt):
    """1080p, sample_ratio=2 (BASELINE.json configs[1] shape, fewer frames): size-independent invariants."""
    import torch
    H, W, T, r = 1080, 1920, 12, 2
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False)
    _, occ = pt.utils.flow_check_device(d["flows_f"], d["flows_b"], 1.0)
    R = pt.trajectory.run_track(d["flows_f"], occ, None, None, r)
    n = len(R)
    GW, GH = (W + r - 1) // r, (H + r - 1) // r
    # offsets are the exclusive scan of the lengths
    assert R.off[0] == 0 and np.array_equal(np.diff(R.off), R.length) and R.off[-1] == R.n_points
    # every track lives inside [0, T-1]; ids are sorted by (last time, birth) -- full_trajs order
    last = R.birth + R.length - 1
    assert R.birth.min() == 0 and last.max() == T - 1 and (R.length >= 1).all()
    assert (np.diff(last) >= 0).all()
    same = np.diff(last) == 0
    assert (np.diff(R.birth)[same] >= 0).all()
    # frame-0 generation is the full grid; first positions are integer grid points
    assert int((R.birth == 0).sum()) == GW * GH
    first = R.xy[R.off[:-1]]
    assert np.array_equal(first, np.round(first)) and (first[:, 0] % r == 0).all() and (first[:, 1] % r == 0).all()
    # all later positions strictly inside the image (trajectory.py:56-57)
    assert (R.xy[:, 0] >= 0).all() and (R.xy[:, 0] <= W - 1).all() and (R.xy[:, 1] >= 0).all() and (R.xy[:, 1] <= H - 1).all()
    # determinism: a second run gives identical bits although lane assignment is free
    R2 = pt.trajectory.run_track(d["flows_f"], occ, None, None, r)
    assert np.array_equal(R.birth, R2.birth) and np.array_equal(R.length, R2.length) and np.array_equal(R.xy, R2.xy)
    # and the first frames agree with the CPU oracle run on the same tensors
    from oracle import oracle as orc
    k = 3
    ff = d["flows_f"][:k].cpu().numpy()
    oo = occ[:k].cpu().numpy()
    O = orc.track(list(ff), list(oo), r)
    Rk = pt.trajectory.run_track(d["flows_f"][:k], occ[:k], None, None, r)
    assert np.array_equal(Rk.birth, O.birth) and np.array_equal(Rk.length, O.length) and np.array_equal(Rk.xy, O.xy)


def test_flow_check_sharded_single_process(pt):
    """The frame-pair-sharded flow_check (psfm_dist) with world size 1 runs the HIP kernel on the whole stack."""
    import torch
    import psfm_dist
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(4, 37, 53, seed=5, sigma=0.4, n_occluders=1, stride2=False)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    occ = psfm_dist.flow_check_sharded(ff, fb, 1.0, lambda f, b, t: pt.utils.flow_check_device(f, b, t)[1])
    _, ref = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    assert np.array_equal(occ.cpu().numpy().astype(bool), np.stack(ref))
    pk = psfm_dist.pack_bits(occ)
    assert torch.equal(psfm_dist.unpack_bits(pk, 37, 53), occ)


def test_main_connect_end_to_end(pt, tmp_path):
    """The stage entry (main_connect_point_trajectories.py:27-62): .flo directories in, track.npy out, consumed
    through the access patterns of sfm/matches_from_flow.py and motion_seg/load_cut_seq.py; ids/lengths equal the
    oracle's, positions within tolerance (track_optimize path), and the skip_path_consistency variant bit-exact."""
    from oracle import oracle as orc
    from point_trajectory import main_connect_point_trajectories
    from point_trajectory.utils import write_flo, load_flows_device
    T, H, W, r = 7, 60, 80, 2
    d = psfm_synth.synth_sequence(T, H, W, seed=77, sigma=0.1, n_occluders=1, stride2=True)
    fd = tmp_path / "optical_flows"
    for key, sub in (("flows_f", "flow_f"), ("flows_b", "flow_b"), ("flows_f2", "flow_f2"), ("flows_b2", "flow_b2")):
        (fd / sub).mkdir(parents=True)
        for i, f in enumerate(d[key]):
            write_flo(str(fd / sub / ("%05d.flo" % i)), f)
    dev = load_flows_device(str(fd / "flow_f"))
    assert np.array_equal(dev.cpu().numpy(), np.stack(d["flows_f"]))
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    _, occ2 = orc.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    for skip in (True, False):
        out = tmp_path / ("traj_%d" % skip)
        main_connect_point_trajectories(str(fd), str(out), sample_ratio=r, skip_path_consistency=skip)
        ts = np.load(str(out / "track.npy"), allow_pickle=True).item()
        O = orc.track(d["flows_f"], occ, r) if skip else orc.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, r)
        keep = [i for i in range(O.n_traj) if O.length[i] >= 3]
        dd = ts.as_dict()
        assert sorted(dd) == keep
        for i in keep[::7]:
            assert dd[i]["frame_ids"] == list(range(O.birth[i], O.birth[i] + O.length[i]))
            err = np.abs(np.array(dd[i]["locations"]) - O.xy[O.off[i]:O.off[i + 1]]).max()
            assert err == 0.0 if skip else err <= 1e-4
        ts.build_invert_indexes()
        win = ts.sample_inside_window(list(range(T)), max_num_tracks=100000)
        assert len(win["traj_ids"]) == len(keep)
        # skip_exists short-circuit (main_connect_point_trajectories.py:31-33)
        main_connect_point_trajectories(str(fd), str(out), sample_ratio=r, skip_path_consistency=skip, skip_exists=True)


def test_flo_stack_ingest(pt, tmp_path, monkeypatch):
    """utils.py:26-56 for a whole stack straight into HBM (psfm_load_flo_stack: reader threads -> pinned ring -> async H2D): the
    bytes of the files, for more files than ring slots and for one file; the Python pipeline of the same design (PSFM_FLO_NATIVE=0)
    gives the same tensor; a truncated file, a foreign file and a frame of another size are refused with the file named."""
    from point_trajectory.utils import write_flo, load_flows_device
    rng = np.random.default_rng(7)
    H, W = 37, 53
    frames = [rng.standard_normal((H, W, 2)).astype(np.float32) for _ in range(41)]
    d = tmp_path / "stack"
    d.mkdir()
    for i, f in enumerate(frames):
        write_flo(str(d / ("%05d.flo" % i)), f)
    for readers in (1, 3, 8):
        got = load_flows_device(str(d), n_readers=readers)
        assert got.shape == (41, H, W, 2) and np.array_equal(got.cpu().numpy(), np.stack(frames))
    monkeypatch.setenv("PSFM_FLO_NATIVE", "0")
    assert np.array_equal(load_flows_device(str(d)).cpu().numpy(), np.stack(frames))
    monkeypatch.delenv("PSFM_FLO_NATIVE")
    one = tmp_path / "one"
    one.mkdir()
    write_flo(str(one / "00000.flo"), frames[0])
    assert np.array_equal(load_flows_device(str(one)).cpu().numpy(), np.stack(frames[:1]))
    assert load_flows_device(str(tmp_path / "nothing")).shape == (0, 0, 0, 2)
    bad = tmp_path / "bad"
    for kind in ("truncated", "foreign", "size"):
        if bad.exists():
            for q in bad.iterdir():
                q.unlink()
        else:
            bad.mkdir()
        for i in range(5):
            write_flo(str(bad / ("%05d.flo" % i)), frames[i])
        victim = bad / "00003.flo"
        raw = victim.read_bytes()
        if kind == "truncated":
            victim.write_bytes(raw[:len(raw) // 2])
        elif kind == "foreign":
            victim.write_bytes(b"NOPE" + raw[4:])          # (the real tag reads "PIEH")
        else:
            write_flo(str(victim), rng.standard_normal((H + 1, W, 2)).astype(np.float32))
        with pytest.raises(RuntimeError) as ei:
            load_flows_device(str(bad))
        assert "00003.flo" in str(ei.value)


@pytest.mark.parametrize("H,W,T,r,seed,thres", [
    (41, 59, 6, 5, 91, 1.0),      # generic-ratio kernel instantiation (R = 0), H*W odd -> scalar flow_check kernel
    (30, 44, 5, 6, 92, 3.0),      # ratio 6, README's ScanNet threshold
    (8, 9, 4, 2, 93, 1.0),        # tiny image: every tap near a border
    (64, 64, 6, 7, 94, 1.0),      # ratio larger than most displacements
])
def test_track_odd_ratios_and_sizes(pt, H, W, T, r, seed, thres):
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.3, n_occluders=1, stride2=False)
    e_o, o_o = orc.flow_check(d["flows_f"], d["flows_b"], thres)
    e_g, o_g = pt.utils.flow_check(d["flows_f"], d["flows_b"], thres)
    for a, b in zip(e_o, e_g):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for a, b in zip(o_o, o_g):
        assert np.array_equal(a, b)
    O = orc.track(d["flows_f"], o_o, r)
    R = pt.track(d["flows_f"], o_g, r)
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)


def test_capacity_overflow_is_reported_and_retried(pt):
    """Tables that are too small return PSFM_ERR_CAPACITY (never a silently truncated result); the Python mirror
    grows them and reruns.  Heavy occlusion makes the trajectory table overflow its minimum size."""
    import ctypes
    import torch
    from oracle import oracle as orc
    hip = pt.hip
    T, H, W, r = 40, 96, 128, 1
    d = psfm_synth.synth_sequence(T, H, W, seed=95, sigma=0.6, n_occluders=4, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ, r)
    ctx = hip.context()
    fl = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    oc = torch.from_numpy(np.stack(occ).astype(np.uint8)).cuda()
    ctx.set_capacity(1.0, 1.0)
    info = hip.TrackInfo()
    st = hip.lib().psfm_track(ctx.handle, hip.ptr(fl), hip.ptr(oc), None, None, T - 1, H, W, r, ctypes.byref(info),
                              hip.current_stream_ptr())
    if O.n_traj > 2 * H * W + 64 * 1024:      # only then is the minimum table guaranteed too small
        assert st == hip.PSFM_ERR_CAPACITY and b"capacity" in hip.lib().psfm_last_error()
    R = pt.track(d["flows_f"], occ, r)        # mirror: retries with larger tables
    assert len(R) == O.n_traj and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)


def test_bad_arguments_are_rejected(pt):
    import ctypes
    hip = pt.hip
    ctx = hip.context()
    L = hip.lib()
    assert L.psfm_flow_check(ctx.handle, None, None, 3, 10, 10, 1.0, None, None, None) == hip.PSFM_ERR_ARG
    assert L.psfm_track(ctx.handle, None, None, None, None, 0, 10, 10, 2, None, None) == hip.PSFM_ERR_ARG
    assert L.psfm_track(ctx.handle, None, None, None, None, 3, 10, 10, 0, None, None) == hip.PSFM_ERR_ARG
    assert L.psfm_grid_sample(ctx.handle, None, 3, 10, 10, None, 5, None, None) == hip.PSFM_ERR_ARG
    assert L.psfm_ctx_set_capacity(ctx.handle, 0.5, 1.0) == hip.PSFM_ERR_ARG
    assert len(L.psfm_last_error()) > 0
    with pytest.raises(ValueError):
        pt.track([], [], 2)
    # frames whose (H,W,2) f32 map does not fit 32-bit byte offsets (8 H W >= 2^32) are refused by every entry point that takes h, w --
    # before a single byte of the (here far too small) buffers is touched
    import torch
    t = torch.zeros(64, dtype=torch.float32, device="cuda")
    o = torch.zeros(64, dtype=torch.uint8, device="cuda")
    d64 = torch.zeros(64, dtype=torch.float64, device="cuda")
    P, sp = hip.ptr, hip.current_stream_ptr()
    info = hip.TrackInfo()
    for h, w in ((32768, 16384), (23171, 23171), (2, 1 << 28)):
        assert 8 * h * w >= 1 << 32
        assert L.psfm_flow_check(ctx.handle, P(t), P(t), 1, h, w, 1.0, P(o), None, sp) == hip.PSFM_ERR_ARG
        assert L.psfm_grid_sample(ctx.handle, P(t), 2, h, w, P(d64), 1, P(t), sp) == hip.PSFM_ERR_ARG
        assert L.psfm_optimize_location(ctx.handle, P(d64), P(d64), P(d64), P(d64), P(t), 1, w, h, P(d64), None, sp) == hip.PSFM_ERR_ARG
        assert L.psfm_path_consistency_eval(ctx.handle, P(d64), P(d64), P(d64), P(d64), P(t), 1, w, h, P(d64), P(d64), sp) == hip.PSFM_ERR_ARG
        assert L.psfm_track(ctx.handle, P(t), P(o), None, None, 1, h, w, 2, ctypes.byref(info), sp) == hip.PSFM_ERR_ARG
        assert L.psfm_connect(ctx.handle, P(t), P(t), None, None, 1, h, w, 1.0, 2, None, None, ctypes.byref(info), sp) == hip.PSFM_ERR_ARG
    assert L.psfm_flow_check(ctx.handle, P(t), P(t), 0, 23170, 23170, 1.0, P(o), None, sp) == hip.PSFM_OK      # (the largest square frame: 8 H W < 2^32)


def test_connect_sequences_concurrently(pt, tmp_path):
    """point_trajectory.batch: several sequences in flight on one GPU (separate streams + contexts) give exactly the
    files that one-at-a-time processing gives."""
    from point_trajectory.batch import connect_sequences
    from point_trajectory import main_connect_point_trajectories
    from point_trajectory.utils import write_flo
    flow_dirs, out_a, out_b = [], [], []
    for sidx in range(4):
        d = psfm_synth.synth_sequence(6, 50 + 4 * sidx, 70, seed=200 + sidx, sigma=0.1, n_occluders=1, stride2=True)
        fd = tmp_path / ("seq%d" % sidx) / "optical_flows"
        for key, sub in (("flows_f", "flow_f"), ("flows_b", "flow_b"), ("flows_f2", "flow_f2"), ("flows_b2", "flow_b2")):
            (fd / sub).mkdir(parents=True)
            for i, f in enumerate(d[key]):
                write_flo(str(fd / sub / ("%05d.flo" % i)), f)
        flow_dirs.append(str(fd))
        out_a.append(str(tmp_path / ("seq%d" % sidx) / "traj_seq"))
        out_b.append(str(tmp_path / ("seq%d" % sidx) / "traj_par"))
        main_connect_point_trajectories(flow_dirs[-1], out_a[-1], sample_ratio=2)
    done = connect_sequences(flow_dirs, out_b, sample_ratio=2, concurrency=3)
    assert done == [0, 1, 2, 3]
    for a, b in zip(out_a, out_b):
        ta = np.load(a + "/track.npy", allow_pickle=True).item()
        tb = np.load(b + "/track.npy", allow_pickle=True).item()
        # (the stage entry writes the reference's pickle state by default, the batch driver the CSR state: compare
        # the trajectories, not the container)
        ca, cb = ta._to_csr(), tb._to_csr()
        for k, (xa, xb) in enumerate(zip(ca[:4], cb[:4])):
            assert np.array_equal(xa, xb), ("ids", "off", "frames", "xy")[k]


def test_chain_modes_report_what_ran(pt, chain_mode):
    """psfm_track_info.chain_mode: 2 = persistent frame loop, 1 = per-frame launches; track_optimize is always 1."""
    d = psfm_synth.synth_sequence(8, 60, 80, seed=5, sigma=0.2, n_occluders=1, stride2=True)
    _, occ = pt.utils.flow_check(d["flows_f"], d["flows_b"], 1.0)
    R = pt.track(d["flows_f"], occ, 2)
    assert R.info["chain_mode"] == (1 if chain_mode == 1 else 2)
    _, occ2 = pt.utils.flow_check(d["flows_f2"], d["flows_b2"], 1.0)
    R2 = pt.track_optimize(d["flows_f"], d["flows_f2"], occ, occ2, 2)
    assert R2.info["chain_mode"] == 1


def test_persistent_loop_hands_over_to_per_frame_launches(pt, chain_mode, monkeypatch):
    """A persistent loop that gives up (here: barrier spin limit forced to 0) must not produce a result of its own:
    psfm_track reruns the sequence with per-frame launches and returns exactly the same trajectories."""
    if chain_mode == 1:
        pytest.skip("per-frame mode never starts the persistent loop")
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(10, 120, 160, seed=77, sigma=0.3, n_occluders=2, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ, 1)
    monkeypatch.setenv("PSFM_PERSIST_SPIN_LIMIT", "0")
    R = pt.track(d["flows_f"], occ, 1)
    assert R.info["chain_mode"] == 1
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
    # the same through psfm_connect, where the loop would also have produced the occlusion maps
    import torch
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    Rc = pt.trajectory.run_connect(ff, fb, None, None, 1.0, 1)
    assert Rc.info["chain_mode"] == 1
    assert np.array_equal(Rc.birth, O.birth) and np.array_equal(Rc.length, O.length) and np.array_equal(Rc.xy, O.xy)
    monkeypatch.delenv("PSFM_PERSIST_SPIN_LIMIT")
    R = pt.track(d["flows_f"], occ, 1)
    assert R.info["chain_mode"] == 2
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
    Rc = pt.trajectory.run_connect(ff, fb, None, None, 1.0, 1)
    assert Rc.info["chain_mode"] == 2
    assert np.array_equal(Rc.birth, O.birth) and np.array_equal(Rc.length, O.length) and np.array_equal(Rc.xy, O.xy)


def test_persistent_loop_cooperative_launch(pt, chain_mode, monkeypatch):
    """PSFM_PERSIST_COOP=1: the persistent frame loop as a cooperative launch (refused by the runtime when its grid cannot be
    co-resident).  Same trajectories as the plain launch; chain_mode 2 reports that the loop ran."""
    if chain_mode == 1:
        pytest.skip("per-frame launches only")
    from oracle import oracle as orc
    d = psfm_synth.synth_sequence(9, 120, 160, seed=66, sigma=0.3, n_occluders=2, stride2=False)
    _, occ = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ, 2)
    monkeypatch.setenv("PSFM_PERSIST_COOP", "1")
    R = pt.track(d["flows_f"], occ, 2)
    assert R.info["chain_mode"] == 2
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)


def test_chain_mode_policy(pt):
    """Mode 2 runs the persistent loop wherever it can and per-frame launches elsewhere (grid larger than the resident
    lanes, track_optimize); mode 0 decides by shape: sample_ratio >= 2 with >= 100 k grid points for psfm_track,
    additionally >= 400 k grid points and <= 6 pixels per grid point for the fused psfm_connect."""
    import ctypes
    import torch
    hip = pt.hip
    ctx = hip.context()

    def track(H, W, r, connect=False):
        fl = torch.zeros((2, H, W, 2), dtype=torch.float32, device="cuda")
        oc = torch.zeros((2, H, W), dtype=torch.uint8, device="cuda")
        info = hip.TrackInfo()
        if connect:
            hip.check(hip.lib().psfm_connect(ctx.handle, hip.ptr(fl), hip.ptr(fl), None, None, 2, H, W, 1.0, r, None, None,
                                             ctypes.byref(info), hip.current_stream_ptr()))
        else:
            hip.check(hip.lib().psfm_track(ctx.handle, hip.ptr(fl), hip.ptr(oc), None, None, 2, H, W, r, ctypes.byref(info),
                                           hip.current_stream_ptr()))
        assert info.n_traj >= ((H + r - 1) // r) * ((W + r - 1) // r)
        return info.chain_mode
    try:
        ctx.set_chain_mode(2)
        assert track(1080, 1920, 1) == 1            # 2.07 M grid points: more than the device holds resident
        assert track(96, 128, 1) == 2               # anything that fits
        ctx.set_chain_mode(0)
        assert track(1080, 1920, 2) == 2 and track(1080, 1920, 2, connect=True) == 2
        assert track(720, 1280, 2) == 2 and track(720, 1280, 2, connect=True) == 1
        assert track(540, 960, 1) == 1 and track(96, 128, 2) == 1
        ctx.set_chain_mode(1)
        assert track(1080, 1920, 2) == 1
    finally:
        ctx.set_chain_mode(0)


@pytest.mark.parametrize("H,W,T,r", [(64, 128, 9, 2), (45, 70, 8, 1), (120, 160, 6, 3)])
def test_connect_fused_flow_check(pt, chain_mode, H, W, T, r):
    """psfm_connect = flow_check + track.  With the device to itself (default mode) the persistent loop computes the
    occlusion maps itself; the maps it hands back and the trajectories equal the two-call form and the oracle."""
    import ctypes
    import torch
    from oracle import oracle as orc
    hip = pt.hip
    d = psfm_synth.synth_sequence(T, H, W, seed=123 + H, sigma=0.3, n_occluders=2, stride2=False)
    _, occ_o = orc.flow_check(d["flows_f"], d["flows_b"], 1.0)
    O = orc.track(d["flows_f"], occ_o, r)
    ff = torch.from_numpy(np.stack(d["flows_f"])).cuda()
    fb = torch.from_numpy(np.stack(d["flows_b"])).cuda()
    ctx = hip.context()
    for give_occ in (False, True):
        occ_out = torch.full((T - 1, H, W), 9, dtype=torch.uint8, device="cuda") if give_occ else None
        info = hip.TrackInfo()
        hip.check(hip.lib().psfm_connect(ctx.handle, hip.ptr(ff), hip.ptr(fb), None, None, T - 1, H, W, 1.0, r,
                                         hip.ptr(occ_out), None, ctypes.byref(info), hip.current_stream_ptr()))
        R = pt.trajectory._result_to_host(ctx, info)
        assert info.chain_mode == (1 if chain_mode == 1 else 2)
        assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
        if give_occ:
            assert np.array_equal(occ_out.cpu().numpy().astype(bool), np.stack(occ_o).astype(bool))


def test_random_sequences_all_paths_agree(pt):
    """Randomised shapes / noise / occluders: psfm_connect with per-frame launches, psfm_connect with the fused
    persistent loop and flow_check + psfm_track (persistent loop on ready maps) return identical trajectories."""
    import torch
    rng = np.random.default_rng(2024)
    ctx = pt.hip.context()
    try:
        for i in range(24):
            H, W = int(rng.integers(20, 300)), int(rng.integers(20, 400))
            T, r = int(rng.integers(2, 30)), int(rng.choice([1, 2, 2, 3, 4]))
            d = psfm_synth.synth_sequence_torch(T, H, W, seed=int(rng.integers(0, 1 << 30)), sigma=float(rng.choice([0.05, 0.3, 0.8])),
                                                n_occluders=int(rng.integers(0, 4)), stride2=False)
            res = []
            for mode in (1, 2):
                ctx.set_chain_mode(mode)
                res.append(pt.trajectory.run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r))
            _, occ = pt.utils.flow_check_device(d["flows_f"], d["flows_b"], 1.0)
            res.append(pt.trajectory.run_track(d["flows_f"], occ, None, None, r))
            assert res[0].info["chain_mode"] == 1 and res[1].info["chain_mode"] == 2 and res[2].info["chain_mode"] == 2
            for B in res[1:]:
                assert np.array_equal(res[0].birth, B.birth) and np.array_equal(res[0].length, B.length)
                assert np.array_equal(res[0].xy, B.xy), (i, H, W, T, r)
    finally:
        ctx.set_chain_mode(0)


def test_region_without_survivors_still_respawns_its_corner(pt):
    """Every track in the top rows dies in one step while the rest of the image survives: grid point (0,0) must respawn
    like its neighbours (the SciPy 'no survivor at all' corner rule applies only when NOTHING survived anywhere)."""
    from oracle import oracle as orc
    H, W, T, r = 48, 64, 5, 1
    flows = [np.zeros((H, W, 2), np.float32) for _ in range(T - 1)]
    occ = [np.zeros((H, W), bool) for _ in range(T - 1)]
    occ[1][:8, :] = True
    O = orc.track(flows, occ, r)
    R = pt.track(flows, occ, r)
    assert np.array_equal(R.birth, O.birth) and np.array_equal(R.length, O.length) and np.array_equal(R.xy, O.xy)
    first = O.xy[np.concatenate([[0], np.cumsum(O.length)[:-1]])]
    assert ((O.birth == 2) & (first[:, 0] == 0) & (first[:, 1] == 0)).any()      # the corner itself respawned at frame 2


def test_motion_seg_window_tensors(pt):
    """SURVEY f-4: psfm_window_sample (device) against the host path the reference takes --
    TrajectorySet.sample_inside_window per window (trajectory_base.cpp:127-185) + resize_point_traj / normalize_point_traj
    (motion_seg/core/dataset/data_utils.py:74-89), restated in NumPy below."""
    from psfm_motion_seg.load_cut_seq import cut_trajectory_windows, window_ranges, sample_window_device
    T, H, W, r = 23, 96, 128, 2
    d = psfm_synth.synth_sequence_torch(T, H, W, seed=31, sigma=0.3, n_occluders=2, stride2=False)
    R = pt.trajectory.run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, r)      # result stays in the context
    ts = R.to_trajectory_set(3)
    ts.build_invert_indexes()
    raw_hw, input_size, window = (H, W), (60, 100), 10
    got = cut_trajectory_windows(T, window, raw_hw, input_size, traj_max_num=10 ** 9, as_numpy=True)
    assert window_ranges(T, window) == [(0, 10), (10, 10), (13, 10)]
    for w, (f0, n) in enumerate(window_ranges(T, window)):
        out = ts.sample_inside_window(list(range(f0, f0 + n)), min_length=3, max_num_tracks=10 ** 9)
        raw = np.concatenate([out["locations"][0][:, :, None], out["locations"][1][:, :, None]], 2)   # load_cut_seq.py:76
        nor = raw.copy()
        nor[:, :, 0] /= float(raw_hw[1]) / float(input_size[1]); nor[:, :, 1] /= float(raw_hw[0]) / float(input_size[0])
        nor[:, :, 0] /= input_size[1]; nor[:, :, 1] /= input_size[0]
        nor = np.clip(nor, 0.0, 1.0)
        mask = (1 - out["masks"]).astype(float)[:, :, None]
        assert np.array_equal(got[4][w], np.asarray(out["traj_ids"], np.int32)) and len(out["traj_ids"]) > 50
        assert np.array_equal(got[0][w], raw) and np.array_equal(got[1][w], nor) and np.array_equal(got[2][w], mask)
        assert np.array_equal(got[3][w], np.arange(f0, f0 + n))
    # window >= sequence: one window over everything (load_cut_seq.py:50-58)
    one = cut_trajectory_windows(T, 64, raw_hw, input_size, traj_max_num=10 ** 9, as_numpy=True)
    full = ts.sample_inside_window(list(range(T)), max_num_tracks=10 ** 9)
    assert len(one[0]) == 1 and np.array_equal(one[4][0], np.asarray(full["traj_ids"], np.int32))
    # more trajectories than max_num_tracks: a seeded random subset (the reference shuffles unseeded)
    ctx = pt.hip.context()
    all_ids = set(full["traj_ids"])
    a = sample_window_device(ctx, 0, T, raw_hw, input_size, traj_max_num=40, seed=5)
    b = sample_window_device(ctx, 0, T, raw_hw, input_size, traj_max_num=40, seed=5)
    c2 = sample_window_device(ctx, 0, T, raw_hw, input_size, traj_max_num=40, seed=6)
    ia, ib, ic = (x[0].cpu().numpy() for x in (a, b, c2))
    assert len(ia) == 40 and len(set(ia.tolist())) == 40 and set(ia.tolist()) <= all_ids
    assert np.array_equal(ia, ib) and not np.array_equal(ia, ic)
    row = {i: k for k, i in enumerate(full["traj_ids"])}
    sel = np.array([row[i] for i in ia.tolist()])
    assert np.array_equal(a[1].cpu().numpy()[:, :, 0], full["locations"][0][sel]) and np.array_equal(a[3].cpu().numpy()[:, :, 0], 1.0 - full["masks"][sel])


def test_saved_set_filtered_on_the_device(pt, tmp_path):
    """psfm_result_filter (min-length filter in HBM, main_connect_point_trajectories.py:56-61) + the protocol-5 .npy
    writer give exactly what the host path gives: TrajectoryList.to_trajectory_set + np.save."""
    from point_trajectory.trajectory import result_to_trajectory_set, save_track_npy
    d = psfm_synth.synth_sequence_torch(14, 90, 120, seed=77, sigma=0.4, n_occluders=2, stride2=False)
    info = pt.trajectory.run_connect(d["flows_f"], d["flows_b"], None, None, 1.0, 2, return_device=True)
    ctx = pt.hip.context()
    host = pt.trajectory._result_to_host(ctx, info).to_trajectory_set(3)
    for pinned in (False, True):
        dev = result_to_trajectory_set(ctx, info, 3, reuse_pinned=pinned)
        for a, b in zip(host._csr[:5], dev._csr[:5]):
            assert np.array_equal(a, b)
        assert len(dev._csr[0]) < info.n_traj        # something was filtered out
    save_track_npy(str(tmp_path / "track.npy"), dev)
    np.save(str(tmp_path / "track_ref.npy"), host)
    a = np.load(str(tmp_path / "track.npy"), allow_pickle=True).item()
    b = np.load(str(tmp_path / "track_ref.npy"), allow_pickle=True).item()
    # (the files carry the reference's pickle state by default: compare the trajectories, whatever backs the sets)
    for x, y in zip(a._to_csr()[:4], b._to_csr()[:4]):
        assert np.array_equal(x, y)
    for x, y in zip(a._to_csr()[:4], dev._to_csr()[:4]):
        assert np.array_equal(x, y)
    a.build_invert_indexes()
    assert sorted(a.as_dict()) == sorted(b.as_dict())
    # empty result of the filter (every trajectory shorter than the minimum)
    none = result_to_trajectory_set(ctx, info, 10 ** 6)
    assert len(none._csr[0]) == 0 and none._csr[4].shape == (0, 2)
