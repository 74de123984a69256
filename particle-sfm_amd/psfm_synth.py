"""Deterministic synthetic optical-flow stacks.

The reference ships no data (its examples are fetched by
scripts/download_examples.sh); every parity / bench input is synthesised here.
Flows are forward/backward consistent (otherwise every track dies on frame 1):

    F_t(x,y) = A * [ sin(2pi*1.5x/W + p0) cos(2pi*y/H + p1),
                     cos(2pi*x/W + p2) sin(2pi*1.5y/H + p3) ] + sigma*N(0,1)
    B_t      = -F_t^clean + sigma*N(0,1)          (inside "occluder" boxes: B = +F)
    F2_t     = F_t^clean + F_{t+1}^clean(id + F_t^clean) + sigma*N(0,1)   (stride 2)

`synth_sequence` is NumPy (host, bit-reproducible from the seed, used by tests and
fixtures); `synth_sequence_torch` evaluates the same formula with torch on any
device (used by bench.py to fill HBM without a PCIe copy; not bit-identical to
the NumPy version, which does not matter because both paths under test read
the same tensors).
"""
import math

import numpy as np


def _clean_np(xx, yy, H, W, amp, ph):
    u = amp * np.sin(2 * np.pi * 1.5 * xx / W + ph[0]) * np.cos(2 * np.pi * yy / H + ph[1])
    v = amp * np.cos(2 * np.pi * xx / W + ph[2]) * np.sin(2 * np.pi * 1.5 * yy / H + ph[3])
    return u, v


def _clean_drift_np(xx, yy, H, W, amp, ph, drift):
    u, v = _clean_np(xx, yy, H, W, amp, ph)
    return u + drift[0], v + drift[1]


def _backward_of(fwd, xx, yy):
    """B(q) = -F(p) with p + F(p) = q, by three fixed-point steps p <- q - F(p) (the flows used here move a few percent per
    pixel, so this is exact to far below the noise)."""
    px, py = xx, yy
    for _ in range(3):
        u, v = fwd(px, py)
        px, py = xx - u, yy - v
    u, v = fwd(px, py)
    return -u, -v


def synth_sequence(n_frames, H, W, seed=0, amp=3.0, sigma=0.05, n_occluders=0, stride2=True, drift=(0.0, 0.0), warp_b=False):
    """Returns dict(flows_f, flows_b[, flows_f2, flows_b2]) of lists of (H,W,2) float32 arrays.

    n_frames images -> n_frames-1 stride-1 pairs and n_frames-2 stride-2 pairs.
    drift: a constant (dx, dy) added to every clean forward flow (large motion: tracks cross the image and leave it, stride-2
    flows reach the reference's 20 px gate, trajectory.py:179).  warp_b: the backward flow is the true inverse of the clean
    forward flow (B(p + F(p)) = -F(p)) instead of -F at the same pixel -- what keeps large flows forward/backward consistent.
    The defaults reproduce the generator the committed fixtures were made with."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    n_pairs = n_frames - 1
    phases = rng.uniform(0, 2 * np.pi, size=(n_pairs + 1, 4))
    plain = (float(drift[0]) == 0.0 and float(drift[1]) == 0.0 and not warp_b)

    def noisy(u, v):
        out = np.stack([u, v], -1)
        if sigma > 0:
            out = out + sigma * rng.standard_normal(out.shape)
        return out.astype(np.float32)

    def occlude(fb, ff_clean):
        for _ in range(n_occluders):
            bh, bw = max(2, H // 8), max(2, W // 8)
            y0 = int(rng.integers(0, H - bh + 1))
            x0 = int(rng.integers(0, W - bw + 1))
            fb[y0:y0 + bh, x0:x0 + bw, :] = ff_clean[y0:y0 + bh, x0:x0 + bw, :]
        return fb

    def fwd1(t):
        return lambda px, py: _clean_drift_np(px, py, H, W, amp, phases[t], drift)

    def fwd2(t):
        def f(px, py):
            u, v = fwd1(t)(px, py)
            u2, v2 = fwd1(t + 1)(px + u, py + v)
            return u + u2, v + v2
        return f

    out = {"flows_f": [], "flows_b": []}
    for t in range(n_pairs):
        if plain:
            u, v = _clean_np(xx, yy, H, W, amp, phases[t])
            bu, bv = -u, -v
        else:
            u, v = fwd1(t)(xx, yy)
            bu, bv = _backward_of(fwd1(t), xx, yy) if warp_b else (-u, -v)
        out["flows_f"].append(noisy(u, v))
        fb = noisy(bu, bv)
        out["flows_b"].append(occlude(fb, np.stack([u, v], -1).astype(np.float32)))
    if stride2:
        out["flows_f2"], out["flows_b2"] = [], []
        for t in range(n_pairs - 1):
            if plain:
                u, v = _clean_np(xx, yy, H, W, amp, phases[t])
                u2, v2 = _clean_np(xx + u, yy + v, H, W, amp, phases[t + 1])
                su, sv = u + u2, v + v2
                bu, bv = -su, -sv
            else:
                su, sv = fwd2(t)(xx, yy)
                bu, bv = _backward_of(fwd2(t), xx, yy) if warp_b else (-su, -sv)
            out["flows_f2"].append(noisy(su, sv))
            fb = noisy(bu, bv)
            out["flows_b2"].append(occlude(fb, np.stack([su, sv], -1).astype(np.float32)))
    return out


def synth_sequence_torch(n_frames, H, W, seed=0, amp=3.0, sigma=0.05, n_occluders=0, stride2=False,
                         device="cuda", drift=(0.0, 0.0), warp_b=False):
    """Same formula evaluated with torch on `device`.  Returns dict of (n,H,W,2) float32 tensors."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    host_rng = np.random.default_rng(seed)
    n_pairs = n_frames - 1
    phases = host_rng.uniform(0, 2 * np.pi, size=(n_pairs + 1, 4))
    ys = torch.arange(H, device=device, dtype=torch.float32)[:, None]
    xs = torch.arange(W, device=device, dtype=torch.float32)[None, :]
    dx, dy = float(drift[0]), float(drift[1])

    def clean(xx, yy, ph):
        u = amp * torch.sin(2 * math.pi * 1.5 * xx / W + ph[0]) * torch.cos(2 * math.pi * yy / H + ph[1]) + dx
        v = amp * torch.cos(2 * math.pi * xx / W + ph[2]) * torch.sin(2 * math.pi * 1.5 * yy / H + ph[3]) + dy
        return u, v

    def clean2(xx, yy, t):
        u, v = clean(xx, yy, phases[t])
        u2, v2 = clean(xx + u, yy + v, phases[t + 1])
        return u + u2, v + v2

    def backward(fwd):
        if not warp_b:
            u, v = fwd(xs, ys)
            return -u, -v
        return _backward_of(fwd, xs, ys)

    def noisy(dst, u, v):
        dst[..., 0] = u
        dst[..., 1] = v
        if sigma > 0:
            dst += sigma * torch.randn(dst.shape, generator=gen, device=device, dtype=torch.float32)

    def occlude(fb, ff):
        for _ in range(n_occluders):
            bh, bw = max(2, H // 8), max(2, W // 8)
            y0 = int(host_rng.integers(0, H - bh + 1))
            x0 = int(host_rng.integers(0, W - bw + 1))
            fb[y0:y0 + bh, x0:x0 + bw, :] = ff[y0:y0 + bh, x0:x0 + bw, :]

    out = {"flows_f": torch.empty((n_pairs, H, W, 2), device=device, dtype=torch.float32),
           "flows_b": torch.empty((n_pairs, H, W, 2), device=device, dtype=torch.float32)}
    if stride2:
        out["flows_f2"] = torch.empty((max(n_pairs - 1, 0), H, W, 2), device=device, dtype=torch.float32)
        out["flows_b2"] = torch.empty((max(n_pairs - 1, 0), H, W, 2), device=device, dtype=torch.float32)
    for t in range(n_pairs):
        u, v = clean(xs, ys, phases[t])
        u, v = u.expand(H, W), v.expand(H, W)
        noisy(out["flows_f"][t], u, v)
        bu, bv = backward(lambda xx, yy: clean(xx, yy, phases[t]))
        noisy(out["flows_b"][t], bu.expand(H, W), bv.expand(H, W))
        occlude(out["flows_b"][t], torch.stack([u, v], -1))
        if stride2 and t < n_pairs - 1:
            su, sv = clean2(xs, ys, t)
            noisy(out["flows_f2"][t], su, sv)
            bu, bv = backward(lambda xx, yy: clean2(xx, yy, t))
            noisy(out["flows_b2"][t], bu, bv)
            occlude(out["flows_b2"][t], torch.stack([su, sv], -1))
    return out


# SURVEY 8(d)'s second distribution ("sigma = 0.3 and 5 % rectangular occluder regions, to exercise deaths / respawn / kinks"):
# three boxes of H/8 x W/8 = 4.7 % of the image per backward field.
HARD = dict(sigma=0.3, n_occluders=3)


NONFINITE_VALUES = (float("nan"), float("inf"), float("-inf"), 1e30, -1e30, 3e38, 1e10, -1e10, 2147483648.0, -2147483904.0)


def poison_nonfinite(d, seed, per_field=40):
    """Overwrite `per_field` random components of every flow field of a synth_sequence() result with NaN, +-Inf and
    huge finite values (what a broken flow network could emit).  Returns the same dict, arrays modified in place."""
    rng = np.random.default_rng(seed)
    vals = np.array(NONFINITE_VALUES, np.float32)
    for k in sorted(d):
        for a in d[k]:
            H, W = a.shape[:2]
            for _ in range(per_field):
                a[rng.integers(0, H), rng.integers(0, W), rng.integers(0, 2)] = vals[rng.integers(0, len(vals))]
    return d



# ------------------------------------------------------------------------------------------------------------------------------
# A third distribution: what RAFT on real video looks like to this path.  The two above are a smooth field with
# i.i.d. per-pixel noise (sigma 0.05: every solve converges in three Gauss-Newton steps; sigma 0.3: the interpolated field is rough
# at the pixel scale and every solve takes 20-40 iterations) and rectangular "occluders" that are just B = +F.  Real flow has
#   * piecewise-smooth motion: a few rigid / affine LAYERS in depth order (a moving camera over a background, objects in front);
#   * TRUE occlusion and disocclusion at the layer boundaries: the backward flow of frame t+1 belongs to whatever is VISIBLE there
#     (the z-buffered inverse warp), so a pixel of frame t that gets covered has F pointing at a pixel whose B points elsewhere --
#     which is what flow_check (utils.py:58-105) is there to find -- and a freshly uncovered pixel has no partner at all;
#   * spatially CORRELATED estimation error (a network's output is smooth: sub-pixel error with a correlation length of ~10 px, not
#     white noise), different in the forward, backward and stride-2 estimates;
#   * a percent or two of OUTLIER blobs where the estimate is simply wrong by a few pixels.
# Layer l has a static support in its own material coordinates (the background: everything; an object: an ellipse) and a smooth
# affine motion  x = M_l(t) u  (rotation, scale, translation).  Visible layer at pixel p of frame t = the topmost layer whose
# u = M_l(t)^-1 p lies in its support; flow t -> t+k at p = M_l(t+k) u - p.
# ------------------------------------------------------------------------------------------------------------------------------
REALISTIC = dict(n_layers=3, err_sigma=0.3, err_corr=12.0, outlier_frac=0.015, outlier_amp=3.0)


def _layer_params(rng, n_layers, H, W):
    """Per layer: centre c, radii (inf for the background), angular rate, scale rate, velocity, wobble -- smooth motions of a few
    px per frame in front of a slowly panning / zooming background."""
    L = []
    for l in range(n_layers):
        if l == 0:
            L.append(dict(c=(0.5 * W, 0.5 * H), rad=None, w=float(rng.uniform(-4e-4, 4e-4)), s=float(rng.uniform(-1e-3, 1e-3)),
                          v=(float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-0.8, 0.8))), a=(0.0, 0.0), ph=0.0))
        else:
            L.append(dict(c=(float(rng.uniform(0.25, 0.75)) * W, float(rng.uniform(0.25, 0.75)) * H),
                          rad=(float(rng.uniform(0.08, 0.18)) * W, float(rng.uniform(0.10, 0.22)) * H),
                          w=float(rng.uniform(-6e-3, 6e-3)), s=float(rng.uniform(-2e-3, 2e-3)),
                          v=(float(rng.uniform(-3.0, 3.0)), float(rng.uniform(-2.0, 2.0))),
                          a=(float(rng.uniform(0.0, 6.0)), float(rng.uniform(0.0, 4.0))), ph=float(rng.uniform(0, 2 * np.pi))))
    return L


def _layer_pose(P, t):
    """(cos, sin, scale, tx, ty) of M(t): x = c + s R (u - c) + d(t)."""
    th, sc = P["w"] * t, 1.0 + P["s"] * t
    dx = P["v"][0] * t + P["a"][0] * math.sin(0.21 * t + P["ph"])
    dy = P["v"][1] * t + P["a"][1] * math.cos(0.17 * t + P["ph"])
    return math.cos(th), math.sin(th), sc, dx, dy


def _realistic_flow(xp, xx, yy, layers, t, k):
    """True flow t -> t + k (k may be negative) of what is visible at every pixel of frame t; xp = numpy or torch."""
    u_out = v_out = None
    for P in layers:       # back to front: a later (nearer) layer overwrites
        c, s, sc, dx, dy = _layer_pose(P, t)
        cx, cy = P["c"]
        # material coordinates of the pixel: u = c + R^T (p - c - d) / scale
        px, py = xx - cx - dx, yy - cy - dy
        ux = cx + (c * px + s * py) / sc
        uy = cy + (-s * px + c * py) / sc
        c2, s2, sc2, dx2, dy2 = _layer_pose(P, t + k)
        qx, qy = ux - cx, uy - cy
        fx = cx + sc2 * (c2 * qx - s2 * qy) + dx2 - xx
        fy = cy + sc2 * (s2 * qx + c2 * qy) + dy2 - yy
        if P["rad"] is None:
            u_out, v_out = fx, fy
        else:
            inside = ((ux - cx) / P["rad"][0]) ** 2 + ((uy - cy) / P["rad"][1]) ** 2 <= 1.0
            u_out = xp.where(inside, fx, u_out)
            v_out = xp.where(inside, fy, v_out)
    return u_out, v_out


def _lowpass_noise_np(rng, H, W, corr, sigma):
    """Smooth error field: white noise on a grid of spacing `corr` px, bilinearly interpolated, two octaves, scaled to std sigma."""
    out = np.zeros((H, W, 2))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    for step, wgt in ((corr, 0.8), (2.0 * corr, 0.6)):
        gh, gw = int(H / step) + 3, int(W / step) + 3
        g = rng.standard_normal((gh, gw, 2))
        ox, oy = rng.uniform(0, step, size=2)
        fx, fy = (xx + ox) / step, (yy + oy) / step
        x0, y0 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
        ax, ay = (fx - x0)[..., None], (fy - y0)[..., None]
        out += wgt * ((1 - ay) * ((1 - ax) * g[y0, x0] + ax * g[y0, x0 + 1]) + ay * ((1 - ax) * g[y0 + 1, x0] + ax * g[y0 + 1, x0 + 1]))
    # (bilinear interpolation of unit white noise has variance 4/9 on average per octave)
    return out * (sigma / math.sqrt((0.8 ** 2 + 0.6 ** 2) * 4.0 / 9.0))


def _outliers_np(rng, field, frac, amp):
    """Elliptical blobs covering ~frac of the image where the estimate is off by a constant of a few pixels."""
    H, W = field.shape[:2]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    area, target = 0.0, frac * H * W
    while area < target:
        rx, ry = float(rng.uniform(0.01, 0.03)) * W, float(rng.uniform(0.01, 0.03)) * H
        cx, cy = float(rng.uniform(0, W)), float(rng.uniform(0, H))
        m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        field[m] += amp * rng.standard_normal(2)
        area += math.pi * rx * ry
    return field


def synth_realistic(n_frames, H, W, seed=0, stride2=True, n_layers=3, err_sigma=0.3, err_corr=12.0, outlier_frac=0.015, outlier_amp=3.0):
    """Layered scene with true (dis)occlusion and correlated flow error (see REALISTIC above): dict(flows_f, flows_b[, flows_f2,
    flows_b2]) of lists of (H,W,2) float32 arrays, bit-reproducible from the seed like synth_sequence."""
    rng = np.random.default_rng(seed)
    layers = _layer_params(rng, n_layers, H, W)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    n_pairs = n_frames - 1

    def estimate(t, k):
        u, v = _realistic_flow(np, xx, yy, layers, t, k)
        f = np.stack([u, v], -1)
        if err_sigma > 0:
            f = f + _lowpass_noise_np(rng, H, W, err_corr, err_sigma * (1.0 if abs(k) == 1 else 1.3))
        if outlier_frac > 0:
            f = _outliers_np(rng, f, outlier_frac, outlier_amp)
        return f.astype(np.float32)

    out = {"flows_f": [], "flows_b": []}
    for t in range(n_pairs):
        out["flows_f"].append(estimate(t, 1))            # frame t -> t + 1, on frame t's pixels
        out["flows_b"].append(estimate(t + 1, -1))       # frame t + 1 -> t, on frame t + 1's pixels (what is visible THERE)
    if stride2:
        out["flows_f2"], out["flows_b2"] = [], []
        for t in range(n_pairs - 1):
            out["flows_f2"].append(estimate(t, 2))
            out["flows_b2"].append(estimate(t + 2, -2))
    return out


def synth_realistic_torch(n_frames, H, W, seed=0, stride2=False, device="cuda", n_layers=3, err_sigma=0.3, err_corr=12.0,
                          outlier_frac=0.015, outlier_amp=3.0):
    """The same scene model evaluated with torch on `device` (bench.py: fills HBM without a PCIe copy; not bit-identical to the NumPy
    version -- both paths under test read the same tensors).  Returns dict of (n,H,W,2) float32 tensors."""
    import torch
    import torch.nn.functional as Fnn
    rng = np.random.default_rng(seed)
    layers = _layer_params(rng, n_layers, H, W)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    ys = torch.arange(H, device=device, dtype=torch.float64)[:, None].expand(H, W)
    xs = torch.arange(W, device=device, dtype=torch.float64)[None, :].expand(H, W)
    n_pairs = n_frames - 1

    def lowpass(sigma):
        acc = torch.zeros((1, 2, H, W), device=device, dtype=torch.float32)
        for step, wgt in ((err_corr, 0.8), (2.0 * err_corr, 0.6)):
            gh, gw = int(H / step) + 3, int(W / step) + 3
            g = torch.randn((1, 2, gh, gw), generator=gen, device=device, dtype=torch.float32)
            up = Fnn.interpolate(g, size=(int(gh * step), int(gw * step)), mode="bilinear", align_corners=False)
            oy, ox = int(rng.integers(0, int(step))), int(rng.integers(0, int(step)))
            acc += wgt * up[:, :, oy:oy + H, ox:ox + W]
        return (acc[0].permute(1, 2, 0) * (sigma / math.sqrt((0.8 ** 2 + 0.6 ** 2) * 4.0 / 9.0)))

    def outliers(field):
        area, target = 0.0, outlier_frac * H * W
        while area < target:
            rx, ry = float(rng.uniform(0.01, 0.03)) * W, float(rng.uniform(0.01, 0.03)) * H
            cx, cy = float(rng.uniform(0, W)), float(rng.uniform(0, H))
            x0, x1, y0, y1 = max(int(cx - rx), 0), min(int(cx + rx) + 1, W), max(int(cy - ry), 0), min(int(cy + ry) + 1, H)
            off = torch.tensor(outlier_amp * rng.standard_normal(2), device=device, dtype=torch.float32)
            if x1 > x0 and y1 > y0:
                m = ((xs[y0:y1, x0:x1] - cx) / rx) ** 2 + ((ys[y0:y1, x0:x1] - cy) / ry) ** 2 <= 1.0
                field[y0:y1, x0:x1] += m[..., None].to(torch.float32) * off
            area += math.pi * rx * ry

    def estimate(dst, t, k):
        u, v = _realistic_flow(torch, xs, ys, layers, t, k)
        dst[..., 0] = u.to(torch.float32)
        dst[..., 1] = v.to(torch.float32)
        if err_sigma > 0:
            dst += lowpass(err_sigma * (1.0 if abs(k) == 1 else 1.3))
        if outlier_frac > 0:
            outliers(dst)

    out = {"flows_f": torch.empty((n_pairs, H, W, 2), device=device, dtype=torch.float32),
           "flows_b": torch.empty((n_pairs, H, W, 2), device=device, dtype=torch.float32)}
    if stride2:
        out["flows_f2"] = torch.empty((max(n_pairs - 1, 0), H, W, 2), device=device, dtype=torch.float32)
        out["flows_b2"] = torch.empty((max(n_pairs - 1, 0), H, W, 2), device=device, dtype=torch.float32)
    for t in range(n_pairs):
        estimate(out["flows_f"][t], t, 1)
        estimate(out["flows_b"][t], t + 1, -1)
        if stride2 and t < n_pairs - 1:
            estimate(out["flows_f2"][t], t, 2)
            estimate(out["flows_b2"][t], t + 2, -2)
    return out
