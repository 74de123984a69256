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

