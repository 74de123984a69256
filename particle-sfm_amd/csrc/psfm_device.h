// psfm_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// The fp32 sampler reproduces, bit for bit, what the reference executes on the CPU:
// point_trajectory/trajectory.py:25-37 -> torch F.grid_sample (bilinear, zeros padding,
// align_corners=True).  Op order: positions are rounded to fp32,
// normalised with a TRUE division by the fp32 scalar (size-1)/2, shifted by -1, un-normalised
// as (g+1)*((size-1)/2), split into floor + fraction, and the four taps are blended as
//   fma(v_se,se, fma(v_sw,sw, fma(v_ne,ne, v_nw*nw))).
// Every operation is spelled with a round-to-nearest intrinsic so the result does not depend
// on the compiler's contraction / fast-math settings (the TU is built -ffp-contract=off too).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define PSFM_WAVE 64
#pragma clang fp contract(off)

struct PsfmTaps {
    int x0, y0;            // north-west tap
    float nw, ne, sw, se;  // bilinear weights
    float fw, fn;          // the fractions they are built from (west->east, north->south)
};

// weights from the two fractions: s*e, s*w, n*e, n*w with e = 1-w, s = 1-n (same op order as psfm_taps)
__device__ __forceinline__ PsfmTaps psfm_weights(float w, float n)
{
    const float e = __fsub_rn(1.0f, w), s = __fsub_rn(1.0f, n);
    PsfmTaps t;
    t.x0 = 0; t.y0 = 0; t.fw = w; t.fn = n;
    t.nw = __fmul_rn(s, e);
    t.ne = __fmul_rn(s, w);
    t.sw = __fmul_rn(n, e);
    t.se = __fmul_rn(n, w);
    return t;
}

// ---- x / c for the launch-invariant divisor c = (size-1)/2, correctly rounded like the true division ----
// r = the refined reciprocal of c (psfm_rcp_host() below, a kernel argument).  q = x*r; two residual corrections
// e = fma(-c,q,x), q = fma(e,r,q): the core of the compiler's own IEEE expansion of `/` (v_div_scale / v_div_fmas /
// v_div_fixup only add the scaling of extreme exponents and the inf/NaN/zero fix-ups), 5 VALU ops instead of 11.
// Bit-identical to x / c for 1e-30 < |x| < 1e30 and x = +0 (1.2e9 random and adversarial operands over every
// c = (W-1)/2, W <= 8192, with r perturbed by +-1 ulp: the enumeration test run by tests/test_abi_and_host.py).  Outside that range (no pixel coordinate
// is) the quotient may differ (overflow -> NaN, -0 -> +0); both still sample "all taps out of bounds".  Callers that
// must reproduce the reference's ERROR MAP for absurd flows use psfm_taps() with the true division.
__device__ __forceinline__ float psfm_div_r(float x, float c, float r)
{
    float q = __fmul_rn(x, r);
    float e = __fmaf_rn(-c, q, x);
    q = __fmaf_rn(e, r, q);
    e = __fmaf_rn(-c, q, x);
    return __fmaf_rn(e, r, q);
}
static inline float psfm_rcp_host(float c)   // host: refined reciprocal for psfm_div_r
{
    const float r0 = (float)(1.0 / (double)c);
    const float e = fmaf(-c, r0, 1.0f);
    return fmaf(e, r0, r0);
}

// cw = (float)((W-1)/2.0), ch = (float)((H-1)/2.0) computed once on the host.
template <bool FAST>
__device__ __forceinline__ PsfmTaps psfm_taps_t(float x32, float y32, float cw, float ch, float rcw, float rch, int H, int W)
{
    float gx = __fsub_rn(FAST ? psfm_div_r(x32, cw, rcw) : __fdiv_rn(x32, cw), 1.0f);
    float gy = __fsub_rn(FAST ? psfm_div_r(y32, ch, rch) : __fdiv_rn(y32, ch), 1.0f);
    const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), cw);
    const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), ch);
    const float fx = floorf(ix), fy = floorf(iy);
    const float w = __fsub_rn(ix, fx), e = __fsub_rn(1.0f, w);
    const float n = __fsub_rn(iy, fy), s = __fsub_rn(1.0f, n);
    PsfmTaps t;
    t.fw = w; t.fn = n;
    t.nw = __fmul_rn(s, e);
    t.ne = __fmul_rn(s, w);
    t.sw = __fmul_rn(n, e);
    t.se = __fmul_rn(n, w);
    // keep the int conversion defined for absurd / NaN coordinates: anything outside
    // [-1, size] has all four taps out of bounds anyway
    float cx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    float cy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    if (!(fx == fx)) cx = -2.0f;
    if (!(fy == fy)) cy = -2.0f;
    t.x0 = (int)cx;
    t.y0 = (int)cy;
    return t;
}
__device__ __forceinline__ PsfmTaps psfm_taps(float x32, float y32, float cw, float ch, int H, int W)
{
    return psfm_taps_t<false>(x32, y32, cw, ch, 0.f, 0.f, H, W);
}
__device__ __forceinline__ PsfmTaps psfm_taps_fast(float x32, float y32, float cw, float ch, float rcw, float rch, int H, int W)
{
    return psfm_taps_t<true>(x32, y32, cw, ch, rcw, rch, H, W);
}

__device__ __forceinline__ float psfm_blend(float vnw, float vne, float vsw, float vse, const PsfmTaps& t)
{
    return __fmaf_rn(vse, t.se, __fmaf_rn(vsw, t.sw, __fmaf_rn(vne, t.ne, __fmul_rn(vnw, t.nw))));
}

// ---- base + 32-bit unsigned BYTE offset addressing: lets the compiler use the SGPR-base + VGPR-offset form of
// global_load / global_store (no 64-bit VALU address arithmetic per access); every map/table here is < 4 GB ----
template <class T>
__device__ __forceinline__ T psfm_ld(const T* __restrict__ base, unsigned byte_off)
{
    return *(const T*)((const char*)base + byte_off);
}
template <class T>
__device__ __forceinline__ void psfm_st(T* __restrict__ base, unsigned byte_off, T v)
{
    *(T*)((char*)base + byte_off) = v;
}

// Tap addressing: every tap is loaded UNCONDITIONALLY from a clamped (always valid) address and zeroed by a
// select afterwards (zeros padding).  Branch-free loads let the compiler issue all taps back to back behind a
// single s_waitcnt; predicated loads compiled to one exec-masked branch + wait per tap (8 serial round trips).
struct PsfmTapIdx {
    int nw, ne, sw, se;          // element offsets of the four taps (clamped into the map; H*W < 2^31)
    bool inw, ine, isw, ise;     // tap inside the map?
};

__device__ __forceinline__ PsfmTapIdx psfm_tap_idx(int H, int W, const PsfmTaps& t)
{
    const int x0 = t.x0, y0 = t.y0, x1 = t.x0 + 1, y1 = t.y0 + 1;
    const bool xw = (x0 >= 0) & (x0 < W), xe = (x1 >= 0) & (x1 < W);
    const bool yn = (y0 >= 0) & (y0 < H), ys = (y1 >= 0) & (y1 < H);
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x1, 0), W - 1);
    const int rn = min(max(y0, 0), H - 1) * W, rs = min(max(y1, 0), H - 1) * W;
    PsfmTapIdx k;
    k.nw = rn + xc0; k.ne = rn + xc1; k.sw = rs + xc0; k.se = rs + xc1;
    k.inw = xw & yn; k.ine = xe & yn; k.isw = xw & ys; k.ise = xe & ys;
    return k;
}

// Two-channel (flow) sample from the .flo-native interleaved (H,W,2) layout: one 8-byte load per tap.
__device__ __forceinline__ float2 psfm_sample_flow(const float2* __restrict__ map, const PsfmTapIdx& k, const PsfmTaps& t)
{
    float2 vnw = psfm_ld(map, (unsigned)k.nw * 8u), vne = psfm_ld(map, (unsigned)k.ne * 8u);
    float2 vsw = psfm_ld(map, (unsigned)k.sw * 8u), vse = psfm_ld(map, (unsigned)k.se * 8u);
    vnw.x = k.inw ? vnw.x : 0.0f; vnw.y = k.inw ? vnw.y : 0.0f;
    vne.x = k.ine ? vne.x : 0.0f; vne.y = k.ine ? vne.y : 0.0f;
    vsw.x = k.isw ? vsw.x : 0.0f; vsw.y = k.isw ? vsw.y : 0.0f;
    vse.x = k.ise ? vse.x : 0.0f; vse.y = k.ise ? vse.y : 0.0f;
    return make_float2(psfm_blend(vnw.x, vne.x, vsw.x, vse.x, t), psfm_blend(vnw.y, vne.y, vsw.y, vse.y, t));
}
__device__ __forceinline__ float2 psfm_sample_flow(const float2* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    return psfm_sample_flow(map, psfm_tap_idx(H, W, t), t);
}

// One-channel sample of a 0/1 byte map (the occlusion masks), values taken as 0.0f / 1.0f.
__device__ __forceinline__ float psfm_sample_mask(const uint8_t* __restrict__ map, const PsfmTapIdx& k, const PsfmTaps& t)
{
    const uint8_t bnw = psfm_ld(map, (unsigned)k.nw), bne = psfm_ld(map, (unsigned)k.ne);
    const uint8_t bsw = psfm_ld(map, (unsigned)k.sw), bse = psfm_ld(map, (unsigned)k.se);
    const float vnw = (k.inw & (bnw != 0)) ? 1.0f : 0.0f;
    const float vne = (k.ine & (bne != 0)) ? 1.0f : 0.0f;
    const float vsw = (k.isw & (bsw != 0)) ? 1.0f : 0.0f;
    const float vse = (k.ise & (bse != 0)) ? 1.0f : 0.0f;
    return psfm_blend(vnw, vne, vsw, vse, t);
}
__device__ __forceinline__ float psfm_sample_mask(const uint8_t* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    return psfm_sample_mask(map, psfm_tap_idx(H, W, t), t);
}

// One-channel f32 sample (API parity for psfm_grid_sample with C == 1).
__device__ __forceinline__ float psfm_sample_f32(const float* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    const PsfmTapIdx k = psfm_tap_idx(H, W, t);
    float vnw = map[k.nw], vne = map[k.ne], vsw = map[k.sw], vse = map[k.se];
    vnw = k.inw ? vnw : 0.0f; vne = k.ine ? vne : 0.0f; vsw = k.isw ? vsw : 0.0f; vse = k.ise ? vse : 0.0f;
    return psfm_blend(vnw, vne, vsw, vse, t);
}

// ---- division by a launch-invariant divisor (Granlund-Montgomery): q = (t + ((n - t) >> sh1)) >> sh2 ----
struct PsfmFastDiv { unsigned m; int sh1, sh2; unsigned d; };
static inline PsfmFastDiv psfm_fastdiv_make(unsigned d)
{
    PsfmFastDiv f; f.d = d;
    int l = 0; while ((1ull << l) < d) ++l;
    f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1; f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ unsigned psfm_fastdiv(unsigned n, const PsfmFastDiv& f)
{
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// ---- wavefront (64-lane) helpers ----------------------------------------------------------
__device__ __forceinline__ int psfm_lane_id() { return (int)__lane_id(); }

// rank of this lane among the set lanes of a 64-bit ballot
__device__ __forceinline__ int psfm_rank_in(unsigned long long mask)
{
    return __popcll(mask & ((1ull << psfm_lane_id()) - 1ull));
}
