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

#define PSFM_WAVE 64
#pragma clang fp contract(off)

struct PsfmTaps {
    int x0, y0;            // north-west tap
    float nw, ne, sw, se;  // bilinear weights
};

// cw = (float)((W-1)/2.0), ch = (float)((H-1)/2.0) computed once on the host.
__device__ __forceinline__ PsfmTaps psfm_taps(float x32, float y32, float cw, float ch, int H, int W)
{
    float gx = __fsub_rn(__fdiv_rn(x32, cw), 1.0f);
    float gy = __fsub_rn(__fdiv_rn(y32, ch), 1.0f);
    const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), cw);
    const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), ch);
    const float fx = floorf(ix), fy = floorf(iy);
    const float w = __fsub_rn(ix, fx), e = __fsub_rn(1.0f, w);
    const float n = __fsub_rn(iy, fy), s = __fsub_rn(1.0f, n);
    PsfmTaps t;
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

__device__ __forceinline__ float psfm_blend(float vnw, float vne, float vsw, float vse, const PsfmTaps& t)
{
    return __fmaf_rn(vse, t.se, __fmaf_rn(vsw, t.sw, __fmaf_rn(vne, t.ne, __fmul_rn(vnw, t.nw))));
}

// Two-channel (flow) sample from the .flo-native interleaved (H,W,2) layout: one 8-byte load per tap.
__device__ __forceinline__ float2 psfm_sample_flow(const float2* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    const int x0 = t.x0, y0 = t.y0;
    const bool xw = (x0 >= 0) & (x0 < W), xe = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool yn = (y0 >= 0) & (y0 < H), ys = (y0 + 1 >= 0) & (y0 + 1 < H);
    const float2 z = make_float2(0.0f, 0.0f);
    const int64_t rn = (int64_t)y0 * W, rs = (int64_t)(y0 + 1) * W;
    const float2 vnw = (xw & yn) ? map[rn + x0] : z;
    const float2 vne = (xe & yn) ? map[rn + x0 + 1] : z;
    const float2 vsw = (xw & ys) ? map[rs + x0] : z;
    const float2 vse = (xe & ys) ? map[rs + x0 + 1] : z;
    return make_float2(psfm_blend(vnw.x, vne.x, vsw.x, vse.x, t), psfm_blend(vnw.y, vne.y, vsw.y, vse.y, t));
}

// One-channel sample of a 0/1 byte map (the occlusion masks), values taken as 0.0f / 1.0f.
__device__ __forceinline__ float psfm_sample_mask(const uint8_t* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    const int x0 = t.x0, y0 = t.y0;
    const bool xw = (x0 >= 0) & (x0 < W), xe = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool yn = (y0 >= 0) & (y0 < H), ys = (y0 + 1 >= 0) & (y0 + 1 < H);
    const int64_t rn = (int64_t)y0 * W, rs = (int64_t)(y0 + 1) * W;
    const float vnw = (xw & yn) ? (map[rn + x0] ? 1.0f : 0.0f) : 0.0f;
    const float vne = (xe & yn) ? (map[rn + x0 + 1] ? 1.0f : 0.0f) : 0.0f;
    const float vsw = (xw & ys) ? (map[rs + x0] ? 1.0f : 0.0f) : 0.0f;
    const float vse = (xe & ys) ? (map[rs + x0 + 1] ? 1.0f : 0.0f) : 0.0f;
    return psfm_blend(vnw, vne, vsw, vse, t);
}

// One-channel f32 sample (API parity for psfm_grid_sample with C == 1).
__device__ __forceinline__ float psfm_sample_f32(const float* __restrict__ map, int H, int W, const PsfmTaps& t)
{
    const int x0 = t.x0, y0 = t.y0;
    const bool xw = (x0 >= 0) & (x0 < W), xe = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool yn = (y0 >= 0) & (y0 < H), ys = (y0 + 1 >= 0) & (y0 + 1 < H);
    const int64_t rn = (int64_t)y0 * W, rs = (int64_t)(y0 + 1) * W;
    const float vnw = (xw & yn) ? map[rn + x0] : 0.0f;
    const float vne = (xe & yn) ? map[rn + x0 + 1] : 0.0f;
    const float vsw = (xw & ys) ? map[rs + x0] : 0.0f;
    const float vse = (xe & ys) ? map[rs + x0 + 1] : 0.0f;
    return psfm_blend(vnw, vne, vsw, vse, t);
}

// ---- wavefront (64-lane) helpers ----------------------------------------------------------
__device__ __forceinline__ int psfm_lane_id() { return (int)__lane_id(); }

// rank of this lane among the set lanes of a 64-bit ballot
__device__ __forceinline__ int psfm_rank_in(unsigned long long mask)
{
    return __popcll(mask & ((1ull << psfm_lane_id()) - 1ull));
}
