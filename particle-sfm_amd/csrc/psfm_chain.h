// psfm_chain.h -- device building blocks of one chain step, shared by the per-frame kernel (psfm_track.hip) and the
// persistent frame loop (psfm_persist.hip).  `A` is any struct with the fields the function names use
// (flow, occ, H, W, cw, ch, rcw, rch / ratio, rdiv, GW, GH, blocked_cur, stamp_cur).
#pragma once
#include "psfm_device.h"

// ---- accesses that must be seen by blocks on OTHER XCDs inside one kernel: each XCD has its own L2, so plain
// stores sit in the writer's L2 until the kernel ends.  System-scope relaxed atomics compile to write-through
// stores / L2-bypassing loads (sc0 sc1) ----
template <class T> __device__ __forceinline__ T psfm_coh_ld(const T* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class T> __device__ __forceinline__ void psfm_coh_st(T* p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <bool COH> __device__ __forceinline__ void psfm_mark(uint8_t* p, uint8_t v)
{
    if (COH) psfm_coh_st(p, v); else *p = v;
}

// ---- one pixel of flow_check (utils.py:58-105): forward flow f at pixel (x,y), backward field B ----
// NEED_ERR: the caller wants the error map (reference API flow_check()): true division and the square root.
// Otherwise only the mask: fast exact division (psfm_div_r) and the threshold moved under the root --
// sqrtf is correctly rounded and monotonic, so sqrtf(s) > thres  <=>  s > t2 with t2 = max{s : sqrtf(s) <= thres}
// (psfm_sq_threshold() on the host).
struct PsfmFcParams { int H, W; float cw, ch, rcw, rch, thres, t2; };
static inline float psfm_sq_threshold(float thres)
{
    if (thres != thres) return INFINITY;              // e > NaN is never true
    if (thres < 0.0f) return -1.0f;                   // e >= 0 > thres for every non-NaN e
    if (thres == INFINITY) return INFINITY;
    float t2 = thres * thres;
    if (t2 == INFINITY) t2 = 3.4028234663852886e38f;
    while (sqrtf(t2) > thres) t2 = nextafterf(t2, -INFINITY);
    while (t2 < 3.4028234663852886e38f && sqrtf(nextafterf(t2, INFINITY)) <= thres) t2 = nextafterf(t2, INFINITY);
    return t2;
}
template <bool NEED_ERR>
__device__ __forceinline__ uint8_t psfm_flow_check_px(const float2* __restrict__ B, int x, int y, float2 f, const PsfmFcParams& q,
                                                      float* err)
{
    // utils.py:73-78: pixel coordinate + flow in fp32
    const float X = __fadd_rn((float)x, f.x), Y = __fadd_rn((float)y, f.y);
    const PsfmTaps t = psfm_taps_t<!NEED_ERR>(X, Y, q.cw, q.ch, q.rcw, q.rch, q.H, q.W);           // utils.py:79-82
    const float2 b = psfm_sample_flow(B, q.H, q.W, t);
    // utils.py:87: torch.norm(warp + flow, dim=1) == sqrtf(fma(ev,ev, eu*eu)); sqrtf is correctly rounded
    const float eu = __fadd_rn(b.x, f.x), ev = __fadd_rn(b.y, f.y);
    const float s2 = __fmaf_rn(ev, ev, __fmul_rn(eu, eu));
    // utils.py:58-68 (oob) and :88-91 (union)
    const bool oob = (X < 0.0f) | (X > (float)(q.W - 1)) | (Y < 0.0f) | (Y > (float)(q.H - 1));
    if (NEED_ERR) {
        const float e = sqrtf(s2);
        *err = e;
        return (uint8_t)((e > q.thres) | oob);
    }
    return (uint8_t)((s2 > q.t2) | oob);
}

#ifndef PSFM_FC_INTERIOR_PAIRS
#define PSFM_FC_INTERIOR_PAIRS 0   // 1: the two taps of a row as one 16-byte load (36 spilled VGPRs in the persistent loop's 64)
#endif
// The same verdict for a pixel whose four taps are known to lie inside the map (the caller tests that for the whole wave):
// the two taps of a row are ONE 16-byte load, nothing is clamped or zero-padded.  Same blend, same bits.
__device__ __forceinline__ uint8_t psfm_flow_check_px_interior(const float2* __restrict__ B, float X, float Y, float2 f, const PsfmTaps& t,
                                                               const PsfmFcParams& q)
{
    const unsigned o = (unsigned)(t.y0 * q.W + t.x0) * 8u, o2 = o + (unsigned)q.W * 8u;
#if PSFM_FC_INTERIOR_PAIRS
    const float4 n4 = *(const float4*)((const char*)B + o);
    const float4 s4 = *(const float4*)((const char*)B + o2);
    const float bx = psfm_blend(n4.x, n4.z, s4.x, s4.z, t), by = psfm_blend(n4.y, n4.w, s4.y, s4.w, t);
#else
    const float2 nw = psfm_ld(B, o), ne = psfm_ld(B, o + 8u), sw = psfm_ld(B, o2), se = psfm_ld(B, o2 + 8u);
    const float bx = psfm_blend(nw.x, ne.x, sw.x, se.x, t), by = psfm_blend(nw.y, ne.y, sw.y, se.y, t);
#endif
    const float eu = __fadd_rn(bx, f.x), ev = __fadd_rn(by, f.y);
    const float s2 = __fmaf_rn(ev, ev, __fmul_rn(eu, eu));
    const bool oob = (X < 0.0f) | (X > (float)(q.W - 1)) | (Y < 0.0f) | (Y > (float)(q.H - 1));
    return (uint8_t)((s2 > q.t2) | oob);
}

struct PsfmStep { bool alive; double2 next; float2 flow; };   // flow: the sampled (fx, fy), next = p + (double)flow

// One chain step split in two so that the caller can issue the gathers of several steps back to back and keep
// them in flight across the block's bookkeeping: psfm_step_issue() computes the tap geometry and performs the eight
// raw loads (4 flow taps, 4 mask taps); psfm_step_finish() REBUILDS the geometry from the position (ALU only: holding
// weights and flags across the barriers costs the registers that decide 7 vs 8 waves per SIMD), blends the taps
// (fp32, bit-exact op order) and applies trajectory.py:50,55-57.
struct PsfmStepLoads {
    float2 fnw, fne, fsw, fse;
    unsigned onw, one, osw, ose;    // mask bytes, zero-extended
};

// Pins the raw tap registers at this program point: nothing computed FROM the loads can be scheduled above it, so
// the s_waitcnt for the gathers cannot drift in front of the bookkeeping that is meant to overlap their latency
// (the compiler otherwise packed the mask bytes right behind the loads, i.e. before the block's atomics).
template <bool MASKS>
__device__ __forceinline__ void psfm_step_pin(PsfmStepLoads& L)
{
    asm volatile("" : "+v"(L.fnw.x), "+v"(L.fnw.y), "+v"(L.fne.x), "+v"(L.fne.y), "+v"(L.fsw.x), "+v"(L.fsw.y),
                      "+v"(L.fse.x), "+v"(L.fse.y));
    // (the zero-extension of a mask byte must sit in the basic block of its load to fold into global_load_ubyte;
    // pinning the lanes' bytes, loaded two blocks earlier, would materialise it -- and a wait -- before the barrier)
    if (MASKS) asm volatile("" : "+v"(L.onw), "+v"(L.one), "+v"(L.osw), "+v"(L.ose));
}

template <class A>
__device__ __forceinline__ PsfmStepLoads psfm_step_issue(const A& a, double2 p)
{
    const PsfmTaps t = psfm_taps_fast((float)p.x, (float)p.y, a.cw, a.ch, a.rcw, a.rch, a.H, a.W);
    const PsfmTapIdx k = psfm_tap_idx(a.H, a.W, t);   // flow and occlusion map share the tap geometry
    PsfmStepLoads L;
    const unsigned onw = (unsigned)k.nw, one = (unsigned)k.ne, osw = (unsigned)k.sw, ose = (unsigned)k.se;
    L.fnw = psfm_ld(a.flow, onw * 8u); L.fne = psfm_ld(a.flow, one * 8u);
    L.fsw = psfm_ld(a.flow, osw * 8u); L.fse = psfm_ld(a.flow, ose * 8u);
    L.onw = psfm_ld(a.occ, onw); L.one = psfm_ld(a.occ, one); L.osw = psfm_ld(a.occ, osw); L.ose = psfm_ld(a.occ, ose);
    return L;
}

template <class A>
__device__ __forceinline__ PsfmStep psfm_step_finish(const A& a, double2 p, const PsfmStepLoads& L)
{
    const PsfmTaps t = psfm_taps_fast((float)p.x, (float)p.y, a.cw, a.ch, a.rcw, a.rch, a.H, a.W);
    const int x0 = t.x0, y0 = t.y0, x1 = t.x0 + 1, y1 = t.y0 + 1;
    const bool xw = (x0 >= 0) & (x0 < a.W), xe = (x1 >= 0) & (x1 < a.W);
    const bool yn = (y0 >= 0) & (y0 < a.H), ys = (y1 >= 0) & (y1 < a.H);
    const bool inw = xw & yn, ine = xe & yn, isw = xw & ys, ise = xe & ys;
    const float z = 0.0f;
    const float fx = psfm_blend(inw ? L.fnw.x : z, ine ? L.fne.x : z, isw ? L.fsw.x : z, ise ? L.fse.x : z, t);
    const float fy = psfm_blend(inw ? L.fnw.y : z, ine ? L.fne.y : z, isw ? L.fsw.y : z, ise ? L.fse.y : z, t);
    const float oc = psfm_blend((inw & (L.onw != 0)) ? 1.0f : z, (ine & (L.one != 0)) ? 1.0f : z,
                                (isw & (L.osw != 0)) ? 1.0f : z, (ise & (L.ose != 0)) ? 1.0f : z, t);
    const double nx = p.x + (double)fx, ny = p.y + (double)fy;
    const bool valid = (nx > 0.0) & (nx < (double)(a.W - 1)) & (ny > 0.0) & (ny < (double)(a.H - 1));
    PsfmStep s;
    s.next = make_double2(nx, ny);
    s.flow = make_float2(fx, fy);
    s.alive = valid & !(oc > 0.1f);
    return s;
}

// occupied_map[int(y), int(x)] = 1 (trajectory.py:144) folded with the EDT test: mark every stride-r grid
// point whose disc of radius r contains this pixel.  Candidates are the 3x3 grid cells around the pixel's cell;
// a cell on the low side can only qualify when the pixel sits exactly on that grid line.  R > 0: compile-time ratio.
template <int R, bool COH = false, class A>
__device__ __forceinline__ void psfm_block_grid(const A& a, int px, int py)
{
    const int r = R > 0 ? R : a.ratio, r2 = r * r;
    int qx, qy;
    if (R == 1) { qx = px; qy = py; }
    else if (R == 2) { qx = px >> 1; qy = py >> 1; }
    else if (R == 4) { qx = px >> 2; qy = py >> 2; }
    else { qx = (int)psfm_fastdiv((unsigned)px, a.rdiv); qy = (int)psfm_fastdiv((unsigned)py, a.rdiv); }
    const int ax = px - qx * r, ay = py - qy * r;
    uint8_t* row = a.blocked_cur + qy * a.GW + qx;
    const uint8_t st = a.stamp_cur;
    // dx for column offsets -1, 0, +1 : -(ax + r) [only when ax == 0], -ax, r - ax
    const int dx0 = ax, dx1 = r - ax, dy0 = ay, dy1 = r - ay;
    const bool xl = (ax == 0) & (qx > 0), xr = qx + 1 < a.GW;
    const bool yl = (ay == 0) & (qy > 0), yr = qy + 1 < a.GH;
    // centre row (dy = -ay)
    if (dx0 * dx0 + dy0 * dy0 <= r2) psfm_mark<COH>(row, st);
    if (xr & (dx1 * dx1 + dy0 * dy0 <= r2)) psfm_mark<COH>(row + 1, st);
    if (xl & (r2 + dy0 * dy0 <= r2)) psfm_mark<COH>(row - 1, st);
    // row below (dy = r - ay)
    if (yr) {
        uint8_t* rb = row + a.GW;
        if (dx0 * dx0 + dy1 * dy1 <= r2) psfm_mark<COH>(rb, st);
        if (xr & (dx1 * dx1 + dy1 * dy1 <= r2)) psfm_mark<COH>(rb + 1, st);
        if (xl & (r2 + dy1 * dy1 <= r2)) psfm_mark<COH>(rb - 1, st);
    }
    // row above (dy = -r, only when the pixel is on the grid line)
    if (yl) {
        uint8_t* ra = row - a.GW;
        if (dx0 * dx0 + r2 <= r2) psfm_mark<COH>(ra, st);
        // the diagonal neighbours are at distance >= r*sqrt(2) > r unless dx == 0, which is the line above
    }
}

__device__ __forceinline__ unsigned long long psfm_key(int last_time, int bf, int idx, int shift_b, int shift_d)
{
    return ((unsigned long long)last_time << shift_d) | ((unsigned long long)bf << shift_b) | (unsigned long long)idx;
}

