// psfm_track.hip -- gfx950 kernels of the trajectory builder:
//   K1 flow_check   (point_trajectory/utils.py:58-105)
//   K2 chain_step   (point_trajectory/trajectory.py:25-37,45-62 + track.py:38-46 + extend_all :129-147)
//   K3 respawn      (point_trajectory/trajectory.py:150-152 + new_traj_all :117-120)
// plus the sampler exposed for API parity (trajectory.py:25-37).
//
// Data layout in HBM:
//   flows    (n,H,W) float2     .flo-native interleaved (u,v): one 8-byte load per bilinear tap
//   occ      (n,H,W) u8         0/1
//   log      (n_flows+1, cap) double2   position of the track living in lane L at time t.  The log IS
//                                the per-track state: chain_step reads slab t and writes slab t+1, the
//                                solver rewrites slabs t, t+1 in place; nothing is compacted or copied.
//   lanes    birth_frame[cap] (-1 = free), birth_idx[cap]; free lanes are recycled through an atomic
//            stack, finished trajectories are recorded as (sort key, lane) pairs.  Physical lane order
//            is free: ids derive from the key (death_step, birth_frame, birth_grid_index) at finalize
//            (SURVEY.md a-17), so no stable compaction is ever needed.
#include "psfm_device.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "psfm_internal.h"
#include "psfm_chain.h"

#define PSFM_BLOCK 256

// ------------------------------------------------------------------------------------------------
// K1  flow_check (utils.py:58-105).  Algorithmic bytes per pair: 8P (F, streamed) + 8P (B, gathered
// near p+F) + P (occ) = 17P.  blockIdx.y = frame pair (one launch for all pairs).  Three kernels: one pixel per thread
// (tiny maps), four pixels per thread strided by the block (8-byte loads; keeps the reference API's error map), and two
// ADJACENT pixels per lane with 16-byte loads (psfm_flow_check_x2v_kernel, the mask-only default).
// ------------------------------------------------------------------------------------------------
template <bool NEED_ERR>
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, PsfmFcParams q,
    uint8_t* __restrict__ occ_out, float* __restrict__ err_out)
{
    const int64_t P = (int64_t)q.H * q.W;
    const int64_t p = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (p >= P) return;
    const int64_t base = (int64_t)blockIdx.y * P;
    const int y = (int)(p / q.W), x = (int)(p - (int64_t)y * q.W);
    float e = 0.f;
    occ_out[base + p] = psfm_flow_check_px<NEED_ERR>(flows_b + base, x, y, flows_f[base + p], q, &e);
    if (NEED_ERR) err_out[base + p] = e;
}

// 4 pixels per thread, strided by the block size: every load/store instruction stays fully coalesced
// (64 lanes x 8 B contiguous) while each wave keeps 4x more bytes in flight.
#define PSFM_FC_UNROLL 4
template <bool NEED_ERR>
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_x4_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, PsfmFcParams q,
    uint8_t* __restrict__ occ_out, float* __restrict__ err_out, PsfmFastDiv wdiv)
{
    const int P = q.H * q.W;
    const int p0 = blockIdx.x * (PSFM_BLOCK * PSFM_FC_UNROLL) + threadIdx.x;
    const int64_t base = (int64_t)blockIdx.y * P;
    const float2* __restrict__ F = flows_f + base;
    const float2* __restrict__ B = flows_b + base;
    float2 f[PSFM_FC_UNROLL];
#pragma unroll
    for (int k = 0; k < PSFM_FC_UNROLL; ++k) {
        const int p = p0 + k * PSFM_BLOCK;
        f[k] = p < P ? psfm_ld(F, (unsigned)p * 8u) : make_float2(0.f, 0.f);
    }
    int y = (int)psfm_fastdiv((unsigned)p0, wdiv), x = p0 - y * q.W;
#pragma unroll
    for (int k = 0; k < PSFM_FC_UNROLL; ++k) {
        const int p = p0 + k * PSFM_BLOCK;
        if (p >= P) break;
        float e = 0.f;
        occ_out[base + p] = psfm_flow_check_px<NEED_ERR>(B, x, y, f[k], q, &e);
        if (NEED_ERR) err_out[base + p] = e;
        x += PSFM_BLOCK;
        while (x >= q.W) { x -= q.W; ++y; }
    }
}


// Two ADJACENT pixels per lane and step: F arrives as one 16-byte load per lane (the access width the memory system is
// fastest at), the two horizontally adjacent taps of a row of B as one 16-byte load (2 loads per pixel instead of 4), the
// mask leaves as one 2-byte store.  PSFM_FC2_UNROLL pairs per thread, block-strided like the x4 kernel.
#define PSFM_FC2_UNROLL 2
struct PsfmFcGeom { PsfmTaps t; float X, Y; bool interior; };
__device__ __forceinline__ PsfmFcGeom psfm_fc_geom(int x, int y, float2 f, const PsfmFcParams& q)
{
    PsfmFcGeom g;
    g.X = __fadd_rn((float)x, f.x); g.Y = __fadd_rn((float)y, f.y);
    g.t = psfm_taps_t<true>(g.X, g.Y, q.cw, q.ch, q.rcw, q.rch, q.H, q.W);
    // all four taps inside the map: the pair load needs no clamping and no tap is zero-padded
    g.interior = (g.t.x0 >= 0) & (g.t.x0 + 1 < q.W) & (g.t.y0 >= 0) & (g.t.y0 + 1 < q.H);
    return g;
}
__device__ __forceinline__ uint8_t psfm_fc_verdict(float bx, float by, float2 f, const PsfmFcGeom& g, const PsfmFcParams& q)
{
    const float eu = __fadd_rn(bx, f.x), ev = __fadd_rn(by, f.y);
    const float s2 = __fmaf_rn(ev, ev, __fmul_rn(eu, eu));
    const bool oob = (g.X < 0.0f) | (g.X > (float)(q.W - 1)) | (g.Y < 0.0f) | (g.Y > (float)(q.H - 1));
    return (uint8_t)((s2 > q.t2) | oob);
}
// INTERIOR (wave-uniform: every lane's taps lie inside the map -- all but the border waves): the west / east taps are the two
// halves of the pair load, nothing is selected.  Same blend, same bits.
template <bool INTERIOR>
__device__ __forceinline__ uint8_t psfm_flow_check_px16(const float2* __restrict__ B, float2 f, const PsfmFcGeom& g, const PsfmFcParams& q)
{
    const PsfmTaps& t = g.t;
    int x0 = t.x0, y0 = t.y0;
    // (the border form's clamps and selects are pure arithmetic: without this fence the compiler computes them ahead of the
    // wave-uniform branch, for every wave)
    if (!INTERIOR) asm volatile("" : "+v"(x0), "+v"(y0));
    if (INTERIOR) {
        const unsigned o = (unsigned)(y0 * q.W + x0) * 8u;
        const float4 n4 = *(const float4*)((const char*)B + o);
        const float4 s4 = *(const float4*)((const char*)B + o + (unsigned)q.W * 8u);
        return psfm_fc_verdict(psfm_blend(n4.x, n4.z, s4.x, s4.z, t), psfm_blend(n4.y, n4.w, s4.y, s4.w, t), f, g, q);
    }
    const int xc = min(max(x0, 0), q.W - 2);                         // the pair (xc, xc + 1) lies inside the row
    const int rn = min(max(y0, 0), q.H - 1), rs = min(max(y0 + 1, 0), q.H - 1);
    const float4 n4 = *(const float4*)((const char*)B + ((unsigned)(rn * q.W + xc)) * 8u);
    const float4 s4 = *(const float4*)((const char*)B + ((unsigned)(rs * q.W + xc)) * 8u);
    const bool xw = (x0 >= 0) & (x0 < q.W), xe = (x0 + 1 >= 0) & (x0 + 1 < q.W);
    const bool yn = (y0 >= 0) & (y0 < q.H), ys = (y0 + 1 >= 0) & (y0 + 1 < q.H);
    const bool wlo = x0 == xc, elo = x0 + 1 == xc;                   // which half of the pair a tap is
    const float z = 0.0f;
    const float nwx = (xw & yn) ? (wlo ? n4.x : n4.z) : z, nwy = (xw & yn) ? (wlo ? n4.y : n4.w) : z;
    const float nex = (xe & yn) ? (elo ? n4.x : n4.z) : z, ney = (xe & yn) ? (elo ? n4.y : n4.w) : z;
    const float swx = (xw & ys) ? (wlo ? s4.x : s4.z) : z, swy = (xw & ys) ? (wlo ? s4.y : s4.w) : z;
    const float sex = (xe & ys) ? (elo ? s4.x : s4.z) : z, sey = (xe & ys) ? (elo ? s4.y : s4.w) : z;
    return psfm_fc_verdict(psfm_blend(nwx, nex, swx, sex, t), psfm_blend(nwy, ney, swy, sey, t), f, g, q);
}

// XCD-aware block -> pixel mapping (xcd_per > 0): workgroups go to the eight XCDs round-robin by linear id, and the XCDs'
// L2s are private.  With blocks walking the map in id order, the rows y and y + 1 of pixels -- whose taps share a row of
// B -- are handled by different XCDs, so every row of B crosses the fabric twice (rocprofv3 FETCH_SIZE: 1.6x the
// algorithmic reads).  Here the launch has 8 * xcd_per blocks per pair and block x takes chunk (x % 8) * xcd_per + x / 8:
// every XCD streams ONE contiguous band of H / 8 rows, top to bottom, and B's rows are re-read from its own L2.
__device__ __forceinline__ int psfm_xcd_chunk(int bx, int xcd_per) { return xcd_per > 0 ? (bx & 7) * xcd_per + (bx >> 3) : bx; }
static inline int psfm_xcd_per(int64_t chunks)
{
    static const int on = getenv("PSFM_FC_XCD") ? atoi(getenv("PSFM_FC_XCD")) : 1;
    return on && chunks >= 64 ? (int)((chunks + 7) / 8) : 0;
}

template <bool NT>     // NT: non-temporal F loads / mask stores (streamed once: keep them out of the way of the B taps in L2)
__device__ __forceinline__ void psfm_flow_check_x2v_body(const float2* __restrict__ F, const float2* __restrict__ B, const PsfmFcParams& q,
                                                         uint8_t* __restrict__ occ_pair, const PsfmFastDiv& wdiv, int xcd_per)
{
    const int P = q.H * q.W;                      // (even: the launcher falls back to the x4 kernel otherwise)
    const int bx = psfm_xcd_chunk((int)blockIdx.x, xcd_per);
    if ((int64_t)bx * (PSFM_BLOCK * PSFM_FC2_UNROLL * 2) >= P) return;
    const int p0 = (bx * (PSFM_BLOCK * PSFM_FC2_UNROLL) + threadIdx.x) * 2;
    float4 f[PSFM_FC2_UNROLL];
#pragma unroll
    for (int k = 0; k < PSFM_FC2_UNROLL; ++k) {
        const int p = p0 + k * PSFM_BLOCK * 2;
        typedef float psfm_v4f __attribute__((ext_vector_type(4)));
        const float4* src = (const float4*)((const char*)F + (unsigned)p * 8u);
        if (p >= P) f[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (NT) { const psfm_v4f v = __builtin_nontemporal_load((const psfm_v4f*)src); f[k] = make_float4(v.x, v.y, v.z, v.w); }
        else f[k] = *src;
    }
#pragma unroll
    for (int k = 0; k < PSFM_FC2_UNROLL; ++k) {
        const int p = p0 + k * PSFM_BLOCK * 2;
        if (p >= P) break;
        const int y = (int)psfm_fastdiv((unsigned)p, wdiv), x = p - y * q.W;
        int x1 = x + 1, y1 = y;
        if (x1 >= q.W) { x1 = 0; ++y1; }
        uchar2 o;
        const float2 fa = make_float2(f[k].x, f[k].y), fb = make_float2(f[k].z, f[k].w);
        const PsfmFcGeom ga = psfm_fc_geom(x, y, fa, q), gb = psfm_fc_geom(x1, y1, fb, q);
        if (__builtin_amdgcn_ballot_w64(!(ga.interior & gb.interior)) == 0ull) {
            o.x = psfm_flow_check_px16<true>(B, fa, ga, q);
            o.y = psfm_flow_check_px16<true>(B, fb, gb, q);
        } else {
            o.x = psfm_flow_check_px16<false>(B, fa, ga, q);
            o.y = psfm_flow_check_px16<false>(B, fb, gb, q);
        }
        if (NT) __builtin_nontemporal_store(*(unsigned short*)&o, (unsigned short*)(occ_pair + p));
        else *(uchar2*)(occ_pair + p) = o;
    }
}

template <bool NT>
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_x2v_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, PsfmFcParams q, uint8_t* __restrict__ occ_out, PsfmFastDiv wdiv,
    int xcd_per)
{
    const int64_t base = (int64_t)blockIdx.y * (q.H * q.W);
    psfm_flow_check_x2v_body<NT>(flows_f + base, flows_b + base, q, occ_out + base, wdiv, xcd_per);
}

// the same for the stacks of a BATCH of sequences (psfm_connect_batch): blockIdx.y = frame pair pair0 + y, blockIdx.z = sequence
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_x2v_batch_kernel(const PsfmFcSeq* __restrict__ T, PsfmFcParams q, PsfmFastDiv wdiv,
                                                                               int xcd_per, int pair0)
{
    const PsfmFcSeq& t = T[blockIdx.z];
    const int pair = pair0 + (int)blockIdx.y;
    if (pair >= t.n_pairs) return;
    const int64_t base = (int64_t)pair * (q.H * q.W);
    psfm_flow_check_x2v_body<false>((const float2*)t.ff + base, (const float2*)t.fb + base, q, t.occ + base, wdiv, xcd_per);
}

PsfmFcParams psfm_fc_params(int h, int w, float thres);

// Background form for the side stream of track_optimize: 32 VGPRs and a dynamic LDS reservation (PSFM_FC_BG_LDS_KB,
// default 48 KB, never touched) -- exactly ONE such block fits on a CU BESIDE the three blocks of the frame kernel (3 waves x
// 160 VGPRs + 32 <= 512 per SIMD lane; 3 x 37 KB + 48 KB <= 160 KB of LDS) and three when the CU is otherwise empty, so
// flow_check runs in the issue slots and memory cycles the latency-bound solve leaves idle instead of taking CUs from it
// in bursts (a frame kernel that overlaps a full-occupancy flow_check chunk takes 2x).  Measured, 1080p x 101 frames with
// path consistency, ms per sequence on one box: full-occupancy chunks 10.28; this kernel without the reservation 10.28,
// 20 KB 10.24, 40 KB 9.99-10.14, 44 KB 9.78, 48 KB 9.76-9.85, 52 KB 10.05, 56 / 64 KB 10.7 (two blocks per empty CU
// starve it); persistent grid-stride blocks (1 / 2 / 4 per CU) 10.30 / 10.65 / 10.55 (they hold what they got).
#define PSFM_FC_BG_LDS_KB_DEFAULT 48
#define PSFM_FC_BG_UNROLL 2   // (32 VGPRs: 3 x 160 of the frame kernel + 32 = 512 per SIMD lane)
__global__ __launch_bounds__(PSFM_BLOCK) __attribute__((amdgpu_num_vgpr(32))) void psfm_flow_check_bg_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, PsfmFcParams q, uint8_t* __restrict__ occ_out,
    PsfmFastDiv wdiv, int n_pairs, int xcd_per)
{
    extern __shared__ char psfm_fc_bg_reserve[];      // (occupancy control only)
    const int P = q.H * q.W;
    const int chunks = (P + PSFM_BLOCK * PSFM_FC_BG_UNROLL - 1) / (PSFM_BLOCK * PSFM_FC_BG_UNROLL);
    // gridDim.y == 1: a few resident blocks walk over the whole chunk of pairs; gridDim.y == n_pairs: one short-lived
    // block per 512 pixels (the LDS reservation then caps how many of them a CU hosts beside the frame kernel's blocks)
    if (gridDim.y > 1 && (int)psfm_xcd_chunk((int)blockIdx.x, xcd_per) >= chunks) return;
    const int64_t w0 = gridDim.y > 1 ? (int64_t)blockIdx.y * chunks + psfm_xcd_chunk((int)blockIdx.x, xcd_per) : blockIdx.x;
    const int64_t wstep = gridDim.y > 1 ? (int64_t)chunks * n_pairs : gridDim.x;
    for (int64_t w = w0; w < (int64_t)chunks * n_pairs; w += wstep) {
        const int pair = (int)(w / chunks), ch = (int)(w - (int64_t)pair * chunks);
        const int64_t base = (int64_t)pair * P;
        const float2* __restrict__ F = flows_f + base;
        const float2* __restrict__ B = flows_b + base;
        const int p0 = ch * (PSFM_BLOCK * PSFM_FC_BG_UNROLL) + threadIdx.x;
        float2 f[PSFM_FC_BG_UNROLL];
#pragma unroll
        for (int k = 0; k < PSFM_FC_BG_UNROLL; ++k) {
            const int p = p0 + k * PSFM_BLOCK;
            f[k] = p < P ? psfm_ld(F, (unsigned)p * 8u) : make_float2(0.f, 0.f);
        }
        int y = (int)psfm_fastdiv((unsigned)p0, wdiv), x = p0 - y * q.W;
#pragma unroll
        for (int k = 0; k < PSFM_FC_BG_UNROLL; ++k) {
            const int p = p0 + k * PSFM_BLOCK;
            if (p >= P) break;
            float e = 0.f;
            occ_out[base + p] = psfm_flow_check_px<false>(B, x, y, f[k], q, &e);
            x += PSFM_BLOCK;
            while (x >= q.W) { x -= q.W; ++y; }
        }
    }
    if (threadIdx.x == 0xffff) psfm_fc_bg_reserve[0] = 0;
}

psfm_status psfm_launch_flow_check_bg(const float* ff, const float* fb, int n_pairs, int h, int w, float thres, uint8_t* occ,
                                      int n_blocks, hipStream_t s)
{
    if (n_pairs <= 0) return PSFM_OK;
    const PsfmFcParams q = psfm_fc_params(h, w, thres);
    dim3 grid((unsigned)n_blocks);
    int xcd_per = 0;
    if (n_blocks <= 0) {      // one short-lived block per PSFM_BLOCK * PSFM_FC_BG_UNROLL pixels
        const int64_t P = (int64_t)h * w;
        const int64_t chunks = (P + PSFM_BLOCK * PSFM_FC_BG_UNROLL - 1) / (PSFM_BLOCK * PSFM_FC_BG_UNROLL);
        xcd_per = psfm_xcd_per(chunks);
        grid = dim3((unsigned)(xcd_per > 0 ? 8 * xcd_per : chunks), (unsigned)n_pairs);
    }
    const int lds_kb = getenv("PSFM_FC_BG_LDS_KB") ? atoi(getenv("PSFM_FC_BG_LDS_KB")) : PSFM_FC_BG_LDS_KB_DEFAULT;
    hipLaunchKernelGGL(psfm_flow_check_bg_kernel, grid, dim3(PSFM_BLOCK), (size_t)(lds_kb > 0 ? lds_kb : 0) * 1024, s, (const float2*)ff,
                       (const float2*)fb, q, occ, psfm_fastdiv_make((unsigned)w), n_pairs, xcd_per);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

PsfmFcParams psfm_fc_params(int h, int w, float thres)
{
    PsfmFcParams q;
    q.H = h; q.W = w;
    q.cw = (float)((double)(w - 1) / 2.0); q.ch = (float)((double)(h - 1) / 2.0);
    q.rcw = psfm_rcp_host(q.cw); q.rch = psfm_rcp_host(q.ch);
    q.thres = thres; q.t2 = psfm_sq_threshold(thres);
    return q;
}

psfm_status psfm_launch_flow_check(const float* ff, const float* fb, int n_pairs, int h, int w, float thres,
                                   uint8_t* occ, float* err, hipStream_t s)
{
    if (n_pairs <= 0) return PSFM_OK;
    const int64_t P = (int64_t)h * w;
    const PsfmFcParams q = psfm_fc_params(h, w, thres);
    static const int fc_kernel = getenv("PSFM_FC_KERNEL") ? atoi(getenv("PSFM_FC_KERNEL")) : 2;
    if (!err && fc_kernel >= 2 && (P % 2) == 0 && P >= PSFM_BLOCK * PSFM_FC2_UNROLL * 2 && ((uintptr_t)ff % 16) == 0 &&
        ((uintptr_t)fb % 16) == 0 && ((uintptr_t)occ % 2) == 0) {
        const int64_t per_block = (int64_t)PSFM_BLOCK * PSFM_FC2_UNROLL * 2;
        const int64_t chunks = (P + per_block - 1) / per_block;
        const int xcd_per = psfm_xcd_per(chunks);
        dim3 grid((unsigned)(xcd_per > 0 ? 8 * xcd_per : chunks), (unsigned)n_pairs);
        if (fc_kernel == 3)
            hipLaunchKernelGGL(psfm_flow_check_x2v_kernel<true>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb, q, occ,
                               psfm_fastdiv_make((unsigned)w), xcd_per);
        else
            hipLaunchKernelGGL(psfm_flow_check_x2v_kernel<false>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb, q, occ,
                               psfm_fastdiv_make((unsigned)w), xcd_per);
    } else if (P >= PSFM_BLOCK * PSFM_FC_UNROLL) {
        dim3 grid((unsigned)((P + PSFM_BLOCK * PSFM_FC_UNROLL - 1) / (PSFM_BLOCK * PSFM_FC_UNROLL)), (unsigned)n_pairs);
        if (err)
            hipLaunchKernelGGL(psfm_flow_check_x4_kernel<true>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb,
                               q, occ, err, psfm_fastdiv_make((unsigned)w));
        else
            hipLaunchKernelGGL(psfm_flow_check_x4_kernel<false>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb,
                               q, occ, err, psfm_fastdiv_make((unsigned)w));
    } else {
        dim3 grid((unsigned)((P + PSFM_BLOCK - 1) / PSFM_BLOCK), (unsigned)n_pairs);
        if (err)
            hipLaunchKernelGGL(psfm_flow_check_kernel<true>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb,
                               q, occ, err);
        else
            hipLaunchKernelGGL(psfm_flow_check_kernel<false>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb,
                               q, occ, err);
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// can the stacks of a batch go through psfm_flow_check_x2v_batch_kernel?  (what psfm_launch_flow_check asks of the mask-only kernel)
bool psfm_flow_check_batch_ok(int h, int w, const void* ff, const void* fb, const void* occ)
{
    const int64_t P = (int64_t)h * w;
    return (P % 2) == 0 && P >= PSFM_BLOCK * PSFM_FC2_UNROLL * 2 && ((uintptr_t)ff % 16) == 0 && ((uintptr_t)fb % 16) == 0 && ((uintptr_t)occ % 2) == 0;
}

psfm_status psfm_launch_flow_check_batch(const PsfmFcSeq* tab_dev, int n_seq, int pair0, int n_pairs, int h, int w, float thres, hipStream_t s)
{
    if (n_pairs <= 0 || n_seq <= 0) return PSFM_OK;
    const int64_t P = (int64_t)h * w;
    const PsfmFcParams q = psfm_fc_params(h, w, thres);
    const int64_t per_block = (int64_t)PSFM_BLOCK * PSFM_FC2_UNROLL * 2;
    const int64_t chunks = (P + per_block - 1) / per_block;
    const int xcd_per = psfm_xcd_per(chunks);
    dim3 grid((unsigned)(xcd_per > 0 ? 8 * xcd_per : chunks), (unsigned)n_pairs, (unsigned)n_seq);
    hipLaunchKernelGGL(psfm_flow_check_x2v_batch_kernel, grid, dim3(PSFM_BLOCK), 0, s, tab_dev, q, psfm_fastdiv_make((unsigned)w), xcd_per, pair0);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// trajectory.py:25-37 exposed as an API (parity tests, optimize_buffer-style callers)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_grid_sample_kernel(const float* __restrict__ map, int C, int H, int W,
                                                                      float cw, float ch, const double2* __restrict__ xy,
                                                                      int64_t n, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double2 p = xy[i];
    const PsfmTaps t = psfm_taps((float)p.x, (float)p.y, cw, ch, H, W);
    if (C == 2) {
        const float2 v = psfm_sample_flow((const float2*)map, H, W, t);
        ((float2*)out)[i] = v;
    } else {
        out[i] = psfm_sample_f32(map, H, W, t);
    }
}

psfm_status psfm_launch_grid_sample(const float* map, int c, int h, int w, const double* xy, int64_t n,
                                    float* out, hipStream_t s)
{
    if (n <= 0) return PSFM_OK;
    const float cw = (float)((double)(w - 1) / 2.0), ch = (float)((double)(h - 1) / 2.0);
    hipLaunchKernelGGL(psfm_grid_sample_kernel, dim3((unsigned)((n + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                       0, s, map, c, h, w, cw, ch, (const double2*)xy, n, out);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// init: frame-0 births on the full stride-r grid (trajectory.py:108,110-120; track.py:33-35)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_track_init_kernel(int64_t cap, int64_t G, int64_t g0, int GW, int ratio,
                                                                     int* __restrict__ birth_frame,
                                                                     int* __restrict__ birth_idx,
                                                                     double2* __restrict__ log0,
                                                                     PsfmCounters* __restrict__ ctr,
                                                                     PsfmShard* __restrict__ shards)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i == 0) { ctr->n_lanes = (int)G; ctr->n_lanes_snap[0] = ctr->n_lanes_snap[1] = (int)G; ctr->overflow = 0; ctr->stall = 0; ctr->sel = 0;
                  ctr->pc_frame = 1; ctr->pc_phase = 0; ctr->pc_owner = 0; ctr->solve_K = 3; }
    if (i < 2 * PSFM_NSHARD) { shards[i].fin_cnt = 0; shards[i].free_top = 0; shards[i].points = (i == 0) ? (unsigned)G : 0u; }
    if (i >= cap) return;
    if (i < G) {   // (G = the grid points this process owns, [g0, g0 + G): the whole grid unless track-sharded)
        const int64_t g = g0 + i;
        birth_frame[i] = 0;
        birth_idx[i] = (int)g;
        log0[i] = make_double2((double)((int)(g % GW) * ratio), (double)((int)(g / GW) * ratio));
    } else {
        birth_frame[i] = -1;
    }
}

psfm_status psfm_launch_track_init(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s)
{
    if (!d.shard_maps) PSFM_HIP(hipMemsetAsync(c->occupied.p, 0, (size_t)d.G * 2, s));
    else PSFM_HIP(hipMemsetAsync(d.shard_maps, 0, (size_t)d.shard_pitch * 2, s));
    PSFM_HIP(hipMemsetAsync(c->survivors.p, 0, sizeof(int) * (size_t)(d.n_flows + 1), s));
    hipLaunchKernelGGL(psfm_track_init_kernel, dim3((unsigned)((d.cap + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                       0, s, d.cap, d.shard_maps ? d.Gband : d.G, d.shard_maps ? d.g0 : (int64_t)0, d.GW, d.ratio, c->birth_frame.as<int>(), c->birth_idx.as<int>(),
                       c->log.as<double2>(), c->counters.as<PsfmCounters>(), c->shards.as<PsfmShard>());
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// K2+K3  chain_step: ONE launch per frame t = `frame`, organised so that every dependent memory round
// trip of the bookkeeping overlaps the bilinear gathers.
//
//  (A) respawn for this frame -- what the reference computes at the end of the previous extend_all
//      (trajectory.py:150-152) and instantiates at the top of the loop body (new_traj_all, :117-120):
//        distance_transform_edt(1 - occupied) > r   on the stride-r grid
//          == "no survivor's pixel inside the integer disc of radius r around the grid point" (pinned by
//             tests/golden; SURVEY A-5).  The previous launch scattered every survivor's pixel into a
//             GRID-resolution `blocked` map (all grid points within distance r of the pixel), so the test is
//             one byte per grid point.  With NO survivor at all SciPy measures to a phantom feature at
//             (y=-1,x=0): every grid point but (0,0) respawns.
//      The newborn's own chain step needs only its grid position, so its gathers are issued together with
//      (B)'s; its lane (a recycled one popped from the free stack, else a fresh one) is only needed for the
//      final stores.
//  (B) chain step of every lane born before t (trajectory.py:25-37,45-62, track.py:38-46, extend_all :129-147):
//        p = log[t][L];  flow = S(F_t, p);  occ = S(occ_t, p) > 0.1;  next = p + flow (f64);
//        alive = strictly inside & !occ  -> log[t+1][L] = next, block the grid points around int(next)
//        dead  -> birth_frame[L] = -2 - birth_frame   ("died at step t", no atomics here)
//  (C) the deaths marked by the PREVIOUS launch are turned into records (key, lane) and free lanes at the top
//      of this launch, where the atomics' latency hides under the gathers.  A lane freed at step t-1 becomes
//      poppable at launch t+1 (free stacks are double-buffered by frame parity, so pops never race pushes).
//  All counts go through wavefront ballots -> LDS -> ONE atomic per block per table on shard
//  blockIdx % PSFM_NSHARD.
//  Algorithmic bytes per alive lane: 16 (p) + 32 (4 flow taps) + 4 (occ taps) + 16 (next) + 1 (occupancy).
// ------------------------------------------------------------------------------------------------
#define PSFM_CHAIN_STEP_MAIN_TU
#include "psfm_chain_step.h"

template <int R, bool OPT>
__global__ __launch_bounds__(PSFM_CHAIN_BLOCK) PSFM_CHAIN_WAVES void psfm_chain_step_kernel(PsfmChainArgs a)
{
    PsfmChainOut o;
    (void)psfm_chain_step_body<R, OPT, false>(a, o);
}

// ---- batched forms (psfm_batch.hip): B same-shape sequences, blockIdx.y = sequence, gridDim.x a multiple of 8 (workgroups go to
// the XCDs round-robin by linear id: block (x, y) then still sits on XCD x % 8, which psfm_xcd_tile counts on) ----
template <int R, bool OPT>
__global__ __launch_bounds__(PSFM_CHAIN_BLOCK) PSFM_CHAIN_WAVES void psfm_chain_step_batch_kernel(const PsfmBatchSeq* __restrict__ seqs, int frame)
{
    const PsfmBatchSeq& q = seqs[blockIdx.y];
    if (frame >= q.n_flows) return;
    PsfmChainArgs a = q.a;
    // (the launch may cover fewer lanes than the table has -- psfm_batch.hip: what the sequence can be expected to use --: say so if it is not enough)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.ctr->n_lanes > (int)(gridDim.x * PSFM_CHAIN_TILE)) atomicOr(&a.ctr->overflow, 16);
    psfm_chain_args_rebase(a, q.st, frame);
    PsfmChainOut o;
    (void)psfm_chain_step_body<R, OPT, false>(a, o);
}

// what psfm_launch_track_init does (two memsets + psfm_track_init_kernel) for every sequence of the batch, from its table row
// (the arguments of frame 1: slab 0 of the log, map 0 and survivor word 0 lie one stride below)
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_track_init_batch_kernel(const PsfmBatchSeq* __restrict__ seqs)
{
    const PsfmBatchSeq& q = seqs[blockIdx.y];
    const PsfmChainArgs& a = q.a;
    const int64_t stride = (int64_t)gridDim.x * PSFM_BLOCK;
    const int64_t i0 = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    const int df = a.frame;                                   // (1: the table holds the arguments of frame 1)
    double2* log0 = a.log_cur - (int64_t)df * q.st.cap;
    uint8_t* maps = (df & 1) ? const_cast<uint8_t*>(a.blocked_prev) : a.blocked_cur;     // map 0 (map 1 follows at + G)
    int* surv = a.surv_cur - df;
    if (i0 == 0) {
        PsfmCounters* ctr = a.ctr;
        ctr->n_lanes = a.G; ctr->n_lanes_snap[0] = ctr->n_lanes_snap[1] = a.G; ctr->overflow = 0; ctr->stall = 0; ctr->sel = 0;
        ctr->abort = 0; ctr->spill_cnt = 0;
        ctr->pc_frame = 1; ctr->pc_phase = 0; ctr->pc_owner = 0; ctr->solve_K = 3;
    }
    if (i0 < 2 * PSFM_NSHARD) { a.sh_fin[i0].fin_cnt = 0; a.sh_fin[i0].free_top = 0; a.sh_fin[i0].points = (i0 == 0) ? (unsigned)a.G : 0u; }
    for (int64_t i = i0; i < 2 * (int64_t)a.G; i += stride) maps[i] = 0;
    for (int64_t i = i0; i <= q.n_flows; i += stride) surv[i] = 0;
    for (int64_t i = i0; i < a.cap; i += stride) {
        if (i < a.G) {
            a.birth_frame[i] = 0;
            a.birth_idx[i] = (int)i;
            log0[i] = make_double2((double)((int)(i % a.GW) * a.ratio), (double)((int)(i / a.GW) * a.ratio));
        } else {
            a.birth_frame[i] = -1;
        }
    }
}

// one row of the batch table: the sequence's chain-step arguments at frame 1 (device-side stamp-wrap clear, as psfm_seq_kernel)
void psfm_batch_fill_seq(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch, PsfmBatchSeq* row)
{
    const int64_t Pix = (int64_t)d.H * d.W;
    memset(row, 0, sizeof(*row));
    psfm_fill_chain_args_nolaunch(c, d, flows + Pix * 2, occ + occ_pitch, 1, row->a);
    row->a.owner_clear = 1;
    row->st.flow = Pix; row->st.occ = occ_pitch; row->st.cap = d.cap;
    row->n_flows = d.n_flows;
}

psfm_status psfm_launch_track_init_batch(const PsfmBatchSeq* tab_dev, int n_seq, int64_t cap_max, hipStream_t s)
{
    int64_t nb = (cap_max + PSFM_BLOCK - 1) / PSFM_BLOCK;
    if (nb > 4096) nb = 4096;       // (grid-stride inside)
    hipLaunchKernelGGL(psfm_track_init_batch_kernel, dim3((unsigned)nb, (unsigned)n_seq), dim3(PSFM_BLOCK), 0, s, tab_dev);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_launch_chain_step_batch(psfm_ctx* owner, const PsfmBatchSeq* tab_dev, int n_seq, int ratio, int64_t cap_max, int frame,
                                         bool optimize, hipStream_t s)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    owner->prof.kernel_span(PSFM_PROF_CHAIN, &e0, &e1);
    const unsigned gx = (unsigned)(((cap_max + PSFM_CHAIN_TILE - 1) / PSFM_CHAIN_TILE + 7) / 8 * 8);
    const dim3 grid(gx, (unsigned)n_seq), block(PSFM_CHAIN_BLOCK);
    if (optimize) {
        switch (ratio) {
            case 1: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<1, true>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            case 2: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<2, true>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            case 4: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<4, true>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            default: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<0, true>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
        }
    } else {
        switch (ratio) {
            case 1: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<1, false>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            case 2: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<2, false>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            case 4: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<4, false>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
            default: hipExtLaunchKernelGGL((psfm_chain_step_batch_kernel<0, false>), grid, block, 0, s, e0, e1, 0, tab_dev, frame); break;
        }
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_clear_map_kernel(const PsfmCounters* __restrict__ ctr,
                                                                     uint8_t* __restrict__ map, int n)
{
    if (ctr->stall) return;
    const int i = blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i < n) map[i] = 0;
}

// the arguments of one chain step (also of the chain part of the merged frame kernel); launches the stamp-wrap clear of
// the blocked map when it is due
static void psfm_fill_chain_args_impl(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame,
                                      PsfmChainArgs& a, hipStream_t s, bool launch_clear);
void psfm_fill_chain_args(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame, PsfmChainArgs& a,
                          hipStream_t s)
{
    psfm_fill_chain_args_impl(c, d, flow, occ, frame, a, s, true);
}
void psfm_fill_chain_args_nolaunch(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame, PsfmChainArgs& a)
{
    psfm_fill_chain_args_impl(c, d, flow, occ, frame, a, nullptr, false);
}
static void psfm_fill_chain_args_impl(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame,
                                      PsfmChainArgs& a, hipStream_t s, bool launch_clear)
{
    a.flow = (const float2*)flow; a.occ = occ;
    a.H = d.H; a.W = d.W; a.cw = d.cw; a.ch = d.ch; a.rcw = psfm_rcp_host(d.cw); a.rch = psfm_rcp_host(d.ch);
    a.ratio = d.ratio; a.GW = d.GW; a.GH = d.GH; a.G = (int)d.G;
    double2* lg = c->log.as<double2>();
    a.log_cur = lg + (int64_t)frame * d.cap;
    a.log_next = lg + (int64_t)(frame + 1) * d.cap;
    a.birth_frame = c->birth_frame.as<int>(); a.birth_idx = c->birth_idx.as<int>();
    // blocked maps / free stacks / shard tables are double-buffered by frame parity; stamps wrap every 254 frames
    uint8_t* maps = d.shard_maps ? d.shard_maps : c->occupied.as<uint8_t>();
    const int64_t pitch = d.shard_maps ? d.shard_pitch : d.G;
    const int cur = frame & 1, prev = cur ^ 1;
    a.blocked_cur = maps + (int64_t)cur * pitch;
    a.blocked_prev = maps + (int64_t)prev * pitch;
    a.g0 = d.shard_maps ? (int)d.g0 : 0; a.Gband = d.shard_maps ? (int)d.Gband : (int)d.G; a.shard = d.shard_maps ? 1 : 0;
    a.stamp_cur = (uint8_t)((frame % 254) + 1);
    a.stamp_prev = (uint8_t)(((frame + 253) % 254) + 1);
    if (launch_clear && frame > 1 && (frame % 254) <= 1) {
        // the map about to be written last saw this stamp value 254 frames ago: clear it (with a kernel that
        // honours the stall flag -- a memset would also run for launches that are going to be re-enqueued)
        hipLaunchKernelGGL(psfm_clear_map_kernel, dim3((unsigned)((d.G + 1 + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                           0, s, c->counters.as<PsfmCounters>(), a.blocked_cur, (int)d.G + (d.shard_maps ? 1 : 0));
    }
    a.surv_prev = c->survivors.as<int>() + (frame > 0 ? frame - 1 : 0);
    a.surv_cur = c->survivors.as<int>() + frame;
    a.ctr = c->counters.as<PsfmCounters>();
    PsfmShard* sh = c->shards.as<PsfmShard>();
    a.sh_pop = sh + (int64_t)cur * PSFM_NSHARD;
    a.sh_push = sh + (int64_t)prev * PSFM_NSHARD;
    a.sh_fin = sh;   // trajectory records: one table for the whole sequence
    int* fs = c->free_stack.as<int>();
    const int64_t set = (int64_t)d.free_cap * PSFM_NSHARD;
    a.free_pop = fs + cur * set;
    a.free_push = fs + prev * set;
    a.fin_keys = c->fin_keys.as<unsigned long long>(); a.fin_lanes = c->fin_lanes.as<int>();
    a.cap = (int)d.cap; a.shard_cap = d.shard_cap; a.free_cap = d.free_cap; a.frame = frame; a.nsh = d.nsh;
    a.shift_b = d.shift_b; a.shift_d = d.shift_d;
    a.gwdiv = psfm_fastdiv_make((unsigned)d.GW); a.rdiv = psfm_fastdiv_make((unsigned)d.ratio);
    a.log_prev = lg + (int64_t)(frame > 0 ? frame - 1 : 0) * d.cap;
    a.xs = c->sol_x.as<double2>(); a.xs_stride = d.cap;
    a.owner_clear = 0;
    {
        static const int on = getenv("PSFM_XCD_TILES") ? atoi(getenv("PSFM_XCD_TILES")) : 1;     // (0: tiles in block order; measurements)
        const int gb = (a.Gband + PSFM_CHAIN_TILE - 1) / PSFM_CHAIN_TILE;
        a.xcd_tiles = (on && gb >= 64) ? gb : 0;
    }
}

psfm_status psfm_launch_chain_step(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ,
                                   int frame, bool optimize, hipStream_t s)
{
    PsfmChainArgs a;
    psfm_fill_chain_args(c, d, flow, occ, frame, a, s);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    c->prof.kernel_span(PSFM_PROF_CHAIN, &e0, &e1);
    const dim3 grid((unsigned)((d.cap + PSFM_CHAIN_TILE - 1) / PSFM_CHAIN_TILE)), block(PSFM_CHAIN_BLOCK);
    if (optimize) {
        switch (d.ratio) {
            case 1: hipExtLaunchKernelGGL((psfm_chain_step_kernel<1, true>), grid, block, 0, s, e0, e1, 0, a); break;
            case 2: hipExtLaunchKernelGGL((psfm_chain_step_kernel<2, true>), grid, block, 0, s, e0, e1, 0, a); break;
            case 4: hipExtLaunchKernelGGL((psfm_chain_step_kernel<4, true>), grid, block, 0, s, e0, e1, 0, a); break;
            default: hipExtLaunchKernelGGL((psfm_chain_step_kernel<0, true>), grid, block, 0, s, e0, e1, 0, a); break;
        }
    } else {
        switch (d.ratio) {
            case 1: hipExtLaunchKernelGGL((psfm_chain_step_kernel<1, false>), grid, block, 0, s, e0, e1, 0, a); break;
            case 2: hipExtLaunchKernelGGL((psfm_chain_step_kernel<2, false>), grid, block, 0, s, e0, e1, 0, a); break;
            case 4: hipExtLaunchKernelGGL((psfm_chain_step_kernel<4, false>), grid, block, 0, s, e0, e1, 0, a); break;
            default: hipExtLaunchKernelGGL((psfm_chain_step_kernel<0, false>), grid, block, 0, s, e0, e1, 0, a); break;
        }
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}
