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
__device__ __forceinline__ uint8_t psfm_flow_check_px16(const float2* __restrict__ B, int x, int y, float2 f, const PsfmFcParams& q)
{
    const float X = __fadd_rn((float)x, f.x), Y = __fadd_rn((float)y, f.y);
    const PsfmTaps t = psfm_taps_t<true>(X, Y, q.cw, q.ch, q.rcw, q.rch, q.H, q.W);
    const int x0 = t.x0, y0 = t.y0;
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
    const float bx = psfm_blend(nwx, nex, swx, sex, t), by = psfm_blend(nwy, ney, swy, sey, t);
    const float eu = __fadd_rn(bx, f.x), ev = __fadd_rn(by, f.y);
    const float s2 = __fmaf_rn(ev, ev, __fmul_rn(eu, eu));
    const bool oob = (X < 0.0f) | (X > (float)(q.W - 1)) | (Y < 0.0f) | (Y > (float)(q.H - 1));
    return (uint8_t)((s2 > q.t2) | oob);
}

template <bool NT>     // NT: non-temporal F loads / mask stores (streamed once: keep them out of the way of the B taps in L2)
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_x2v_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, PsfmFcParams q, uint8_t* __restrict__ occ_out, PsfmFastDiv wdiv)
{
    const int P = q.H * q.W;                      // (even: the launcher falls back to the x4 kernel otherwise)
    const int64_t base = (int64_t)blockIdx.y * P;
    const float2* __restrict__ F = flows_f + base;
    const float2* __restrict__ B = flows_b + base;
    const int p0 = (blockIdx.x * (PSFM_BLOCK * PSFM_FC2_UNROLL) + threadIdx.x) * 2;
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
        o.x = psfm_flow_check_px16(B, x, y, make_float2(f[k].x, f[k].y), q);
        o.y = psfm_flow_check_px16(B, x1, y1, make_float2(f[k].z, f[k].w), q);
        if (NT) __builtin_nontemporal_store(*(unsigned short*)&o, (unsigned short*)(occ_out + base + p));
        else *(uchar2*)(occ_out + base + p) = o;
    }
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
        dim3 grid((unsigned)((P + per_block - 1) / per_block), (unsigned)n_pairs);
        if (fc_kernel == 3)
            hipLaunchKernelGGL(psfm_flow_check_x2v_kernel<true>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb, q, occ,
                               psfm_fastdiv_make((unsigned)w));
        else
            hipLaunchKernelGGL(psfm_flow_check_x2v_kernel<false>, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb, q, occ,
                               psfm_fastdiv_make((unsigned)w));
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
    if (i == 0) { ctr->n_lanes = (int)G; ctr->overflow = 0; ctr->stall = 0; ctr->sel = 0; }
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
#define PSFM_PROBE 8

struct PsfmChainArgs {
    const float2* flow; const uint8_t* occ;
    int H, W; float cw, ch, rcw, rch;
    int ratio, GW, GH; int G;
    double2* log_cur; double2* log_next;
    int* birth_frame; int* birth_idx;
    const uint8_t* blocked_prev; uint8_t* blocked_cur; uint8_t stamp_prev, stamp_cur;
    const int* surv_prev; int* surv_cur;
    PsfmCounters* ctr;
    PsfmShard* sh_pop; PsfmShard* sh_push; PsfmShard* sh_fin;
    const int* free_pop; int* free_push;
    unsigned long long* fin_keys; int* fin_lanes;
    int cap, shard_cap, free_cap, frame, shift_b, shift_d;
    int nsh;                   // free-lane stacks in use: min(PSFM_NSHARD, blocks of the grid) -- a small grid must not probe stacks nobody fills
    PsfmFastDiv gwdiv, rdiv;   // division by GW (grid index -> row/col) and by the sample ratio
    // track_optimize (OPT kernels): the fused solve of the previous frame leaves the accepted positions of its tracks
    // (times frame-1, frame) in iterate buffer ctr->sel (0: already in the log); this launch moves them into the log
    // slabs on its way and steps from them
    double2* log_prev; const double2* xs; int64_t xs_stride;
    // track-sharded runs (psfm_shard_*): this process owns the births on grid points [g0, g0 + Gband) -- thread i tests
    // grid point g0 + i -- and the "a track survived" flag travels as byte G of the blocked maps (stamped like them), so
    // that ONE all-reduce(max) of G + 1 bytes per frame carries everything the ranks owe each other
    int g0, Gband, shard;
};

#ifndef PSFM_CHAIN_BLOCK
#define PSFM_CHAIN_BLOCK 256
#endif
#ifndef PSFM_LPT
#define PSFM_LPT 1          // lanes (and grid points) per thread: lane u of thread t is tile_base + u*BLOCK + t (2 measured no faster)
#endif
#ifndef PSFM_CHAIN_WPE
#define PSFM_CHAIN_WPE 8    // waves per SIMD the register allocation targets (64 VGPRs)
#endif
#define PSFM_CHAIN_WAVES __attribute__((amdgpu_waves_per_eu(PSFM_CHAIN_WPE, PSFM_CHAIN_WPE)))
#define PSFM_CHAIN_TILE (PSFM_CHAIN_BLOCK * PSFM_LPT)
#define PSFM_CHAIN_NW (PSFM_CHAIN_BLOCK / PSFM_WAVE)
#define PSFM_CHAIN_NSEG (PSFM_CHAIN_NW * PSFM_LPT)

// Every thread owns PSFM_LPT lanes, block-strided so that each wave-level load stays fully coalesced.  PSFM_LPT = 2
// makes the whole 1080p/r=2 grid resident in ONE dispatch round (1017 tiles of 512 at 4 waves per SIMD) with twice the
// bytes in flight per wave; measured 17.6 us vs 17.1 us for PSFM_LPT = 1 (two rounds at 7 waves per SIMD): the launch
// is bound by its serialized phases (latency + transfer of two dependent round trips), not by residency.
#ifdef PSFM_TIMELINE
// debug builds only (PSFM_EXTRA_FLAGS=-DPSFM_TIMELINE): per-block phase timestamps of ONE chosen launch
#define PSFM_TL_SLOTS 8
__device__ unsigned long long g_psfm_tl[8192 * PSFM_TL_SLOTS];
__device__ int g_psfm_tl_frame = -1;
#define PSFM_TL(k) do { if (tl_on && tid == 0) g_psfm_tl[blockIdx.x * PSFM_TL_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int psfm_debug_timeline(int frame, unsigned long long* out_host, int n_blocks)
{
    if (out_host) {
        if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_psfm_tl), (size_t)n_blocks * PSFM_TL_SLOTS * 8) != hipSuccess) return 1;
    } else {
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_psfm_tl_frame), &frame, sizeof(int)) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define PSFM_TL(k) do {} while (0)
#endif

template <int R, bool OPT>
__global__ __launch_bounds__(PSFM_CHAIN_BLOCK) PSFM_CHAIN_WAVES void psfm_chain_step_kernel(PsfmChainArgs a)
{
    __shared__ int s_births[PSFM_CHAIN_NSEG], s_pend[PSFM_CHAIN_NSEG];
    __shared__ int s_new_g[PSFM_CHAIN_TILE];        // grid index of the births, one 64-slot segment per (u, wave)
    __shared__ int s_pend_lane[PSFM_CHAIN_TILE];    // lanes of the tracks that died in the previous step, same layout
    __shared__ int s_seg_start[PSFM_PROBE + 1];
    __shared__ int s_seg_end[PSFM_PROBE + 1];
    __shared__ int s_nseg, s_alive_any, s_base_fin, s_base_free;
    const int tid = threadIdx.x, lane = psfm_lane_id(), wave = tid / PSFM_WAVE;
    const int tile = blockIdx.x * PSFM_CHAIN_TILE;
    const int frame = a.frame;
    const int ratio = R > 0 ? R : a.ratio;
#ifdef PSFM_TIMELINE
    const bool tl_on = (a.frame == g_psfm_tl_frame) && blockIdx.x < 8192;
    if (tl_on && threadIdx.x == 0) {
        g_psfm_tl[blockIdx.x * PSFM_TL_SLOTS + 0] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_psfm_tl[blockIdx.x * PSFM_TL_SLOTS + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    if (a.ctr->stall) return;   // an earlier path-consistency solve is unfinished: this launch will be re-enqueued
    const int sel = OPT ? a.ctr->sel : 0;   // (same cache line as `stall`)
    // tiles past both the lane high-water mark and the grid have nothing to do (lanes handed out during
    // this launch are born at `frame` and are stepped by their allocator, not by their own thread)
    if (tile >= max(a.ctr->n_lanes, a.Gband)) return;
    if (tid == 0) s_alive_any = 0;

    // ---- independent early loads: lane state, (speculative) tail position, respawn byte ----
    int bf[PSFM_LPT];
    double2 p[PSFM_LPT], sx1[PSFM_LPT], sx2[PSFM_LPT];
    bool birth[PSFM_LPT], live[PSFM_LPT], pend[PSFM_LPT];
    int pend_idx[PSFM_LPT];
    unsigned long long bm[PSFM_LPT], pm[PSFM_LPT];
    const int surv_prev = (frame > 0) ? (a.shard ? (int)(psfm_ld(a.blocked_prev, (unsigned)a.G) == a.stamp_prev) : *a.surv_prev) : 1;
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) {
        const int i = tile + u * PSFM_CHAIN_BLOCK + tid;
        bf[u] = -1;
        p[u] = make_double2(0.0, 0.0);
        if (i < a.cap) { bf[u] = psfm_ld(a.birth_frame, (unsigned)i * 4u); p[u] = psfm_ld(a.log_cur, (unsigned)i * 16u); }
        if (OPT && sel != 0 && i < a.cap) {   // (speculative like the tail: whether this lane took part is known with bf)
            sx1[u] = a.xs[(int64_t)(2 * sel - 2) * a.xs_stride + i];
            sx2[u] = a.xs[(int64_t)(2 * sel - 1) * a.xs_stride + i];
        }
        birth[u] = false;
        if (frame > 0 && i < a.Gband) {
            const int g = a.g0 + i;
            if (surv_prev == 0) {
                const int gy = (int)psfm_fastdiv((unsigned)g, a.gwdiv), gx = g - gy * a.GW;
                const int cx = gx * ratio, cy = gy * ratio;
                birth[u] = ((cy + 1) * (cy + 1) + cx * cx) > ratio * ratio;
            } else {
                birth[u] = psfm_ld(a.blocked_prev, (unsigned)g) != a.stamp_prev;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) {
        const int i = tile + u * PSFM_CHAIN_BLOCK + tid;
        // lanes born AT `frame` (allocated concurrently by other blocks) are not ours; a marker -2-b with b < frame
        // is a death recorded by the previous launch
        live[u] = (bf[u] >= 0) & ((bf[u] < frame) | (frame == 0));
        pend[u] = (bf[u] <= -2) & ((-2 - bf[u]) < frame);
        if (OPT && sel != 0 && live[u] && bf[u] <= frame - 2) {   // took part in the solve of frame-1 (three buffered points)
            p[u] = sx2[u];
            psfm_st(a.log_cur, (unsigned)i * 16u, sx2[u]);
            psfm_st(a.log_prev, (unsigned)i * 16u, sx1[u]);
        }
        pend_idx[u] = 0;
        // ---- block-level counts; the births' grid indices and the just-died lanes are compacted through LDS ----
        bm[u] = __ballot(birth[u]);
        pm[u] = __ballot(pend[u]);
        const int seg = u * PSFM_CHAIN_NW + wave;
        if (lane == 0) { s_births[seg] = __popcll(bm[u]); s_pend[seg] = __popcll(pm[u]); }
        if (birth[u]) s_new_g[seg * PSFM_WAVE + psfm_rank_in(bm[u])] = a.g0 + i;
        if (pend[u]) s_pend_lane[seg * PSFM_WAVE + psfm_rank_in(pm[u])] = i;
    }

    // ---- gathers of the lanes' steps (unconditional: idle lanes sample pixel (0,0)) ----
    double2 p1[PSFM_LPT];
    PsfmStepLoads l1[PSFM_LPT];
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) {
        p1[u] = live[u] ? p[u] : make_double2(0.0, 0.0);
        l1[u] = psfm_step_issue(a, p1[u]);
    }
    // birth index of the tracks that died in the previous step: needed for their records; read BEHIND the gathers (an
    // earlier load would be waited for together with the tail position, i.e. one more round trip in front of the
    // gathers) and before the second barrier, after which a newborn of this block may inherit and overwrite the lane
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u)
        if (pend[u]) pend_idx[u] = psfm_ld(a.birth_idx, (unsigned)(tile + u * PSFM_CHAIN_BLOCK + tid) * 4u);

    PSFM_TL(1);
    __syncthreads();
    PSFM_TL(2);
    // ---- the newborns' first step, compacted onto the first threads of the block ----
    // Newborn #t first inherits the lane of the block's t-th just-died track (no atomics, the lane is recycled
    // immediately); only the surplus of births pops the free stacks and only the surplus of deaths pushes them.
    int nb = 0, npd = 0, g2 = -1, L2 = -1;
    int pend_before[PSFM_LPT];   // just-died tracks of this block ranked before this wave's, per u
    {
        int before = 0, pbefore = 0;
#pragma unroll
        for (int sg = 0; sg < PSFM_CHAIN_NSEG; ++sg) {
            const int c = s_births[sg], pc = s_pend[sg];
            if (tid >= before && tid < before + c) g2 = s_new_g[sg * PSFM_WAVE + (tid - before)];
            if (tid >= pbefore && tid < pbefore + pc) L2 = s_pend_lane[sg * PSFM_WAVE + (tid - pbefore)];
#pragma unroll
            for (int u = 0; u < PSFM_LPT; ++u)
                if (sg == u * PSFM_CHAIN_NW + wave) pend_before[u] = pbefore;
            before += c;
            pbefore += pc;
        }
        nb = before;
        npd = pbefore;
    }
    const int matched = nb < npd ? nb : npd;
    const bool newborn = tid < nb;
    PsfmStepLoads l2 = {};
    if (newborn) {
        const int gy = (int)psfm_fastdiv((unsigned)g2, a.gwdiv), gx = g2 - gy * a.GW;
        l2 = psfm_step_issue(a, make_double2((double)(gx * ratio), (double)(gy * ratio)));
    }
    const int shard = blockIdx.x % PSFM_NSHARD;
    if (tid == 0) {
        int need = nb - matched;            // births that must pop a lane
        const int n_push = npd - matched;   // deaths whose lane goes back to the free stack
        // up to three independent atomics, issued back to back: one round trip
        int old_top = 0, bfin = 0, bfree = 0;
        const int sh0 = blockIdx.x % a.nsh;
        if (need > 0) old_top = atomicSub(&a.sh_pop[sh0].free_top, need);
        if (npd > 0) bfin = atomicAdd(&a.sh_fin[shard].fin_cnt, npd);
        if (n_push > 0) bfree = atomicAdd(&a.sh_push[sh0].free_top, n_push);
        s_base_fin = bfin; s_base_free = bfree;
        int nseg = 0, done = 0;
        for (int k = 0; k < PSFM_PROBE && need > 0; ++k) {
            const int sh = (sh0 + k * 7) % a.nsh;
            if (k > 0) old_top = atomicSub(&a.sh_pop[sh].free_top, need);
            const int take = old_top < 0 ? 0 : (old_top > need ? need : old_top);
            if (take < need) atomicAdd(&a.sh_pop[sh].free_top, need - take);   // give back what the stack did not have
            if (take > 0) {
                s_seg_start[nseg] = sh * a.free_cap + old_top - 1;   // rank q of the segment -> entry start - q
                done += take;
                s_seg_end[nseg] = done;
                ++nseg;
                need -= take;
            }
        }
        if (need > 0) {
            const int base_new = atomicAdd(&a.ctr->n_lanes, need);
            s_seg_start[nseg] = -(base_new + 1);   // negative: fresh lanes base_new, base_new+1, ...
            done += need;
            s_seg_end[nseg] = done;
            ++nseg;
        }
        s_nseg = nseg;
    }
    // the dead tracks' birth indices must be in registers before any newborn may overwrite birth_idx[lane]
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) asm volatile("" : : "v"(pend_idx[u]) : "memory");
    PSFM_TL(3);
    __syncthreads();
    PSFM_TL(4);
    // positions / grid index made opaque: the tap geometry is recomputed from them below instead of being carried
    // across the barriers in registers
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) asm volatile("" : "+v"(p1[u].x), "+v"(p1[u].y));
    asm volatile("" : "+v"(g2));
    double2 p2 = make_double2(0.0, 0.0);
    if (newborn) {
        const int gy = (int)psfm_fastdiv((unsigned)g2, a.gwdiv), gx = g2 - gy * a.GW;
        p2 = make_double2((double)(gx * ratio), (double)(gy * ratio));
    }
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) psfm_step_pin<false>(l1[u]);
    psfm_step_pin<true>(l2);

    bool any_alive = false;
    int npts = 0;
#pragma unroll
    for (int u = 0; u < PSFM_LPT; ++u) {
        const int i = tile + u * PSFM_CHAIN_BLOCK + tid;
        // ---- (C) deaths of the previous step -> record (+ free lane unless a newborn inherits it) ----
        if (pend[u]) {
            const int r = pend_before[u] + psfm_rank_in(pm[u]);
            if (r >= matched) {
                a.birth_frame[i] = -1;
                const int fpos = s_base_free + (r - matched);
                if (fpos < a.free_cap) a.free_push[(int64_t)(blockIdx.x % a.nsh) * a.free_cap + fpos] = i;
                else atomicOr(&a.ctr->overflow, 1);
            }
            const int rpos = s_base_fin + r;
            if (rpos < a.shard_cap) {
                const int64_t o = (int64_t)shard * a.shard_cap + rpos;
                a.fin_keys[o] = psfm_key(frame - 1, -2 - bf[u], pend_idx[u], a.shift_b, a.shift_d);
                a.fin_lanes[o] = i;
            } else {
                atomicOr(&a.ctr->overflow, 2);
            }
        }
        // ---- (B) results of the lane's step ----
        if (live[u]) {
            const PsfmStep s1 = psfm_step_finish(a, p1[u], l1[u]);
            if (s1.alive) {
                psfm_st(a.log_next, (unsigned)i * 16u, s1.next);
                psfm_block_grid<R>(a, (int)s1.next.x, (int)s1.next.y);
                any_alive = true;
                ++npts;
            } else {
                a.birth_frame[i] = -2 - bf[u];
            }
        }
    }
    // ---- (A) the newborns: lane, state, first step.  Thread t handles newborn #t; a mass respawn with more births
    //      than threads (rare) loops with a synchronous sample ----
    for (int t = tid; t < nb; t += PSFM_CHAIN_BLOCK) {
        PsfmStep s2;
        int g = g2, L = L2;
        double2 pg = p2;
        if (PSFM_LPT == 1 || t == tid) {   // (one lane per thread: at most one newborn per thread)
            s2 = psfm_step_finish(a, p2, l2);
        } else {
            int before = 0, pbefore = 0;
            g = -1; L = -1;
            for (int sg = 0; sg < PSFM_CHAIN_NSEG; ++sg) {
                const int c = s_births[sg], pc = s_pend[sg];
                if (t >= before && t < before + c) g = s_new_g[sg * PSFM_WAVE + (t - before)];
                if (t >= pbefore && t < pbefore + pc) L = s_pend_lane[sg * PSFM_WAVE + (t - pbefore)];
                before += c;
                pbefore += pc;
            }
            const int gy = (int)psfm_fastdiv((unsigned)g, a.gwdiv), gx = g - gy * a.GW;
            pg = make_double2((double)(gx * ratio), (double)(gy * ratio));
            const PsfmStepLoads lx = psfm_step_issue(a, pg);
            s2 = psfm_step_finish(a, pg, lx);
        }
        if (t >= matched) {                  // popped / fresh lane (t < matched: inherited from a just-died track)
            const int qq = t - matched;
            int k = 0, prev = 0;
            while (k < s_nseg - 1 && qq >= s_seg_end[k]) { prev = s_seg_end[k]; ++k; }
            const int q = qq - prev;
            const int st = s_seg_start[k];
            L = st >= 0 ? a.free_pop[st - q] : (-(st + 1) + q);
        }
        if (L >= 0 && L < a.cap) {
            ++npts;
            a.birth_idx[L] = g;
            a.log_cur[L] = pg;
            if (s2.alive) {
                ++npts;
                a.birth_frame[L] = frame;
                a.log_next[L] = s2.next;
                psfm_block_grid<R>(a, (int)s2.next.x, (int)s2.next.y);
                any_alive = true;
            } else {
                a.birth_frame[L] = -2 - frame;   // born and lost in the same step: a length-1 trajectory
            }
        } else {
            atomicOr(&a.ctr->overflow, 1);
        }
        if (PSFM_LPT == 1) break;
    }
    // ---- "some track survived this step" (the degenerate respawn rule of the next launch) ----
    const unsigned long long am = __ballot(any_alive);
    if (lane == 0 && am != 0ull) s_alive_any = 1;   // benign race: every writer stores 1
    // trajectory points written by this wave: one per surviving step (log_next) + one per birth (log_cur);
    // fire-and-forget atomic, summed on the host at finalize to size the result without a second sync
    {
        int w = npts;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
        if (lane == 0 && w > 0) atomicAdd(&a.sh_fin[shard].points, (unsigned)w);
    }
    PSFM_TL(5);
    __syncthreads();
    if (tid == 0 && s_alive_any) { if (a.shard) a.blocked_cur[a.G] = a.stamp_cur; else *a.surv_cur = 1; }
    PSFM_TL(6);
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_clear_map_kernel(const PsfmCounters* __restrict__ ctr,
                                                                     uint8_t* __restrict__ map, int n)
{
    if (ctr->stall) return;
    const int i = blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i < n) map[i] = 0;
}

psfm_status psfm_launch_chain_step(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ,
                                   int frame, bool optimize, hipStream_t s)
{
    PsfmChainArgs a;
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
    if (frame > 1 && (frame % 254) <= 1) {
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
