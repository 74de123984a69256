// psfm_chain_step.h -- the per-frame chain step (K2 + K3: trajectory.py:25-37,45-62 + track.py:38-46 + extend_all :129-152)
// as a device function, shared by psfm_chain_step_kernel (psfm_track.hip) and the merged frame kernel of track_optimize
// (psfm_solver.hip: chain step + the whole path-consistency solve of the frame in ONE launch).
#pragma once
#include "psfm_internal.h"
#include "psfm_chain.h"

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
    // device-paced sequence (psfm_seq_kernel): no host-side clear of the blocked map at the stamp wrap -- the grid point's
    // owner thread zeroes the byte it has just read when the map is about to be written with recycled stamps
    int owner_clear;
    // XCD-banded tiles (psfm_xcd_tile): > 0 = blocks of the grid (ceil(Gband / tile)); block b below it works on tile
    // band(b % 8) + b / 8 instead of tile b -- each XCD's private L2 then serves one band of grid rows (see psfm_persist.hip)
    int xcd_tiles;
};

// What changes from one frame to the next in the arguments of a chain step (everything else is per-sequence): strides of
// the per-frame stacks.  psfm_chain_args_rebase turns the arguments of frame `a.frame` into those of frame f -- the
// same values psfm_fill_chain_args computes on the host (checked there under PSFM_SEQ_CHECK).
struct PsfmSeqStride { int64_t flow, occ, cap; };
__host__ __device__ inline void psfm_chain_args_rebase(PsfmChainArgs& a, const PsfmSeqStride& st, int f)
{
    const int64_t df = (int64_t)f - a.frame;
    a.flow += df * st.flow; a.occ += df * st.occ;
    a.log_cur += df * st.cap; a.log_next += df * st.cap;
    a.log_prev = a.log_cur - (f > 0 ? st.cap : 0);
    if (df & 1) {
        const uint8_t* bp = a.blocked_prev; a.blocked_prev = a.blocked_cur; a.blocked_cur = const_cast<uint8_t*>(bp);
        PsfmShard* sp = a.sh_pop; a.sh_pop = a.sh_push; a.sh_push = sp;
        const int* fp = a.free_pop; a.free_pop = a.free_push; a.free_push = const_cast<int*>(fp);
    }
    a.stamp_cur = (uint8_t)((f % 254) + 1);
    a.stamp_prev = (uint8_t)(((f + 253) % 254) + 1);
    a.surv_cur += df; a.surv_prev = a.surv_cur - (f > 0 ? 1 : 0);
    a.frame = f;
}

// One sequence of a BATCH (psfm_connect_batch: B same-shape sequences -- the directory of sequences the reference's driver walks,
// run_particlesfm.py:168-176 -- through ONE launch per frame, blockIdx.y = sequence): the arguments of its chain step at frame
// a.frame and the strides that turn them into any other frame's.  Every sequence keeps its own context (lane tables, log,
// counters, result), so a block only ever needs its sequence's row of this table.
struct PsfmBatchSeq { PsfmChainArgs a; PsfmSeqStride st; int n_flows; int pad; };

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
#if defined(PSFM_TIMELINE) && defined(PSFM_CHAIN_STEP_MAIN_TU)
#define PSFM_TL_ON 1
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

void psfm_fill_chain_args(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame, PsfmChainArgs& a,
                          hipStream_t s);
void psfm_fill_chain_args_nolaunch(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ, int frame, PsfmChainArgs& a);

// What the merged frame kernel (psfm_solver.hip) takes over from the chain step of a thread's lane: does the lane's track
// take part in the path-consistency solve of this frame (alive after the step, born at least two frames ago), and its
// three buffered positions p0 (time frame-1), p1 (the tail, time frame), p2 (= p1 + flow, time frame+1).
struct PsfmChainOut { bool solve; double2 p0, p1, p2; int tile; };     // tile: first lane of the block's tile

// Tile of block b: a bijection on the n grid blocks [0, n) that hands XCD x (= b % 8: workgroups go to the XCDs round-robin) the
// x-th of eight contiguous bands of tiles -- sizes differ by at most one --; blocks beyond the grid (spare lanes) keep their own.
__device__ __forceinline__ int psfm_xcd_tile(int b, int n)
{
    if (n <= 0 || b >= n) return b;
    const int q0 = n >> 3, r0 = n & 7, x = b & 7;
    return x * q0 + (x < r0 ? x : r0) + (b >> 3);
}

// One chain step of the block's lanes (+ the births of the frame): the body of psfm_chain_step_kernel, also the first
// part of the merged frame kernel.  MERGED: the tile bound comes from the lane-count snapshot of the previous launch
// (PsfmCounters::n_lanes_snap -- every block of the launch must agree on which blocks take part in the solve's tickets,
// and n_lanes itself grows while the launch runs), p0 is loaded with the other early loads, and `o` is filled.
// Returns false when the block has nothing to do (stalled sequence / tile beyond the lanes and the grid).
template <int R, bool OPT, bool MERGED>
__device__ __forceinline__ bool psfm_chain_step_body(const PsfmChainArgs& a, PsfmChainOut& o)
{
    __shared__ int s_births[PSFM_CHAIN_NSEG], s_pend[PSFM_CHAIN_NSEG];
    __shared__ int s_new_g[PSFM_CHAIN_TILE];        // grid index of the births, one 64-slot segment per (u, wave)
    __shared__ int s_pend_lane[PSFM_CHAIN_TILE];    // lanes of the tracks that died in the previous step, same layout
    __shared__ int s_seg_start[PSFM_PROBE + 1];
    __shared__ int s_seg_end[PSFM_PROBE + 1];
    __shared__ int s_nseg, s_alive_any, s_base_fin, s_base_free;
    const int tid = threadIdx.x, lane = psfm_lane_id(), wave = tid / PSFM_WAVE;
    const int tile = psfm_xcd_tile((int)blockIdx.x, a.xcd_tiles) * PSFM_CHAIN_TILE;
    o.tile = tile;
    const int frame = a.frame;
    const int ratio = R > 0 ? R : a.ratio;
#ifdef PSFM_TL_ON
    const bool tl_on = (a.frame == g_psfm_tl_frame) && blockIdx.x < 8192;
    if (tl_on && threadIdx.x == 0) {
        g_psfm_tl[blockIdx.x * PSFM_TL_SLOTS + 0] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_psfm_tl[blockIdx.x * PSFM_TL_SLOTS + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    o.solve = false;
    if (a.ctr->stall) return false;   // an earlier path-consistency solve is unfinished: this launch will be re-enqueued
    const int sel = OPT ? a.ctr->sel : 0;   // (same cache line as `stall`)
    // tiles past both the lane high-water mark and the grid have nothing to do (lanes handed out during
    // this launch are born at `frame` and are stepped by their allocator, not by their own thread)
    if (tile >= max(MERGED ? a.ctr->n_lanes_snap[frame & 1] : a.ctr->n_lanes, a.Gband)) return false;
    if (tid == 0) s_alive_any = 0;

    // ---- independent early loads: lane state, (speculative) tail position, respawn byte ----
    int bf[PSFM_LPT];
    double2 p[PSFM_LPT], sx1[PSFM_LPT], sx2[PSFM_LPT];
    double2 p0m = make_double2(0.0, 0.0);     // MERGED: position at time frame-1 (speculative, like the tail)
    if (MERGED && tile + tid < a.cap) p0m = psfm_ld((const double2*)a.log_prev, (unsigned)(tile + tid) * 16u);
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
            // (this map is written again at frame + 1; stamps come round every 254 frames: its reader wipes it in time)
            if (a.owner_clear && ((frame + 1) % 254) <= 1) const_cast<uint8_t*>(a.blocked_prev)[g] = 0;
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
            if (MERGED) p0m = sx1[u];
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
            if (MERGED && u == 0 && s1.alive && bf[u] <= frame - 1) { o.solve = true; o.p0 = p0m; o.p1 = p1[u]; o.p2 = s1.next; }
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
    return true;
}
