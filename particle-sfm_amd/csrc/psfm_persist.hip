// psfm_persist.hip -- K2+K3 as ONE persistent launch per sequence (track mode, track.py:24-50).
//
// Why: a per-frame chain_step launch moves 35 MB and lasts ~16 us on MI355X, of which ~2 us are launch/drain, ~4 us the
// round trip that re-reads the state the previous launch had just written (birth frame, tail position), and the
// rest two more dependent round trips -- latency, not bandwidth (per-block timeline).  Here
// every block stays resident for the whole sequence (all blocks co-resident: 8 blocks of 256 per CU), a thread IS a
// lane and keeps its track (birth frame, birth grid index, f64 position) in registers, and frames are separated by
// a device-wide barrier (two-level arrival counters + 64 replicated release flags: ~2.5 us) instead of a launch.
//
// What crosses blocks inside the kernel goes through write-through stores / L2-bypassing loads (the eight XCDs
// have private L2s): the grid-resolution `blocked` maps, the "some track survived" flag, the free-lane stacks and
// the hand-off slots below.  Everything else (the trajectory log, the death records) is only read after the kernel.
//
// Per frame t and thread (lane L = blockIdx*256 + tid, grid point g = L):
//   A  a live track issues its eight gathers at once (position from registers) -- before the barrier wait
//   B  wait for barrier t-1 (all marks of step t-1 are visible)
//   C  read-and-clear blocked[g] -> birth?; a POOLED lane polls its hand-off slot (a lane popped by another
//      block at t-1 finds its newborn there); lanes that died at t-1 (PEND) offer themselves as hosts
//   D  the first threads of the block take the "phase-2 steps": the block's newborns (from their grid points) and
//      the tracks adopted in C; newborn #k is hosted by the block's k-th PEND lane (state through LDS), surplus
//      births pop lanes from the global stacks (hand-off slot written for the owner), surplus PEND lanes are pushed
//   E  finish: blends, advance, bounds; log[t+1], marks of the survivors, death records into the block's PRIVATE
//      record segment (no atomics; spill to a shared tail when a segment is full)
//   F  all stores acknowledged -> arrive at barrier t
// A block also owns PP_GUESTS "guest" lanes whose state sits in LDS: births that find no local PEND lane go there
// before anything is popped from the global stacks, and live guests are stepped as phase-2 entries.  They are the
// head-room above the 256 x (resident blocks) thread lanes (1080p at sample_ratio 2 peaks at 525.5k lanes against
// 524288 threads on an MI355X).
// The results (ids, lengths, positions) do not depend on which lane hosts which track: ids derive from the key
// (last valid time, birth frame, birth grid index) at finalize.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include "psfm_internal.h"
#include "psfm_chain.h"

#define PP_BLOCK 256
#define PP_NW (PP_BLOCK / PSFM_WAVE)
#define PP_PROBE 8
#ifndef PP_PRE_WAVES
#define PP_PRE_WAVES 4   // waves of a block that run E ahead of the barrier wait (measured: 4 -> 13.9, 3 -> 14.3, 2 -> 14.8, 0 -> 19 us per frame)
#endif
#ifndef PP_KEEP
#define PP_KEEP 0      // free thread lanes a block keeps for its own future births; only the excess goes to the global stacks
#endif
#ifndef PP_FC_AHEAD
#define PP_FC_AHEAD 4    // fused flow_check: how far ahead of the current step a waiting wave may work (3 is mandatory).
                         // 1080p sequence end to end: window 4 -> 2.80 ms, 3 -> 2.84, 8 -> 2.96, unbounded -> 3.19 (front-loads the
                         // memory system), slices also behind the barrier -> 3.12, always one slice per frame -> 2.86
#endif
#ifndef PP_FC_FAST
#define PP_FC_FAST 0     // fused flow_check: wave-uniform "all taps interior" form (16-byte tap pairs, no padding selects)
#endif
#define PP_GUESTS 32   // extra lanes per block whose state lives in LDS (stepped as phase-2 entries)
#ifndef PP_WHATIF
#define PP_WHATIF 0      // TIMING EXPERIMENTS ONLY (the results are wrong by construction): what a frame of the loop would cost without
                         // one of the links of its chain -- 1: no births / adoptions behind the barrier (the blocked bytes are still
                         // loaded; no phase-2 gathers), 2: arrive without waiting for the marks' acknowledgement, 4: no barrier wait
                         // (blocks run free), 8: no blocked-byte / hand-off loads either, 16: no mandatory flow_check slices in front of the arrival,
                         // 32: none in the barrier wait (16 / 48: the maps stay incomplete).  profiles/EXPERIMENTS.md sections 6.2, 6.8.
#endif
#ifndef PP_WAVES_N
#define PP_WAVES_N 8     // waves per SIMD the loop is compiled for (8: 64 VGPRs, every lane of a 1080p / ratio-2 grid resident)
#endif
#define PP_WAVES __attribute__((amdgpu_waves_per_eu(PP_WAVES_N, PP_WAVES_N)))

// lane states kept in the `bf` register: >= 0 birth frame of the live track; PP_POOLED: free, reachable through the
// global stacks (or never used) -> polls its hand-off slot; PP_PEND: free, kept by the block for its own births
// (its track died in an earlier step)
#ifdef PSFM_PERSIST_CHECK
// debug builds: bounds-check every lane-indexed store, remember the first offending site in ctr->pad
#define PP_CHK(idx, lim, code) (((idx) >= 0 && (idx) < (lim)) ? true : (atomicCAS(&a.ctr->pad[0], 0, (code)), atomicMax(&a.ctr->pad[1], (int)(idx)), false))
#else
#define PP_CHK(idx, lim, code) true
#endif
#ifndef PP_COH_MARKS
#define PP_COH_MARKS true
#endif
#define PP_POOLED (-1)
#define PP_PEND (-2)

struct PsfmPersistArgs {
    const float2* flows; const uint8_t* occ;   // (n_flows,H,W,2) f32 / n_flows maps of H*W u8, `occ_pitch` bytes apart
    // fused flow_check (psfm_connect): the blocks compute the occlusion maps themselves, in the time they would spend
    // waiting at the frame barriers, always at least three frames ahead of the step that samples them
    const float2* flows_b; uint8_t* occ_w; float thres, t2; int fc; int fc_xcd_per;
    int xcd_per;                               // > 0: blocks below 8 * xcd_per own the lanes / grid points of block (b % 8) * xcd_per + b / 8 (see psfm_vblock)
    int64_t occ_pitch; PsfmFastDiv wdiv;
    int H, W; float cw, ch, rcw, rch;
    int ratio, GW, GH, G;
    float2* dlog; int cap;                     // (n_flows, cap) sampled flow of every SURVIVED step: slab t, column = lane.  A
                                               // trajectory is its birth grid point plus the running f64 sum of its column's
                                               // entries (finalize re-runs the same additions); cap = gridDim.x * (256 + PP_GUESTS)
    int cap_main;                              // gridDim.x * 256 thread lanes (columns [cap_main, cap) are the guests)
    uint8_t* maps;                             // 3 x G, 0/1, zeroed: marks of step t go to map t % 3
    unsigned* survsh;                          // 2 x 64 words (128 B apart): frame+1 of the last step a block of the shard had a survivor in
    PsfmCounters* ctr;
    PsfmShard* shards;                         // 2 x PSFM_NSHARD (free_top per parity set)
    int* free_stack; int free_cap; int nsh;   // nsh: free-lane stacks in use = min(PSFM_NSHARD, blocks)
    unsigned long long* handoff;               // cap x 3: x bits, y bits, gi | (2*birth_frame + alive) << 32
    unsigned long long* fin_keys; int* fin_lanes;
    int seg_cap; int spill_base; int spill_cap;
    int2* seg_info;                            // per block: records, points
    unsigned* bar;                             // [0, 64) shard counters, [64] top counter, [65, 129) release flags; 32 words apart
    int n_flows, shift_b, shift_d;
    PsfmFastDiv gwdiv, rdiv;
    int spin_limit;
};

// what psfm_step_issue / psfm_step_finish / psfm_block_grid need for one frame
struct PsfmFrameView {
    const float2* flow; const uint8_t* occ;
    int H, W; float cw, ch, rcw, rch;
    int ratio, GW, GH;
    uint8_t* blocked_cur; uint8_t stamp_cur;
    PsfmFastDiv rdiv;
};

__device__ __forceinline__ void psfm_put_record(const PsfmPersistArgs& a, int slot, unsigned long long key, int lane)
{
    int64_t o;
    if (slot < a.seg_cap) {
        o = (int64_t)blockIdx.x * a.seg_cap + slot;
    } else {   // private segment full: shared tail
        const int q = atomicAdd(&a.ctr->spill_cnt, 1);
        if (q >= a.spill_cap) { atomicOr(&a.ctr->overflow, 2); return; }
        o = (int64_t)a.spill_base + q;
    }
    if (!PP_CHK(lane, a.cap, 6)) return;
    a.fin_keys[o] = key;
    a.fin_lanes[o] = lane;
}

// slots for `n_dead` records of this wave in the block's private segment (LDS counter, no global atomic)
__device__ __forceinline__ int psfm_record_slot(int* s_rec_cnt, bool dead)
{
    const unsigned long long dm = __ballot(dead);
    if (dm == 0ull) return 0;
    int base = 0;
    if (psfm_lane_id() == (int)__builtin_ctzll(dm)) base = atomicAdd(s_rec_cnt, __popcll(dm));
    base = __shfl(base, (int)__builtin_ctzll(dm));
    return base + psfm_rank_in(dm);
}

#ifdef PSFM_TIMELINE
__device__ unsigned long long g_pp_tl[2 * 4096 * 8];   // two consecutive frames
__device__ int g_pp_st[4096 * 8];
__device__ int g_pp_tl_frame = -1;
#define PP_TL(k) do { if ((unsigned)(t - g_pp_tl_frame) < 2u && tid == 0) g_pp_tl[((t - g_pp_tl_frame) * 4096 + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int psfm_debug_persist_timeline(int frame, unsigned long long* out_host, int n_blocks)
{
    if (out_host && n_blocks < 0) return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pp_st), (size_t)(-n_blocks) * 32) != hipSuccess;
    if (out_host) return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pp_tl), (size_t)2 * 4096 * 64) != hipSuccess;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_tl_frame), &frame, sizeof(int)) != hipSuccess;
}
#else
#define PP_TL(k) do {} while (0)
#endif

// ---- device-wide barrier #k (k = 0: prologue, k = t + 1: end of frame t).  Arrival: 64 shard counters -> the last
// arriver of a shard bumps the top counter -> the last of those publishes k + 1 in 64 replicated release flags ----
__device__ __forceinline__ void psfm_bar_arrive(const PsfmPersistArgs& a, int shard, int k, int lane)
{
    int last = 0;
    if (lane == 0) {
        const int nblk = (int)gridDim.x;
        const unsigned members = (unsigned)((nblk - shard + PSFM_NSHARD - 1) / PSFM_NSHARD);
        const unsigned old = __hip_atomic_fetch_add(a.bar + shard * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == members * (unsigned)(k + 1)) {
            const unsigned nsh = (unsigned)(nblk < PSFM_NSHARD ? nblk : PSFM_NSHARD);
            const unsigned o2 = __hip_atomic_fetch_add(a.bar + 64 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == nsh * (unsigned)(k + 1)) last = 1;
        }
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) psfm_coh_st(a.bar + (65 + lane) * 32, (unsigned)(k + 1));
}
// thread 0 only: spin until barrier #k is released; false = gave up / somebody else did
__device__ __forceinline__ bool psfm_bar_wait(const PsfmPersistArgs& a, int shard, int k)
{
    const unsigned* flag = a.bar + (65 + shard) * 32;
    int n = 0;
    while (psfm_coh_ld(flag) < (unsigned)(k + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > a.spin_limit) { atomicOr(&a.ctr->overflow, 8); psfm_coh_st(&a.ctr->abort, 1); return false; }
        if ((n & 127) == 0 && psfm_coh_ld(&a.ctr->abort)) return false;
    }
    return true;
}

// ---- fused flow_check (utils.py:94-105): this thread's pixels of frame pair f.  Chunks of 1024 pixels, one per block
// and round (block-strided over the map); a thread owns 4 pixels 256 apart, so every load / store instruction of a
// wave is one contiguous run.  The mask bytes are written through: other XCDs sample them a few frames later ----
__device__ __forceinline__ void psfm_fc_slice(const PsfmPersistArgs& a, int f, int tid)
{
    const int P = a.H * a.W;
    PsfmFcParams q;
    q.H = a.H; q.W = a.W; q.cw = a.cw; q.ch = a.ch; q.rcw = a.rcw; q.rch = a.rch; q.thres = a.thres; q.t2 = a.t2;
    const float2* __restrict__ F = a.flows + (size_t)f * P;
    const float2* __restrict__ B = a.flows_b + (size_t)f * P;
    uint8_t* O = a.occ_w + (size_t)f * a.occ_pitch;
    // chunk -> block: XCD-aware when a.fc_xcd_per > 0 (block b runs on XCD b % 8 and takes chunk (b % 8) * per + b / 8 of every
    // round: each XCD covers one band of rows, so a row of B is gathered through ONE private L2 instead of two -- psfm_track.hip)
    const int nch = (P + 1023) / 1024;
    const int per = a.fc_xcd_per;
    for (int q0 = blockIdx.x; q0 < (per > 0 ? 8 * per : nch); q0 += gridDim.x) {
        const int ci = per > 0 ? (q0 & 7) * per + (q0 >> 3) : q0;
        if (ci >= nch) continue;
        const int p0 = ci * 1024 + tid;
        float2 fv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k * 256;
            fv[k] = p < P ? psfm_ld(F, (unsigned)p * 8u) : make_float2(0.f, 0.f);
        }
        int y = (int)psfm_fastdiv((unsigned)p0, a.wdiv), x = p0 - y * a.W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = p0 + k * 256;
            if (p >= P) break;
            uint8_t o;
#if PP_FC_FAST
            {   // wave-uniform short cut: every lane's taps inside the map (all but the border waves)
                const float X = __fadd_rn((float)x, fv[k].x), Y = __fadd_rn((float)y, fv[k].y);
                const PsfmTaps t = psfm_taps_t<true>(X, Y, q.cw, q.ch, q.rcw, q.rch, q.H, q.W);
                const bool interior = (t.x0 >= 0) & (t.x0 + 1 < q.W) & (t.y0 >= 0) & (t.y0 + 1 < q.H);
                if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) o = psfm_flow_check_px_interior(B, X, Y, fv[k], t, q);
                else { float e; int xb = x, yb = y; asm volatile("" : "+v"(xb), "+v"(yb)); o = psfm_flow_check_px<false>(B, xb, yb, fv[k], q, &e); }
            }
#else
            float e;
            o = psfm_flow_check_px<false>(B, x, y, fv[k], q, &e);
#endif
            psfm_coh_st(O + p, o);
            x += 256;
            while (x >= a.W) { x -= a.W; ++y; }
        }
    }
}

// Which 256 lanes (= grid points) a block owns.  Workgroups go to the eight XCDs round-robin and each XCD has a private L2; with
// block b on lanes [256 b, 256 b + 256) vertically adjacent grid rows -- 3.75 blocks apart at 1080p / sample_ratio 2 -- sit on
// different XCDs, and a flow row between two grid rows (tracks have sub-pixel positions: their taps span rows floor(y), floor(y) + 1)
// is pulled through two L2s: 31.0 MB of reads per step counted at the fabric for 19.2 MB of taps (profiles/r04_o_*).  Banded like
// flow_check's chunks (block b -> virtual block (b % 8) * per + b / 8), an XCD owns one band of grid rows.  Blocks beyond the grid's
// (spare lanes only) keep their own index; ids do not depend on which lane hosts which track.
__device__ __forceinline__ int psfm_vblock(const PsfmPersistArgs& a)
{
    const int b = (int)blockIdx.x;
    return (a.xcd_per > 0 && b < 8 * a.xcd_per) ? (b & 7) * a.xcd_per + (b >> 3) : b;
}

// entry k of the phase-2 list: kind 1 newborn (ex = grid index, host = local PEND thread or -1), kind 2 adopted
// track (ex = owner thread), kind 3 live guest (ex = guest slot), kind 0 none
struct PsfmEntry { int kind, ex, host; };

template <int R>
__global__ __launch_bounds__(PP_BLOCK) PP_WAVES void psfm_chain_persist_kernel(PsfmPersistArgs a)
{
    __shared__ int s_births[PP_NW], s_adopt[PP_NW], s_pend[PP_NW];
    __shared__ int s_new_g[PP_BLOCK];       // grid index of the births, one 64-slot segment per wave
    __shared__ int s_ad_tid[PP_BLOCK];      // threads that adopted a handed-off track in this frame, same layout
    __shared__ int s_pend_tid[PP_BLOCK];    // threads whose track died in the previous step, same layout
    __shared__ double2 s_xp[PP_BLOCK];      // exchange slots indexed by the OWNER thread: position in / out
    __shared__ int s_xg[PP_BLOCK];          // ... birth grid index
    __shared__ int s_xf[PP_BLOCK];          // ... 0 nothing, 1 track alive (take it), 2 track died in its step
    __shared__ int s_gi[PP_BLOCK];          // birth grid index of the thread's live track (only needed when it dies)
    __shared__ int s_npts[PP_BLOCK];        // trajectory points written by the thread
    __shared__ int s_seg_start[PP_PROBE + 1], s_seg_end[PP_PROBE + 1];
    __shared__ int s_nseg, s_base_free, s_ok, s_alive_any[2], s_rec_cnt, s_done;   // s_alive_any: two slots, by frame parity; s_done: waves of this frame whose stores are all acknowledged
    __shared__ double2 s_gp[PP_GUESTS];     // guest lanes: position at time t,
    __shared__ int s_gbf[PP_GUESTS];        // birth frame (-1 free),
    __shared__ int s_ggi[PP_GUESTS];        // birth grid index;
    __shared__ int s_glive[PP_GUESTS], s_gfree[PP_GUESTS], s_nglive, s_ngfree;   // this frame's live / free slots

    int tid = threadIdx.x;
    int L = psfm_vblock(a) * PP_BLOCK + tid;
    const int ratio = R > 0 ? R : a.ratio;
    const int shard = blockIdx.x % PSFM_NSHARD;
    const size_t P = (size_t)a.H * a.W;

    // (the per-thread indices are re-derived from `tid` at the top of every frame: values hoisted out of the frame loop
    // cost registers for its whole body, and this kernel must fit 64 VGPRs for 8 blocks per CU)
    // ---- frame-0 births on the full grid (trajectory.py:108,110-120) ----
    // registers across frames: bf (state / birth frame) and p (position at time t; zero unless live)
    int bf = PP_POOLED;
    double2 p = make_double2(0.0, 0.0);
    s_gi[tid] = L;
    s_npts[tid] = 0;
    if (L < a.G) {
        const int gy = (int)psfm_fastdiv((unsigned)L, a.gwdiv), gx = L - gy * a.GW;
        p = make_double2((double)(gx * ratio), (double)(gy * ratio));
        bf = 0;
        s_npts[tid] = 1;
    }
    s_xf[tid] = 0;
    if (tid < PP_GUESTS) s_gbf[tid] = -1;
    if (tid == 0) { s_rec_cnt = 0; s_alive_any[0] = 0; s_alive_any[1] = 0; s_done = 0; }
    // ---- prologue: the occlusion maps of the first two frame pairs, then barrier #0 ----
    int fc_next = 0;     // (per wave) frame pairs whose occlusion map this wave has finished its share of
    if (a.fc) {
        for (; fc_next < 2 && fc_next < a.n_flows; ++fc_next) psfm_fc_slice(a, fc_next, tid);
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    if (tid / PSFM_WAVE == PP_NW - 1) psfm_bar_arrive(a, shard, 0, tid & (PSFM_WAVE - 1));
    if (a.fc) {   // step 0 samples map 0 ahead of its barrier wait: every block's share of it must be in HBM first
        if (tid == 0) s_ok = psfm_bar_wait(a, shard, 0) ? 1 : 0;
        __syncthreads();
        if (!s_ok) return;
    }

    for (int t = 0; t < a.n_flows; ++t) {
        asm volatile("" : "+v"(tid));
        L = psfm_vblock(a) * PP_BLOCK + tid;
        const int lane = tid & (PSFM_WAVE - 1), wave = tid / PSFM_WAVE;
        PsfmFrameView v;
        v.flow = a.flows + (size_t)t * P; v.occ = a.occ + (size_t)t * a.occ_pitch;
        v.H = a.H; v.W = a.W; v.cw = a.cw; v.ch = a.ch; v.rcw = a.rcw; v.rch = a.rch; v.ratio = a.ratio; v.GW = a.GW; v.GH = a.GH; v.rdiv = a.rdiv;
        v.blocked_cur = a.maps + (size_t)(t % 3) * a.G; v.stamp_cur = 1;
        float2* dlog_t = a.dlog + (size_t)t * a.cap;
        const int cur = t & 1, prev = cur ^ 1;

        PP_TL(0);
        const bool live = bf >= 0;
        bool any_alive = false;          // a track of this thread survived step t
        int npts = 0;
        // one step (s = t or t + 1) of this thread's own track from the gathered taps `ld`: log entry, marks, or the death record
        auto own_step = [&](const PsfmFrameView& vs, float2* slab, int s, const PsfmStepLoads& ld, bool mine, bool& alive_any) {
            PsfmStep s1;
            s1.alive = true;
            if (mine) s1 = psfm_step_finish(vs, p, ld);
            const int slot1 = psfm_record_slot(&s_rec_cnt, !s1.alive);
            if (mine) {
                if (s1.alive) {
                    if (PP_CHK(L, a.cap, 1)) slab[L] = s1.flow;
                    psfm_block_grid<R, PP_COH_MARKS>(vs, (int)s1.next.x, (int)s1.next.y);
                    p = s1.next;
                    alive_any = true;
                    ++npts;
                } else {
                    psfm_put_record(a, slot1, psfm_key(s, bf, s_gi[tid], a.shift_b, a.shift_d), L);
                    bf = PP_PEND;       // free from the next frame on (`live` keeps it out of this frame's PEND list)
                    p = make_double2(0.0, 0.0);
                }
            }
        };
        // ---- A: gathers of the live tracks (unconditional: idle lanes sample pixel (0,0)) ----
        PsfmStepLoads l1 = psfm_step_issue(v, p);

        // ---- E: the live tracks' own step.  It needs nothing from other blocks, and its marks go to a map nobody reads
        // before barrier t (three maps), so the first PP_PRE_WAVES waves run it BEFORE waiting for barrier t-1 (their
        // ALU hides under the barrier latency) and the others behind the loads of C (under that round trip) ----
        auto do_E = [&]() { own_step(v, dlog_t, t, l1, live, any_alive); };
        // (letting a block that finds barrier t-1 already released skip ahead and run E behind C made no difference)
        const bool e_first = wave < PP_PRE_WAVES;
        if (e_first) do_E();

        // ---- B: barrier #t (end of frame t-1; #0 = prologue).  While it is not released the waves work ahead on the
        //      occlusion maps (each wave for itself: a slice has no LDS and no block barrier in it) ----
        if (a.fc) {
            const unsigned* flag = a.bar + (65 + shard) * 32;
            const int fc_lim = t + PP_FC_AHEAD < a.n_flows ? t + PP_FC_AHEAD : a.n_flows;
            while (fc_next < fc_lim) {
                if (__builtin_amdgcn_readfirstlane((int)psfm_coh_ld(flag)) >= t + 1 || (PP_WHATIF & 32)) break;
                psfm_fc_slice(a, fc_next, tid);
                ++fc_next;
            }
        }
        if (tid == 0) s_ok = (PP_WHATIF & 4) ? 1 : (psfm_bar_wait(a, shard, t) ? 1 : 0);
        __syncthreads();
        if (!s_ok) return;

#ifndef PP_NO_SETPRIO
        __builtin_amdgcn_s_setprio(3);   // behind the barrier a block is on the frame's critical path ...
#endif
        PP_TL(1);
        // ---- C (issue): respawn byte, survivor flag, hand-off slot -- consumed after E, which runs under their latency ----
        const bool poll = t > 0 && bf == PP_POOLED && !(PP_WHATIF & 8);
        const bool gridpt = t > 0 && L < a.G && !(PP_WHATIF & 8);
        uint8_t* map_prev = a.maps + (size_t)((t + 2) % 3) * a.G;   // marks of step t-1; cleared here, written again at t+2
        unsigned sv = 0, byte = 0;
        unsigned long long h0 = 0, h1 = 0, h2 = ~0ull;
        // "did any track survive step t-1?" only matters to grid point 0 (see below): one wave reads the 64 shard words.
        // (One flag word read by every thread would queue half a million L2-bypassing loads on one memory channel.)
        if (t > 0 && blockIdx.x == 0 && tid < PSFM_NSHARD) sv = psfm_coh_ld(a.survsh + (prev * PSFM_NSHARD + tid) * 32);
        if (gridpt) byte = psfm_coh_ld(map_prev + L);
        if (poll) {
            const unsigned long long* h = a.handoff + (size_t)L * 3;
            h0 = psfm_coh_ld(h); h1 = psfm_coh_ld(h + 1); h2 = psfm_coh_ld(h + 2);
        }

        if (!e_first) do_E();

        // ---- C (consume): respawn test, adoption ----
        const unsigned long long svm = __ballot(sv == (unsigned)t);   // (all 64 lanes of block 0 / wave 0 vote: shards with a survivor)
        bool birth = false;
        if (gridpt) {
            if (byte != 0) psfm_coh_st(map_prev + L, (uint8_t)0);
            birth = byte == 0;
            // No survivor at all: nothing is marked, every grid point respawns -- except that SciPy's EDT then measures
            // to a phantom feature at (y=-1, x=0): (cy+1)^2 + cx^2 > r^2 fails at grid point 0 only (1 > r^2 is false).
            if (L == 0 && svm == 0ull) birth = ((0 + 1) * (0 + 1) + 0 * 0) > ratio * ratio;
            if (PP_WHATIF & 1) birth = false;
        }
        bool adopted = false;
        if (poll) {
            const int tag = (int)(h2 >> 32);
            if ((tag >> 1) == t - 1 && !(PP_WHATIF & 1)) {          // a block popped this lane for a track born at t-1
                if (tag & 1) {
                    adopted = true;             // stepped below as a phase-2 entry; picked up after it
                    s_xp[tid] = make_double2(__longlong_as_double((long long)h0), __longlong_as_double((long long)h1));
                    s_xg[tid] = (int)(unsigned)h2;
                } else {
                    bf = PP_PEND;               // born and lost in its first step: the lane is free again
                }
            }
        }
        const bool pend = (bf == PP_PEND) & !live;
        const unsigned long long bm = __ballot(birth), am = __ballot(adopted), pm = __ballot(pend);
        if (lane == 0) { s_births[wave] = __popcll(bm); s_adopt[wave] = __popcll(am); s_pend[wave] = __popcll(pm); }
        if (birth) s_new_g[wave * PSFM_WAVE + psfm_rank_in(bm)] = L;
        if (adopted) s_ad_tid[wave * PSFM_WAVE + psfm_rank_in(am)] = tid;
        if (pend) s_pend_tid[wave * PSFM_WAVE + psfm_rank_in(pm)] = tid;
        if (wave == 0) {   // guest slots: live ones are stepped below, free ones take births
            const int gb = tid < PP_GUESTS ? s_gbf[tid] : -1;
            const bool gl = gb >= 0, gf = (tid < PP_GUESTS) & (gb < 0);
            const unsigned long long glm = __ballot(gl), gfm = __ballot(gf);
            if (lane == 0) { s_nglive = __popcll(glm); s_ngfree = __popcll(gfm); }
            if (gl) s_glive[psfm_rank_in(glm)] = tid;
            if (gf) s_gfree[psfm_rank_in(gfm)] = tid;
        }
        __syncthreads();

        PP_TL(2);
        int nb = 0, nad = 0, npd = 0, my_pend_before = 0;
#pragma unroll
        for (int w = 0; w < PP_NW; ++w) {
            if (w == wave) my_pend_before = npd;
            nb += s_births[w]; nad += s_adopt[w]; npd += s_pend[w];
        }
        const int matched = nb < npd ? nb : npd;                  // newborns hosted by the block's PEND lanes
        const int ngl = s_nglive;
        const int gmatched = (nb - matched) < s_ngfree ? (nb - matched) : s_ngfree;   // ... by its free guest lanes
        const int n2 = nb + nad + ngl;

        // ---- D/E: phase-2 steps in passes of one entry per thread (a second pass only after a mass respawn) ----
        for (int base = 0; base == 0 || base < n2; base += PP_BLOCK) {
            const int k = base + tid;
            PsfmEntry e;
            e.kind = 0; e.ex = -1; e.host = -1;
            if (k < nb) {
                int before = 0, pb = 0;
#pragma unroll
                for (int w = 0; w < PP_NW; ++w) {
                    const int c = s_births[w], pc = s_pend[w];
                    if (k >= before && k < before + c) e.ex = s_new_g[w * PSFM_WAVE + (k - before)];
                    if (k >= pb && k < pb + pc) e.host = s_pend_tid[w * PSFM_WAVE + (k - pb)];   // k < matched only
                    before += c; pb += pc;
                }
                e.kind = 1;
            } else if (k < nb + nad) {
                const int kk = k - nb;
                int before = 0;
#pragma unroll
                for (int w = 0; w < PP_NW; ++w) {
                    const int c = s_adopt[w];
                    if (kk >= before && kk < before + c) e.ex = s_ad_tid[w * PSFM_WAVE + (kk - before)];
                    before += c;
                }
                e.kind = 2;
            } else if (k < n2) {
                e.ex = s_glive[k - nb - nad];
                e.kind = 3;
            }
            PsfmStepLoads l2 = {};
            if (e.kind != 0) {
                double2 q;
                if (e.kind == 1) {
                    const int gy = (int)psfm_fastdiv((unsigned)e.ex, a.gwdiv), gx = e.ex - gy * a.GW;
                    q = make_double2((double)(gx * ratio), (double)(gy * ratio));
                } else if (e.kind == 2) {
                    q = s_xp[e.ex];
                } else {
                    q = s_gp[e.ex];
                }
                l2 = psfm_step_issue(v, q);
            }
#ifdef PSFM_TIMELINE
            if (base == 0 && tid == 0 && t == g_pp_tl_frame) {
                int* st = g_pp_st + blockIdx.x * 8;
                st[0] = nb; st[1] = nad; st[2] = npd; st[3] = ngl; st[4] = s_ngfree; st[5] = nb - matched - gmatched; st[6] = npd - matched - PP_KEEP;
            }
#endif
            if (base == 0 && tid == 0) {
                PsfmShard* sh_pop = a.shards + cur * PSFM_NSHARD;
                PsfmShard* sh_push = a.shards + prev * PSFM_NSHARD;
                int need = nb - matched - gmatched; // births that must pop a lane
                const int n_push = npd - matched - PP_KEEP;   // free lanes beyond the block's own reserve go to the global stacks
                int bfree = 0;
                const int fsh = blockIdx.x % a.nsh;   // (the barrier keeps its own 64 shards)
                if (n_push > 0) bfree = atomicAdd(&sh_push[fsh].free_top, n_push);
                int nseg = 0, done = 0;
                // pops: rounds of four independent atomics (own shard first), one round trip per round.  The request is
                // SPLIT over the four stacks: a stack is only ever asked for what will be taken from it if it has it, so
                // a count is given back only by a popper that found the stack exhausted (top < 0 meanwhile: nobody else
                // can pop below entries that are about to be handed back)
                for (int rd = 0; rd < PP_PROBE / 4 && need > 0; ++rd) {
                    const int q4 = need >> 2, r4 = need & 3;
                    int sh[4], ask[4], old[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sh[j] = (fsh + (rd * 4 + j) * 7) % a.nsh;
                        ask[j] = q4 + (j < r4 ? 1 : 0);
                        old[j] = ask[j] > 0 ? atomicSub(&sh_pop[sh[j]].free_top, ask[j]) : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int take = old[j] < 0 ? 0 : (old[j] > ask[j] ? ask[j] : old[j]);
                        if (take < ask[j]) atomicAdd(&sh_pop[sh[j]].free_top, ask[j] - take);
                        if (take > 0) {
                            s_seg_start[nseg] = sh[j] * a.free_cap + old[j] - 1;   // rank q of the segment -> entry start - q
                            done += take;
                            s_seg_end[nseg] = done;
                            ++nseg;
                            need -= take;
                        }
                    }
                }
                s_base_free = bfree;
                if (need > 0) {
                    const int base_new = atomicAdd(&a.ctr->n_lanes, need);
                    s_seg_start[nseg] = -(base_new + 1);   // negative: fresh lanes base_new, base_new+1, ...
                    done += need;
                    s_seg_end[nseg] = done;
                    ++nseg;
                }
                s_nseg = nseg;
            }
            PP_TL(3);
            __syncthreads();
            PP_TL(4);

            if (base == 0) {
                // ---- free lanes beyond the newborns they host and the block's reserve go to the global stacks (poppable
                //      from the next frame on); the others stay PEND = free, local ----
                if (pend) {
                    const int r = my_pend_before + psfm_rank_in(pm);
                    if (r >= matched + PP_KEEP) {
                        int* free_push = a.free_stack + (size_t)prev * a.free_cap * PSFM_NSHARD;
                        const int fpos = s_base_free + (r - matched - PP_KEEP);
                        if (fpos < a.free_cap && PP_CHK(fpos, a.free_cap, 7)) psfm_coh_st(free_push + (size_t)(blockIdx.x % a.nsh) * a.free_cap + fpos, L);
                        else atomicOr(&a.ctr->overflow, 1);
                        bf = PP_POOLED;
                    }
                }
            }
            // ---- phase-2 results: ONE finish for the three kinds.  The block's first wave holds all of them (newborns,
            //      adopted tracks, live guests); three divergent copies of the blend / bounds / marks code would run one
            //      after the other in exactly the wave every other wave of the block is waiting for ----
            psfm_step_pin<true>(l2);
            asm volatile("" : "+v"(e.ex));   // the entry's position is rebuilt, not carried across the barrier
            if (e.kind != 0) {
                double2 q;
                int bfk, gik, col, gs = -1;      // birth frame, birth grid index, log column (= lane) of the entry's track
                if (e.kind == 1) {
                    const int gy = (int)psfm_fastdiv((unsigned)e.ex, a.gwdiv), gx = e.ex - gy * a.GW;
                    q = make_double2((double)(gx * ratio), (double)(gy * ratio));
                    bfk = t; gik = e.ex;
                    // lane of the newborn: the block's k-th PEND lane, else a free guest lane, else a popped / fresh lane
                    if (k < matched) {
                        col = psfm_vblock(a) * PP_BLOCK + e.host;
                    } else if (k < matched + gmatched) {
                        gs = s_gfree[k - matched];
                        col = a.cap_main + blockIdx.x * PP_GUESTS + gs;
                    } else {
                        const int* free_pop = a.free_stack + (size_t)cur * a.free_cap * PSFM_NSHARD;
                        const int qq = k - matched - gmatched;
                        int j = 0, pv = 0;
                        while (j < s_nseg - 1 && qq >= s_seg_end[j]) { pv = s_seg_end[j]; ++j; }
                        const int st = s_seg_start[j];
                        if (st >= 0) (void)PP_CHK(st - (qq - pv), 2 * a.free_cap * PSFM_NSHARD, 9);
                        col = st >= 0 ? psfm_coh_ld(free_pop + (st - (qq - pv))) : (-(st + 1) + (qq - pv));
                        if (col >= a.cap_main) col = -1;
                    }
                } else if (e.kind == 2) {
                    q = s_xp[e.ex];
                    bfk = t - 1; gik = s_xg[e.ex];
                    col = psfm_vblock(a) * PP_BLOCK + e.ex;
                } else {
                    q = s_gp[e.ex];
                    bfk = s_gbf[e.ex]; gik = s_ggi[e.ex];
                    col = a.cap_main + blockIdx.x * PP_GUESTS + e.ex;
                }
                if (col >= 0) {
                    (void)PP_CHK(col, a.cap, 2);
                    const PsfmStep s2 = psfm_step_finish(v, q, l2);
                    if (e.kind == 1) ++npts;                  // the birth point itself
                    if (s2.alive) {
                        if (PP_CHK(col, a.cap, 3)) dlog_t[col] = s2.flow;
                        psfm_block_grid<R, PP_COH_MARKS>(v, (int)s2.next.x, (int)s2.next.y);
                        any_alive = true;
                        ++npts;
                    } else {   // (a newborn lost in its first step is a length-1 trajectory)
                        const int slot = atomicAdd(&s_rec_cnt, 1);
                        psfm_put_record(a, slot, psfm_key(t, bfk, gik, a.shift_b, a.shift_d), col);
                    }
                    // ---- where the track lives from here on ----
                    if (e.kind == 1) {
                        const bool popped = k >= matched + gmatched;
                        if (popped && s2.alive && t == a.n_flows - 1) {
                            // born in the last frame on a popped lane: its owner never gets to adopt it, so the
                            // "still active at the end" record (last valid time n_flows) is written here
                            const int slot = atomicAdd(&s_rec_cnt, 1);
                            psfm_put_record(a, slot, psfm_key(a.n_flows, t, e.ex, a.shift_b, a.shift_d), col);
                        }
                        if (k < matched) {
                            s_xp[e.host] = s2.next; s_xg[e.host] = e.ex; s_xf[e.host] = s2.alive ? 1 : 2;
                        } else if (gs >= 0) {
                            if (s2.alive) { s_gp[gs] = s2.next; s_ggi[gs] = e.ex; s_gbf[gs] = t; }
                        } else if (PP_CHK(col, a.cap_main, 8)) {
                            unsigned long long* h = a.handoff + (size_t)col * 3;
                            psfm_coh_st(h, (unsigned long long)__double_as_longlong(s2.next.x));
                            psfm_coh_st(h + 1, (unsigned long long)__double_as_longlong(s2.next.y));
                            psfm_coh_st(h + 2, (unsigned long long)(unsigned)e.ex |
                                                   ((unsigned long long)(unsigned)(2 * t + (s2.alive ? 1 : 0)) << 32));
                        }
                    } else if (e.kind == 2) {
                        if (s2.alive) s_xp[e.ex] = s2.next;
                        s_xf[e.ex] = s2.alive ? 1 : 2;
                    } else {
                        if (s2.alive) s_gp[e.ex] = s2.next;
                        else s_gbf[e.ex] = -1;   // free from the next frame on (this frame's free list is already fixed)
                    }
                } else {
                    atomicOr(&a.ctr->overflow, 4);   // more tracks than resident lanes: the per-frame path takes over
                }
            }
        }
        PP_TL(5);
        if (npts) s_npts[tid] += npts;
        const unsigned long long alm = __ballot(any_alive);
        if (lane == 0 && alm != 0ull) s_alive_any[t & 1] = 1;   // benign race: every writer stores 1
        __syncthreads();
        // ---- owners pick up what the phase-2 threads computed for them ----
        {
            const int f = s_xf[tid];
            if (f != 0) {
                s_xf[tid] = 0;
                if (f == 1) {
                    bf = adopted ? t - 1 : t;   // adopted: born at t-1 in the block that popped this lane; else hosted newborn
                    s_gi[tid] = s_xg[tid];
                    p = s_xp[tid];
                } else {
                    bf = PP_PEND;               // the track died in this step (its record is written): free from t+1
                }
            }
        }
        if (tid == 0 && s_alive_any[t & 1]) { psfm_coh_st(a.survsh + (cur * PSFM_NSHARD + shard) * 32, (unsigned)(t + 1)); s_alive_any[t & 1] = 0; }
        // ---- F: everything this block wrote is acknowledged -> arrive ----
        // (a step samples the occlusion map of its frame as early as right after the previous arrival, when only the
        // barrier before that one is known to be complete: maps up to t+2 must be finished before arriving at #t+1)
        if (a.fc) {
            const int need = t + 3 < a.n_flows ? t + 3 : a.n_flows;
            if (!(PP_WHATIF & 16)) for (; fc_next < need; ++fc_next) psfm_fc_slice(a, fc_next, tid);
        }
        // ---- F: arrive.  No block barrier in front of it: a wave whose stores are acknowledged counts itself in (LDS) and
        //      goes on to the next frame; the wave that comes last arrives for the block ----
        if (!(PP_WHATIF & 2)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        PP_TL(6);
        {
            int last_wave = 0;
            if (lane == 0) last_wave = atomicAdd(&s_done, 1) == PP_NW - 1 ? 1 : 0;
            last_wave = __builtin_amdgcn_readfirstlane(last_wave);
            if (last_wave) {
                if (lane == 0) s_done = 0;
                PP_TL(7);
                psfm_bar_arrive(a, shard, t + 1, lane);
            }
        }
#ifndef PP_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);   // ... its next step ahead of the barrier is not
#endif
    }

    // ---- the tracks still active at the end (clear_active, trajectory.py:154-158): last valid time = n_flows ----
    const int lane = tid & (PSFM_WAVE - 1), wave = tid / PSFM_WAVE;
    {
        const bool alive = bf >= 0;
        const int slot = psfm_record_slot(&s_rec_cnt, alive);
        if (alive) psfm_put_record(a, slot, psfm_key(a.n_flows, bf, s_gi[tid], a.shift_b, a.shift_d), L);
        if (tid < PP_GUESTS && s_gbf[tid] >= 0) {
            const int gslot = atomicAdd(&s_rec_cnt, 1);
            psfm_put_record(a, gslot, psfm_key(a.n_flows, s_gbf[tid], s_ggi[tid], a.shift_b, a.shift_d),
                            a.cap_main + blockIdx.x * PP_GUESTS + tid);
        }
    }
    // trajectory points written by this block (sizes the result without a second host sync)
    {
        __shared__ int s_pts[PP_NW];
        int w = s_npts[tid];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o);
        if (lane == 0) s_pts[wave] = w;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int j = 0; j < PP_NW; ++j) tot += s_pts[j];
            const int c = s_rec_cnt;
            a.seg_info[blockIdx.x] = make_int2(c < a.seg_cap ? c : a.seg_cap, tot);
        }
    }
}

// one launch instead of three memsets + a counter kernel in front of every sequence
__global__ __launch_bounds__(256) void psfm_persist_init_kernel(PsfmCounters* ctr, PsfmShard* shards, int G,
                                                                unsigned long long* maps64, size_t n_maps64,
                                                                unsigned long long* bar64, size_t n_bar64,
                                                                unsigned long long* handoff, size_t n_handoff)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (i == 0) { ctr->n_lanes = G; ctr->overflow = 0; ctr->stall = 0; ctr->abort = 0; ctr->spill_cnt = 0; ctr->pad[0] = 0; ctr->pad[1] = 0; }
    if (i < 2 * PSFM_NSHARD) { shards[i].fin_cnt = 0; shards[i].free_top = 0; shards[i].points = 0u; }
    for (size_t k = i; k < n_maps64; k += stride) maps64[k] = 0ull;
    for (size_t k = i; k < n_bar64; k += stride) bar64[k] = 0ull;
    for (size_t k = i; k < n_handoff; k += stride) handoff[k] = ~0ull;   // tag -1: no hand-off
}

// Largest grid of 256-thread blocks that is resident at once on this device (0: unknown / not available).
int psfm_persist_guests(void) { return PP_GUESTS; }

int psfm_persist_max_blocks(psfm_ctx* c)
{
    // the minimum over the instantiations that can be launched (they differ by a few registers; the grid barrier needs
    // every block resident whichever one runs)
    if (c->persist_max_blocks >= 0) return c->persist_max_blocks;
    int cus = 0;
    c->persist_max_blocks = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int per_cu = 1 << 30;
    const void* inst[4] = {(const void*)psfm_chain_persist_kernel<0>, (const void*)psfm_chain_persist_kernel<1>,
                           (const void*)psfm_chain_persist_kernel<2>, (const void*)psfm_chain_persist_kernel<4>};
    for (int k = 0; k < 4; ++k) {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, inst[k], PP_BLOCK, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        per_cu = v < per_cu ? v : per_cu;
    }
    // MI355X_MICROARCH.md: 256-thread blocks are admitted up to 8 per CU whatever the API answers
    if (per_cu > 8) per_cu = 8;
    c->persist_max_blocks = per_cu * cus;
    return c->persist_max_blocks;
}

psfm_status psfm_launch_chain_persist(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ,
                                      int64_t occ_pitch, const float* flows_b, float thres, hipStream_t s)
{
    const size_t bar_bytes = (size_t)(4 * PSFM_NSHARD + 1) * 128;   // barrier lines + 2 x 64 survivor words
    {
        const size_t n_maps64 = ((size_t)d.G * 3 + 7) / 8, n_bar64 = bar_bytes / 8, n_handoff = (size_t)d.nblk * PP_BLOCK * 3;
        hipLaunchKernelGGL(psfm_persist_init_kernel, dim3(1024), dim3(256), 0, s, c->counters.as<PsfmCounters>(),
                           c->shards.as<PsfmShard>(), (int)d.G, c->occupied.as<unsigned long long>(), n_maps64,
                           c->persist_bar.as<unsigned long long>(), n_bar64, c->handoff.as<unsigned long long>(), n_handoff);
    }
    PsfmPersistArgs a;
    a.flows = (const float2*)flows; a.occ = occ;
    a.occ_pitch = occ_pitch;
    a.flows_b = (const float2*)flows_b; a.occ_w = const_cast<uint8_t*>(occ); a.thres = thres; a.t2 = psfm_sq_threshold(thres); a.fc = flows_b != nullptr;
    {
        static const int xcd_on = getenv("PSFM_FC_XCD") ? atoi(getenv("PSFM_FC_XCD")) : 1;
        const int64_t nch = ((int64_t)d.H * d.W + 1023) / 1024;
        a.fc_xcd_per = xcd_on && nch >= 64 ? (int)((nch + 7) / 8) : 0;
    }
    {
        // XCD-banded lane ownership (psfm_vblock): the blocks of the grid in eight bands, when every banded index is a block of the launch
        static const int pp_xcd = getenv("PSFM_PP_XCD") ? atoi(getenv("PSFM_PP_XCD")) : 1;
        const int64_t gb = (d.G + PP_BLOCK - 1) / PP_BLOCK;
        const int per = (int)((gb + 7) / 8);
        a.xcd_per = (pp_xcd && gb >= 64 && 8 * per <= d.nblk) ? per : 0;
    }
    a.wdiv = psfm_fastdiv_make((unsigned)d.W);
    a.H = d.H; a.W = d.W; a.cw = d.cw; a.ch = d.ch; a.rcw = psfm_rcp_host(d.cw); a.rch = psfm_rcp_host(d.ch);
    a.ratio = d.ratio; a.GW = d.GW; a.GH = d.GH; a.G = (int)d.G;
    a.dlog = c->log.as<float2>(); a.cap = (int)d.cap; a.cap_main = d.nblk * PP_BLOCK;
    a.maps = c->occupied.as<uint8_t>();
    a.ctr = c->counters.as<PsfmCounters>();
    a.shards = c->shards.as<PsfmShard>();
    a.free_stack = c->free_stack.as<int>(); a.free_cap = d.free_cap; a.nsh = d.nsh;
    a.handoff = c->handoff.as<unsigned long long>();
    a.fin_keys = c->fin_keys.as<unsigned long long>(); a.fin_lanes = c->fin_lanes.as<int>();
    a.seg_cap = d.seg_cap; a.spill_base = d.nblk * d.seg_cap; a.spill_cap = d.spill_cap;
    a.seg_info = c->seg_info.as<int2>();
    a.bar = c->persist_bar.as<unsigned>();
    a.survsh = a.bar + (2 * PSFM_NSHARD + 1) * 32;
    a.n_flows = d.n_flows; a.shift_b = d.shift_b; a.shift_d = d.shift_d;
    a.gwdiv = psfm_fastdiv_make((unsigned)d.GW); a.rdiv = psfm_fastdiv_make((unsigned)d.ratio);
    a.spin_limit = 1 << 18;   // ~35 ms of polling before a block gives up (the per-frame path then reruns the sequence)
    if (const char* e = getenv("PSFM_PERSIST_SPIN_LIMIT")) a.spin_limit = atoi(e);   // tests: 0 forces the hand-over
    hipEvent_t e0 = nullptr, e1 = nullptr;
    c->prof.kernel_span(PSFM_PROF_CHAIN, &e0, &e1, true);
    const dim3 grid((unsigned)d.nblk), block(PP_BLOCK);
    // PSFM_PERSIST_COOP=1: a COOPERATIVE launch -- the runtime refuses it when the grid cannot be co-resident on the device
    // (instead of the loop discovering that after its barrier spin limit) and runs it on the device's cooperative queue.
    // Measured on MI355X: +50 us per 100-frame 1080p sequence (2.73 vs 2.68 ms per step), which is why the plain launch
    // behind the residency query + the per-device gate stays the default; the bounded spin and the hand-over to per-frame
    // launches remain the safety net in both forms.
    const bool coop = getenv("PSFM_PERSIST_COOP") && atoi(getenv("PSFM_PERSIST_COOP")) != 0;
    const void* fn = d.ratio == 1 ? (const void*)psfm_chain_persist_kernel<1> : d.ratio == 2 ? (const void*)psfm_chain_persist_kernel<2>
                   : d.ratio == 4 ? (const void*)psfm_chain_persist_kernel<4> : (const void*)psfm_chain_persist_kernel<0>;
    if (coop) {
        void* kargs[] = {(void*)&a};
        if (e0) PSFM_HIP(hipEventRecord(e0, s));
        const hipError_t le = hipLaunchCooperativeKernel(fn, grid, block, kargs, 0, s);
        if (le != hipSuccess) {
            (void)hipGetLastError();
            psfm_set_error("cooperative launch of the persistent frame loop refused: %s", hipGetErrorString(le));
            return PSFM_ERR_CAPACITY;      // the caller reruns the sequence with one launch per frame
        }
        if (e1) PSFM_HIP(hipEventRecord(e1, s));
        return PSFM_OK;
    }
    switch (d.ratio) {
        case 1: hipExtLaunchKernelGGL(psfm_chain_persist_kernel<1>, grid, block, 0, s, e0, e1, 0, a); break;
        case 2: hipExtLaunchKernelGGL(psfm_chain_persist_kernel<2>, grid, block, 0, s, e0, e1, 0, a); break;
        case 4: hipExtLaunchKernelGGL(psfm_chain_persist_kernel<4>, grid, block, 0, s, e0, e1, 0, a); break;
        default: hipExtLaunchKernelGGL(psfm_chain_persist_kernel<0>, grid, block, 0, s, e0, e1, 0, a); break;
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}
