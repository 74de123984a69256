// psfm_solver.hip -- K5/K6: the path-consistency solve (trajectory.py:161-194 +
// optimize/src/trajectory_optimize.cpp:30-96) as a device-resident Ceres-compatible trust-region loop.
//
// Objective per track (path_consistency_cost.h:42-59), x = (x1,y1,x2,y2):
//     r = [ p1 - ref1 ; s (p2 - ref2) ; (p2 - p1) - F12(p1) ],   cost = 1/2 sum r^2
// F12 is the f64 clamp-to-edge bilinear interpolator of linear_interpolation.h:97-123 (NOT the fp32
// zero-padded sampler).  The Hessian is block diagonal (one 4x4 block per track), but Ceres' trust-region
// loop is global: ONE cost, ONE radius, ONE accept/reject and ONE termination test over all tracks
// (SURVEY.md Appendix B).  To land on the same iterate the whole control flow of Ceres 2.0.0
// (TrustRegionMinimizer + DoglegStrategy/TRADITIONAL_DOGLEG + Jacobi scaling + SPARSE_NORMAL_CHOLESKY,
// max 200 iterations) is reproduced with its scalars reduced over all tracks:
//
//   fused     (track_optimize's default) ONE launch per solve -- per FRAME together with the chain step (psfm_frame_kernel):
//             thread = track runs K trust-region iterations in registers, speculating what Ceres does on every solve that
//             converges without a rejection (Gauss-Newton step inside the trust region, mu = min_mu, step accepted); the K
//             x 13 sums are reduced wave -> block -> group -> grid and the last block REPLAYS the control flow below over
//             them.  The first decision that is not "Gauss-Newton step accepted" ends the belief: termination = done, anything
//             else = the solve is redone by the launch chain from the values it started with.
//   pc_init   (launch chain) per track: refs/scale (fp32 sampler, trajectory.py:173-183), Jacobi scaling, cost and the
//             Gauss-Newton system at the start values; its control step runs Ceres' iteration 0 and fixes the first step
//   pc_iter   ONE launch per trust-region iteration, one pass per track: the candidate x + a u + b d of the dogleg step the
//             control step chose, its cost, and -- evaluated AHEAD, as if the step were accepted -- the Gauss-Newton system
//             and its sums at the candidate.  The sums at an iterate (|ghat|^2, |gn|^2, ghat.gn, |J u|^2, (J u).(J d),
//             |J d|^2) price EVERY dogleg step of that iterate -- norm and model decrease -- in the control step, so a
//             rejection (radius halves, same system, new coefficients) and an acceptance (the system at the new iterate is
//             already reduced) both go on with the next launch: launches = iterations, no re-issued launches.  (Round 2
//             speculated the Gauss-Newton step per launch and re-issued the launch whenever the reduced norms prescribed
//             another dogleg case: iterations + non-Gauss-Newton steps launches.)  An invalid step (model decrease <= 0)
//             raises mu, which changes the system at the same x: one "refresh" launch.  x ping-pongs between iterate
//             buffers 1 and 2; buffer 0 (the caller's values / the log slabs) is only written by the write-back.
//   control   pc_chain_control (launch chain) / pc_control_step (replay of a fused launch): Ceres' scalar logic for one
//             iteration from the global sums -- accept / reject / radius / mu / the three tolerances.  Consecutive rejections
//             whose shrunken radius still contains the Gauss-Newton step reproduce the same candidate, so they are replayed
//             without relaunching (bit-identical to Ceres, which recomputes the same step each time).  Run by the last block
//             of each launch (ticket after write-through partials), or -- track-sharded runs -- by psfm_pc_control_kernel on
//             the totals over all ranks.  Nothing returns to the host inside the loop.
//
// All arithmetic f64 (psfm_pc_core.h: explicit fma); reductions have a fixed order (bitwise reproducible for a given
// lane assignment).
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <hip/hip_ext.h>

#include "psfm_device.h"
#include "psfm_internal.h"
#include "psfm_chain_step.h"
#include "psfm_pc_core.h"
#include "psfm_pc_control.h"

#define PC_BLOCK 256
#include "psfm_pc_resident.h"
#include "psfm_pc_reduce.h"
#ifndef PC_FUSED_PAIR
#define PC_FUSED_PAIR false  // ... and the fused solve's: its lanes are neighbours in the image, and the frame kernel spills 7 VGPRs with them
#endif
#ifndef PC_RED_ROWS
#define PC_RED_ROWS 32        // partial rows a thread of the last block keeps in flight
#endif
#ifndef PC_MAX_BLOCKS
#define PC_MAX_BLOCKS 1024   // rows of the launch chain's partial sums
#endif
#ifndef PC_CHAIN_BLOCKS
#define PC_CHAIN_BLOCKS 512  // blocks of a launch-chain kernel: 512 measured best on the 401-frame 1080p run in round 2 (152 VGPRs, 3
                             // waves/SIMD: 66 ms; 768: 67.6, 384: 70.9, 430: 69.2, 537: 71 -- an equal number of blocks on every CU
                             // matters more than an equal number of tracks per thread)
#endif
#define PC_KMAX 8            // fused solve: most trust-region iterations speculated in one launch
#ifndef PSFM_FRAME_WAVES
#define PSFM_FRAME_WAVES 4    // psfm_frame_kernel (host-paced merged frames, the sharded engine's frames)
#endif
#ifndef PSFM_SEQ_WAVES_DEFAULT
#define PSFM_SEQ_WAVES_DEFAULT 4
#endif
#define PC_GROUP 32          // fused solve: blocks per first-level reduction group



struct PcParams {
    // geometry
    int H, W;
    float cw, ch;
    // frame mode: lanes; batch mode: rows
    const int* birth_frame;   // NULL in batch mode
    int max_birth;            // track participates iff 0 <= birth_frame <= max_birth
    const int* n_lanes_ptr;   // device count of lanes (frame mode) or NULL
    int n_rows;               // upper bound of the index range (cap or batch n)
    // iterate buffers: buffer 0 = (x1a, x2a) -- the caller's values (frame mode: the log slabs), never overwritten before
    // the write-back; buffer m >= 1 = (xs + (2m-2) * xs_stride, xs + (2m-1) * xs_stride).  The launch chain ping-pongs
    // between buffers 1 and 2; the fused solve stores the iterate after m accepted steps in buffer m.
    double2 *x1a, *x2a;
    double2* xs; int64_t xs_stride;
    const double2* p0;        // frame mode: log slab f-1
    // per-track constants
    double2 *ref1, *ref2;
    double* scale;
    double2* jscale;          // (S0^2, S1^2): squared Jacobi scaling of columns 0, 1 (columns 2, 3 need none: psfm_pc_core.h)
    const float2* flow12;
    // init-only inputs (frame mode)
    const float2* flow01;
    const float2* flow02;
    const uint8_t* occ02;
    double* partials;         // [PC_MAX_BLOCKS][PC_NSUM]
    // launch chain / resident solve: the lanes that take part in this solve, compacted per block by pc_init (pc_build_list):
    // block b's entries are list[b * list_pitch + 0 .. list_n[b]), thread t walks entries t, t + PC_BLOCK, ...
    int* list; int* list_n; int list_pitch; int list_banded;
    PsfmSolveCtrl* ctrl;
    int* stall;               // != 0: an earlier solve of this sequence ran out of unrolled iterations -> do nothing
    unsigned* ticket;         // last-block detection
    int frame;                // frame index of this solve (stall value / stats slot)
    psfm_solve_stats* stats_dev;   // per-frame statistics (frame mode) or NULL
    // fused solve (psfm_pc_fused_kernel)
    int K;                    // speculated trust-region iterations per launch (<= PC_KMAX)
    double* gpart;            // [groups][PC_KMAX][PC_NSUM] sums of PC_GROUP consecutive blocks
    unsigned* gticket;        // [groups] + 1 (top)
    int* sel;                 // PsfmCounters::sel: buffer that holds the accepted iterate of the last solve (0: the log)
    int* n_lanes_snap;        // PsfmCounters::n_lanes_snap[2] (frame mode)
    // track-sharded runs: the tracks of the solve are spread over several processes.  A launch then only EXPORTS its sums
    // ([K or 1][PC_NSUM], this process's tracks); the ranks combine them (all-gather, rank order) and every rank runs the
    // same control step on the totals (psfm_pc_control_kernel)
    double* export_sums;
};

__device__ __forceinline__ double2* pc_buf1(const PcParams& P, int m) { return m == 0 ? P.x1a : P.xs + (int64_t)(2 * m - 2) * P.xs_stride; }
__device__ __forceinline__ double2* pc_buf2(const PcParams& P, int m) { return m == 0 ? P.x2a : P.xs + (int64_t)(2 * m - 1) * P.xs_stride; }
// the launch chain's candidate buffer while the iterate sits in buffer `cur`

// The per-track arithmetic (evaluation, normal equations through the 2x2 Schur complement, dogleg step, the terms of the 13
// sums) lives in psfm_pc_core.h; it is shared by the launch chain and the fused solve, so the two give the same bits.
// P.jscale holds (S0^2, S1^2): the squared Jacobi scaling of columns 0, 1.
__device__ __forceinline__ PcConst pc_const_load(double s, double2 js)
{
    PcConst c;
    c.s = s; c.S0q = js.x; c.S1q = js.y; c.H22 = fma(s, s, 1.0);
    return c;
}

__device__ __forceinline__ double pc_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ __forceinline__ double pc_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
    return v;
}

// ... -> partials[blockIdx] (launch chain)
__device__ __forceinline__ void pc_block_reduce(double acc[PC_NSUM], double* __restrict__ partials)
{
    __shared__ double s_blk[PC_NSUM];
    pc_block_sums<PC_NSUM>(acc, s_blk);
    // write-through (sc0 sc1) store: the last block of the launch reads these with matching loads, so no
    // agent-scope release (an L2 write-back per block) is needed -- MI355X_MICROARCH.md, valid hand-off forms
    if (threadIdx.x < PC_NSUM)
        __hip_atomic_store(&partials[(int64_t)blockIdx.x * PC_NSUM + threadIdx.x], s_blk[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


__device__ __forceinline__ bool pc_participates(const PcParams& P, int i, int n)
{
    if (i >= n) return false;
    if (!P.birth_frame) return true;
    const int bf = P.birth_frame[i];
    return bf >= 0 && bf <= P.max_birth;
}

// The tracks of a solve, dealt to the blocks of the launch chain: block b looks at the lane chunks b, b + gridDim.x, ... (PC_BLOCK
// lanes each) and compacts the lanes that take part (ballot -> wave offsets through LDS) into ITS list, in lane order.  Every
// later launch of the solve -- pc_iter, the resident solve -- walks that list: thread t takes entries t, t + PC_BLOCK, ...; no
// launch after pc_init reads a birth frame or skips a lane, the resident solve knows how many tracks a thread will ever hold,
// and all forms add the tracks' terms in the same order.  Returns the block's count (also left in list_n[b]).
__device__ __forceinline__ int pc_build_list(const PcParams& P, int n)
{
    __shared__ int s_wc[PC_BLOCK / PSFM_WAVE];
    int* lst = P.list + (int64_t)blockIdx.x * P.list_pitch;
    const int lane = threadIdx.x & (PSFM_WAVE - 1), w = threadIdx.x / PSFM_WAVE;
    // Which chunks (pc_list_plan, psfm_pc_resident.h): by default XCD-BANDED -- the blocks of XCD x = b % 8 share the x-th eighth of the
    // chunks, so the taps of an XCD's tracks come from one band of the flow field (2 MB of 16.6 at 1080p: it stays in that XCD's 4 MB
    // L2 across the rounds of a solve) instead of from all of it.  P.list_banded == 0 or a grid that is not a multiple of 8: chunks b,
    // b + gridDim.x, ...
    const PcListPlan plan = pc_list_plan((int)blockIdx.x, (int)gridDim.x, n, P.list_banded);
    const int first = plan.first, step = plan.step, band0 = plan.band0;
    int base = 0;
    for (int q = first; pc_list_chunk_ok(plan, q); q += step) {
        const int i = (band0 + q) * PC_BLOCK + (int)threadIdx.x;
        const bool part = pc_participates(P, i, n);
        const unsigned long long m = __ballot(part);
        if (lane == 0) s_wc[w] = __popcll(m);
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int qq = 0; qq < PC_BLOCK / PSFM_WAVE; ++qq) { if (qq < w) off += s_wc[qq]; tot += s_wc[qq]; }
        if (part) lst[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        base += tot;
        __syncthreads();      // (s_wc is rewritten by the next chunk; the list entries are visible to the block behind it)
    }
    if (threadIdx.x == 0) P.list_n[blockIdx.x] = base;
    return base;
}


// (force-inlined: as a called function it dragged the call ABI's register budget into the kernels -- 248 VGPRs,
// 2 waves/SIMD -- although the per-track code needs 152)
__device__ __forceinline__ void pc_reduce_and_control(PsfmSolveCtrl* __restrict__ ctrl, double* partials, int n_blocks,
                                                      int is_init, double* export_sums);

// Publish this block's partials and find out whether it is the last one to finish: write-through payload ->
// drain -> ticket (device-scope atomic); the last block reads the payload with cache-bypassing loads.
__device__ __forceinline__ bool pc_is_last_block(unsigned* ticket)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1u);
        if (s_last) *ticket = 0u;   // next launch starts from zero (visible after the kernel boundary)
    }
    __syncthreads();
    return s_last != 0;
}

// ------------------------------------------------------------------------------------------------
// pc_init: iteration 0.  Frame mode also prepares ref1/ref2/scale (trajectory.py:173-183).
// ------------------------------------------------------------------------------------------------
// refs / scale of a track from its p0 (trajectory.py:173-183): the fp32 sampler on flow01, flow02, occ02
__device__ __forceinline__ void pc_refs(const PcParams& P, double2 p0, double2& r1, double2& r2, double& s)
{
    const PsfmTaps t = psfm_taps((float)p0.x, (float)p0.y, P.cw, P.ch, P.H, P.W);
    const PsfmTapIdx k = psfm_tap_idx(P.H, P.W, t);
    const float2 f01 = psfm_sample_flow(P.flow01, k, t);
    const float2 f02 = psfm_sample_flow(P.flow02, k, t);
    const float o02 = psfm_sample_mask(P.occ02, k, t);
    // (1.0 - occ02) * (|flow02| < 20) in fp32; numpy's norm = sqrt(u*u + v*v) without fma (trajectory.py:179)
    const float nrm = sqrtf(__fadd_rn(__fmul_rn(f02.x, f02.x), __fmul_rn(f02.y, f02.y)));
    const float sf = __fmul_rn(__fsub_rn(1.0f, o02), nrm < 20.0f ? 1.0f : 0.0f);
    s = (double)sf;
    r1 = make_double2(p0.x + (double)f01.x, p0.y + (double)f01.y);
    r2 = make_double2(p0.x + (double)f02.x, p0.y + (double)f02.y);
}

// iteration 0 of ONE track (lane i): refs / scale (frame mode: from its p0; batch mode: given), Jacobi scaling from the Jacobian
// at the start values, cost and system there (mu = min_mu); its terms go to acc[], what a solve keeps of it to o.
struct PcInit { double2 r1, r2; double s; PcConst c; double x[4]; PcSys y; };
__device__ __forceinline__ void pc_init_entry(const PcParams& P, int i, bool store, double acc[PC_NSUM], PcInit& o)
{
    const double mu = 1e-8;
    if (P.p0) {
        pc_refs(P, P.p0[i], o.r1, o.r2, o.s);
        if (store) { P.ref1[i] = o.r1; P.ref2[i] = o.r2; P.scale[i] = o.s; }
    } else {
        o.r1 = P.ref1[i]; o.r2 = P.ref2[i]; o.s = P.scale[i];
    }
    const double2 p1 = P.x1a[i], p2 = P.x2a[i];
    o.x[0] = p1.x; o.x[1] = p1.y; o.x[2] = p2.x; o.x[3] = p2.y;
    // Jacobi scaling 1/(1+sqrt(colnorm^2)) from the Jacobian at x0, computed once (trust_region_minimizer.cc)
    double r0[6], j0[4];
    pc_core_eval((const PcF2*)P.flow12, P.H, P.W, o.x, o.r1.x, o.r1.y, o.r2.x, o.r2.y, o.s, r0, j0);
    o.c = pc_core_const(o.s, j0);
    if (store) P.jscale[i] = make_double2(o.c.S0q, o.c.S1q);
    acc[SUM_CNT] += 1.0;
    acc[SUM_COST0] += pc_core_cost(r0);
    pc_core_system<true>(o.x, r0, j0, o.c, mu, pc_core_iA22(o.c, mu), acc, o.y, CH_QUD, CH_QDD);
}

// iteration 0 for this block's tracks: the block's list, then its entries
__device__ __forceinline__ void pc_init_tracks(const PcParams& P, double acc[PC_NSUM])
{
    // the chain step in front of this solve has consumed PsfmCounters::sel (positions of an earlier fused solve): the launch
    // chain works in buffer 0 from here on, whatever becomes of it
    if (P.sel && blockIdx.x == 0 && threadIdx.x == 0) *P.sel = 0;
    const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
#pragma unroll
    for (int k = 0; k < PC_NSUM; ++k) acc[k] = 0.0;
    const int cnt = pc_build_list(P, n);
    const int* lst = P.list + (int64_t)blockIdx.x * P.list_pitch;
    for (int p = threadIdx.x; p < cnt; p += PC_BLOCK) {
        PcInit o;
        pc_init_entry(P, lst[p], true, acc, o);
    }
}

__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_init_kernel(PcParams P)
{
    if (*P.stall) return;
    double acc[PC_NSUM];
    pc_init_tracks(P, acc);
    pc_block_reduce(acc, P.partials);
    if (pc_is_last_block(P.ticket)) pc_reduce_and_control(P.ctrl, P.partials, (int)gridDim.x, 1, P.export_sums);
}

// What one launch of the chain does for ONE track whose state is in memory (lane i): the refresh form (the system at x for a
// raised mu) or the evaluate-ahead form (candidate of (a, b), its cost, the system there).  Sums into acc[].
// (Round 3 measured how early these loads can be requested -- the state with the birth frame, the next track's state under this
// track's arithmetic, a fully software-pipelined loop, two tracks per thread at a time: level or slower, profiles/EXPERIMENTS.md
// 5.2.  What helped is not streaming the state at all: psfm_pc_resident_kernel below.)
#ifndef PC_ITER_PAIR
#define PC_ITER_PAIR true   // the launch chain's taps as 16-byte pairs (psfm_pc_core.h)
#endif
__device__ __forceinline__ void pc_refresh_entry(const PcParams& P, int i, const double2* xc1, const double2* xc2, double mu, double acc[PC_NSUM])
{
    const double2 r1 = P.ref1[i], r2 = P.ref2[i];
    const double s = P.scale[i];
    const PcConst c = pc_const_load(s, P.jscale[i]);
    const double2 p1 = xc1[i], p2 = xc2[i];
    const double x[4] = {p1.x, p1.y, p2.x, p2.y};
    double r[6], jac[4];
    PcSys y;
    pc_core_eval<PC_ITER_PAIR>((const PcF2*)P.flow12, P.H, P.W, x, r1.x, r1.y, r2.x, r2.y, s, r, jac);
    pc_core_system<true>(x, r, jac, c, mu, pc_core_iA22(c, mu), acc, y, CH_QUD, CH_QDD);
}

__device__ __forceinline__ void pc_ahead_entry(const PcParams& P, int i, const double2* xc1, const double2* xc2, double2* xn1, double2* xn2,
                                               double mu, double mu_next, double a, double b, double acc[PC_NSUM])
{
    const PcF2* F12 = (const PcF2*)P.flow12;
    const double2 r1 = P.ref1[i], r2 = P.ref2[i];
    const double s = P.scale[i];
    const PcConst c = pc_const_load(s, P.jscale[i]);
    const double2 p1 = xc1[i], p2 = xc2[i];
    const double x[4] = {p1.x, p1.y, p2.x, p2.y};
    double r[6], jac[4], xp[4], unused[PC_NSUM];      // (unused: the sums at x are in the control block already)
    PcSys y;
#pragma unroll
    for (int k = 0; k < PC_NSUM; ++k) unused[k] = 0.0;
    pc_core_eval<PC_ITER_PAIR>(F12, P.H, P.W, x, r1.x, r1.y, r2.x, r2.y, s, r, jac);
    pc_core_system<false>(x, r, jac, c, mu, pc_core_iA22(c, mu), unused, y, 0, 0);
    pc_core_step<false, false>(x, r, jac, c, y, a, b, acc, xp);
    xn1[i] = make_double2(xp[0], xp[1]);
    xn2[i] = make_double2(xp[2], xp[3]);
    // the candidate's cost and, ahead of the decision, the system there (what the next iteration needs if it is accepted)
    pc_core_eval<PC_ITER_PAIR>(F12, P.H, P.W, xp, r1.x, r1.y, r2.x, r2.y, s, r, jac);
    acc[SUM_COST] += pc_core_cost(r);
    pc_core_system<true>(xp, r, jac, c, mu_next, pc_core_iA22(c, mu_next), acc, y, CH_QUD, CH_QDD);
}

// DoglegStrategy::StepAccepted: the mu of the system at a candidate that is accepted
__device__ __forceinline__ double pc_mu_next(double mu) { return fmax(1e-8, 2.0 * mu / 10.0); }

// ... for the entries p0, p0 + PC_BLOCK, ... of this block's list (the launch chain: all of them; the resident solve: the ones
// beyond the slots it keeps on chip)
__device__ __forceinline__ void pc_iter_tracks(const PcParams& P, int p0, int cur, double mu, double a, double b, bool refresh, double acc[PC_NSUM])
{
    const double2* xc1 = pc_buf1(P, cur);
    const double2* xc2 = pc_buf2(P, cur);
    double2* xn1 = pc_buf1(P, pc_other(cur));
    double2* xn2 = pc_buf2(P, pc_other(cur));
    const double mu_next = pc_mu_next(mu);
    const int cnt = P.list_n[blockIdx.x];
    const int* lst = P.list + (int64_t)blockIdx.x * P.list_pitch;
    if (refresh) {      // the system at x for the mu an invalid step has raised (rare)
        for (int p = p0; p < cnt; p += PC_BLOCK) pc_refresh_entry(P, lst[p], xc1, xc2, mu, acc);
        return;
    }
    for (int p = p0; p < cnt; p += PC_BLOCK) pc_ahead_entry(P, lst[p], xc1, xc2, xn1, xn2, mu, mu_next, a, b, acc);
}

// ------------------------------------------------------------------------------------------------
// pc_iter: one trust-region iteration at the current iterate.
// ------------------------------------------------------------------------------------------------
#ifdef PSFM_TIMELINE
// debug builds only: phase timestamps of the pc_iter launches (first 64 real launches), per block
__device__ unsigned long long g_pc_tl[64 * 1024 * 4];
__device__ int g_pc_tl_n = 0;
extern "C" int psfm_debug_solver_timeline(unsigned long long* out_host, int* n_host)
{
    if (hipMemcpyFromSymbol(n_host, HIP_SYMBOL(g_pc_tl_n), sizeof(int)) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pc_tl), sizeof(unsigned long long) * 64 * 1024 * 4) != hipSuccess;
}
#define PC_TL(k) do { if (tl_slot >= 0 && threadIdx.x == 0) g_pc_tl[((size_t)tl_slot * 1024 + blockIdx.x) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// resident solve: rounds PC_RTL_R0 .. + 7 of the launch with epoch PC_RTL_EPOCH, 16 stamps per block and round: [8][512][16]
#define PC_RTL_EPOCH 12u
#define PC_RTL_R0 8u
#define PC_RTL(k) do { if (rtl >= 0 && threadIdx.x == 0) g_pc_tl[((size_t)rtl * 512 + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PC_TL(k) do {} while (0)
#define PC_RTL(k) do {} while (0)
#endif

__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_iter_kernel(PcParams P)
{
    if (*P.stall) return;
    const PsfmSolveCtrl C = *P.ctrl;
    if (C.done) return;
#ifdef PSFM_TIMELINE
    const int tl_slot = C.iteration == 1 && g_pc_tl_n < 64 ? g_pc_tl_n : -1;   // the first pc_iter launch of a solve
    PC_TL(0);
#endif
    double acc[PC_NSUM];
#pragma unroll
    for (int k = 0; k < PC_NSUM; ++k) acc[k] = 0.0;
    pc_iter_tracks(P, (int)threadIdx.x, C.cur, C.mu, C.dl_a, C.dl_b, C.kind_next != 0, acc);
    PC_TL(1);
    pc_block_reduce(acc, P.partials);
    const bool last_block = pc_is_last_block(P.ticket);
    PC_TL(2);
    if (last_block) {
        pc_reduce_and_control(P.ctrl, P.partials, (int)gridDim.x, 0, P.export_sums);
        PC_TL(3);
#ifdef PSFM_TIMELINE
        if (threadIdx.x == 0 && tl_slot >= 0) { g_pc_tl[((size_t)tl_slot * 1024 + 1023) * 4 + 0] = blockIdx.x; g_pc_tl_n = tl_slot + 1; }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// The order in which the blocks' sums are added (every form of the chain: launches, the resident solve, the sharded export),
// with L = min(PC_LEADERS, n_blocks) and Q = ceil(ceil(n_blocks / PC_LEADERS) / 4):
//     total = ((t_0 + t_1) + t_2) + t_3,      t_j  = S_{8j} + S_{8j+1} + ... + S_{8j+7}  (those below L, in order)
//     S_x   = ((s_x0 + s_x1) + s_x2) + s_x3,  s_xj = the sums of blocks x + PC_LEADERS m, m in [j Q, (j + 1) Q), in order
// -- S_x is what ONE block of the resident solve (the "leader" x) adds up from its members' granules, the rest is what every
// block adds up from the leaders' (pc_res_allreduce): two hops, a few loads per lane in each.  SUM_GMAX by max.
// ------------------------------------------------------------------------------------------------
// (PC_LEADERS, pc_tree_q and the same order as plain arithmetic on rows -- pc_tree_totals, what the host mirror adds up with: psfm_pc_resident.h)

// Executed by the LAST block of pc_init / pc_iter to finish (detected with a ticket behind write-through partials; the loads
// here bypass the caches -- cdna_hip_programming.md G16): the reduction above, then the scalar control step on thread 0.
// totals of the launch's partial rows, in LDS (pc_totals()), valid for every thread of the block after the call
__device__ __forceinline__ double* pc_totals()
{
    __shared__ double s_tot[PC_NSUM];
    return s_tot;
}
__device__ __forceinline__ void pc_reduce_totals(double* partials, int n_blocks)
{
    __shared__ double s_sub[PC_LEADERS][4][PC_NSUM + 1];
    __shared__ double s_S[PC_LEADERS][PC_NSUM + 1];
    double* s_tot = pc_totals();
    const int L = n_blocks < PC_LEADERS ? n_blocks : PC_LEADERS;
    const int Q = pc_tree_q(n_blocks);
    // work item (x, j, k): PC_LEADERS x 4 x 16 of them, 8 per thread; ALL loads of a thread are issued before the first is used
    // (they bypass the caches: every dependent batch would be a full round trip on the tail of the launch), then added in
    // member order
    constexpr int ITEMS = PC_LEADERS * 4 * 16 / PC_BLOCK, QMAX = PC_MAX_BLOCKS / PC_LEADERS / 4;
    double pv[ITEMS][QMAX];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int w = threadIdx.x + it * PC_BLOCK;
        const int k = w & 15, j = (w >> 4) & 3, x = w >> 6;
        const int cnt = (n_blocks - x + PC_LEADERS - 1) / PC_LEADERS;
#pragma unroll
        for (int u = 0; u < QMAX; ++u) {
            const int m = j * Q + u;
            pv[it][u] = (k < PC_NSUM && x < L && u < Q && m < cnt)
                            ? __hip_atomic_load(&partials[(int64_t)(x + PC_LEADERS * m) * PC_NSUM + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                            : 0.0;
        }
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int w = threadIdx.x + it * PC_BLOCK;
        const int k = w & 15, j = (w >> 4) & 3, x = w >> 6;
        if (k >= PC_NSUM || x >= L) continue;
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < QMAX; ++u) v = (k == SUM_GMAX) ? fmax(v, pv[it][u]) : v + pv[it][u];
        s_sub[x][j][k] = v;
    }
    __syncthreads();
    for (int w = threadIdx.x; w < PC_LEADERS * 16; w += PC_BLOCK) {
        const int k = w & 15, x = w >> 4;
        if (k >= PC_NSUM || x >= L) continue;
        const double s0 = s_sub[x][0][k], s1 = s_sub[x][1][k], s2 = s_sub[x][2][k], s3 = s_sub[x][3][k];
        s_S[x][k] = (k == SUM_GMAX) ? fmax(fmax(fmax(s0, s1), s2), s3) : ((s0 + s1) + s2) + s3;
    }
    __syncthreads();
    if (threadIdx.x < PC_NSUM) {
        const int k = threadIdx.x;
        double t[4];
        for (int j = 0; j < 4; ++j) {
            double v = 0.0;
            for (int x = 8 * j; x < 8 * j + 8 && x < L; ++x) v = (k == SUM_GMAX) ? fmax(v, s_S[x][k]) : v + s_S[x][k];
            t[j] = v;
        }
        s_tot[k] = (k == SUM_GMAX) ? fmax(fmax(fmax(t[0], t[1]), t[2]), t[3]) : ((t[0] + t[1]) + t[2]) + t[3];
    }
    __syncthreads();
}

__device__ __forceinline__ void pc_reduce_and_control(PsfmSolveCtrl* __restrict__ ctrl, double* partials, int n_blocks,
                                                      int is_init, double* export_sums)
{
    pc_reduce_totals(partials, n_blocks);
    const double* s_tot = pc_totals();
    if (export_sums) {
        if (threadIdx.x < PC_NSUM) export_sums[threadIdx.x] = s_tot[threadIdx.x];
        return;
    }
    if (threadIdx.x != 0) return;
    PsfmSolveCtrl C = *ctrl;
    pc_chain_control(C, s_tot, is_init ? 0 : 1);
    C.launches += 1;
    *ctrl = C;
}

__device__ __forceinline__ void pc_writeback_tracks(const PcParams& P, const PsfmSolveCtrl& C, double* out_rows);
__device__ __forceinline__ void pc_writeback_scalars(const PcParams& P, const PsfmSolveCtrl& C);

// ------------------------------------------------------------------------------------------------
// pc_resident: the launch chain's loop inside ONE launch, with the tracks' solver state ON CHIP.  Solves that reject steps take
// 20-40 trust-region iterations.  As launches each of them is a launch gap, a tail behind the last block and a control block
// round trip; round 3's persistent form removed those but still STREAMED the tracks through every round (88 B of state in,
// 32 B of candidate out, per track and round: 46 MB at 1080p), walked a thread's 3-4 tracks one behind the other -- four
// dependent round trips each -- and re-derived the system at x that the round before had already solved (16.5 us per round, the
// SIMDs issuing VALU a quarter of the time).  Here:
//   * a thread HOLDS its tracks: slot k of thread t is entry k * PC_BLOCK + t of the block's list (pc_build_list).  Per slot in
//     registers: refs, weight, Jacobi scaling, the iterate x and its steepest-descent / Gauss-Newton directions (u, d); in LDS:
//     (u', d') at the candidate.  A round is  x' = x + a u + b d  ->  ONE gather (the taps at x', every slot's in flight
//     together)  ->  residuals, cost, the system at x' (evaluate-ahead: its sums and (u', d'))  ->  the sums.  Accepted: x <- x'
//     (recomputed, same bits), (u, d) <- (u', d'); rejected: same x, same (u, d), new (a, b).  Nothing is loaded but taps,
//     nothing stored.  Entries beyond NS slots per thread (grids larger than the register files) are streamed as the launch
//     chain does it (pc_iter_tracks), behind the slots in the same order of summation.
//   * the hand-off is an ALL-REDUCE of tagged granules, two hops: every block publishes its PC_RES_SUMS sums as 8-byte granules
//     {tag : 32 | half of a double : 32} (valid by themselves: nothing orders the stores, the poll IS the read -- form R2 of
//     MI355X_MICROARCH.md); the PC_LEADERS leaders add up their members' (pc_reduce_totals' first level) and publish theirs;
//     EVERY block adds up the leaders' (the upper levels) and runs the control step itself on its own copy of the control block --
//     same numbers, same function, same decisions everywhere; no reducer, no packet, no counter, no store acknowledgement.
//     tag = {launch epoch : 20 | round : 12}: a granule left by an earlier launch never matches.
// Same per-track functions, same order of summation, same control step as the launches: bit-identical iterates
// (tests/test_gpu_solver.py::test_launch_chain_as_one_persistent_launch_is_the_same_solve).  A block whose poll exceeds the spin
// limit (the grid was not co-resident after all) poisons its granules -- leaders pass the poison on -- and every block leaves
// without having written anything: the control block still says "not done", the write-back kernel behind raises the stall
// flag and the host redoes the solve with launches.
// ------------------------------------------------------------------------------------------------
#define PC_RES_SUMS 11          // SUM_MCC .. SUM_FAIL: what a round reduces (SUM_CNT / SUM_COST0 belong to pc_init)
#define PC_RES_ROW 32           // granules per row (256 B): 2 per sum
#define PC_RES_NS_MAX 3         // slots per thread: 3 x 27 doubles of state + the arithmetic's working set is what 256 registers
                                // and half a CU's LDS hold
#define PC_RES_BLOCKS 512       // two blocks per CU
static_assert(PC_RES_BLOCKS / PC_LEADERS / 4 <= 4, "a leader lane keeps its 4 members' granules in flight");
static_assert(sizeof(PsfmSolveCtrl) % 8 == 0, "the control block is copied as 8-byte words");
#define PC_CTRL_WORDS ((int)(sizeof(PsfmSolveCtrl) / 8))

struct PcRound { int done, cur, kind, accepted, giveup; double mu, a, b; };   // what a round of the loop needs from the control block

// ONE solve over several GPUs / processes (track-sharded runs, psfm_shard.hip): every rank runs this launch on ITS tracks and the
// second hop of the all-reduce crosses the ranks -- a leader publishes its sums into EVERY rank's leader area (its own included)
// through peer-mapped pointers (hipIpcOpenMemHandle / P2P over xGMI; plain pointers between host threads of one process), and
// every block adds up the leader rows of all ranks from its OWN rank's area, rank by rank in rank order:
//     total = (...((T_0 + T_1) + T_2) ...) + T_{world-1},   T_r = rank r's total as pc_tree_totals forms it
// -- the same numbers in the same order on every rank, hence the same control decisions everywhere, with no host and no collective
// library in the loop.  Leader area of a rank: [set 0, 1][source rank][PC_LEADERS][PC_RES_ROW] granules.  world == 1 is the
// one-GPU launch bit for bit (the template parameter only removes the loop).
struct PcPeers {
    int world, rank;
    int L[PSFM_MAX_PEERS];                       // leaders of every rank's launch: min(PC_LEADERS, its blocks)
    unsigned long long* lead[PSFM_MAX_PEERS];    // rank r's leader area as THIS process addresses it
};
__device__ __forceinline__ unsigned long long* pc_peer_rows(unsigned long long* area, int world, unsigned set, int src_rank)
{
    return area + ((size_t)(set * (unsigned)world + (unsigned)src_rank) * PC_LEADERS) * PC_RES_ROW;
}

__device__ __forceinline__ unsigned long long pc_gran_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pc_gran_store(unsigned long long* p, unsigned tag, unsigned half)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned pc_half(double v, int lo)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    return lo ? (unsigned)bits : (unsigned)(bits >> 32);
}
__device__ __forceinline__ double pc_unhalf(unsigned long long hi, unsigned long long lo)
{
    return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
}

// The all-reduce of one round, run by WAVE 0 of every block (the other waves wait at the caller's barrier): blk[] = this
// block's sums (LDS), gran = [n_blocks rows of members][PC_LEADERS rows of leaders].  Returns 0 with the totals in tot[], 1
// when the round is given up (this block timed out, or saw the poison of one that did).  Lane (k, j) = 4 k + j works on sum k.
// The all-reduce of one round, run by WAVE 0 of every block (the other waves wait at the caller's barrier): blk[] = this
// block's sums (LDS), gran = [n_blocks rows of members][PC_LEADERS rows of leaders].  Returns 0 with the totals in tot[], 1
// when the round is given up (this block timed out, or saw the poison of one that did).  Lane (k, j) = 4 k + j works on sum k.
template <int NSUMS, bool PEERS = false>
__device__ __forceinline__ int pc_res_allreduce(const double* blk, unsigned long long* gran, int n_blocks, unsigned tag, unsigned poison,
                                                int spin_limit, bool quit, double tot[PC_NSUM], int rtl, const PcPeers* peers = nullptr)
{
    static_assert(4 * NSUMS <= PSFM_WAVE && 2 * NSUMS <= PC_RES_ROW, "one lane per (sum, quarter), two granules per sum");
    const int lane = threadIdx.x;            // (wave 0: lane == thread)
    const int b = (int)blockIdx.x;
    const int L = n_blocks < PC_LEADERS ? n_blocks : PC_LEADERS;
    unsigned long long* mine = gran + (size_t)b * PC_RES_ROW;
    // leader rows: TWO sets, used by alternate rounds (consecutive tags differ in bit 0).  A leader may publish its round r + 1 sums as
    // soon as its own members are through round r -- with one set, a block under another leader that has not yet seen every leader's
    // round-r granules would find tag r + 1 there, never match, and poison the solve at its spin limit (a spurious give-up under
    // preemption).  Round r + 2 cannot be published before every block has read round r: it needs every leader's r + 1 row, which
    // needs every member's r + 1 granule, which a member only writes behind its round-r totals.
    unsigned long long* lead = PEERS ? pc_peer_rows(peers->lead[peers->rank], peers->world, tag & 1u, peers->rank)
                                     : gran + ((size_t)n_blocks + (size_t)(tag & 1u) * PC_LEADERS) * PC_RES_ROW;
    const int k = lane >> 2, j = lane & 3;
    const bool work = lane < 4 * NSUMS;
    int bad = quit ? 1 : 0;
    if (!bad && lane < 2 * NSUMS) pc_gran_store(mine + lane, tag, pc_half(blk[lane >> 1], lane & 1));
    PC_RTL(4);
    if (!bad && b < L) {
        // ---- leader: lane (k, j) adds the sums of its Q members in member order, then the four j's in order ----
        const int Q = pc_tree_q(n_blocks);      // (<= 4: PC_RES_BLOCKS / PC_LEADERS / 4)
        const int cnt = (n_blocks - b + PC_LEADERS - 1) / PC_LEADERS;
        double v = 0.0;
        for (int spins = 0;; ++spins) {
            unsigned long long wh[4], wl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = j * Q + u;
                const bool use = work && u < Q && m < cnt;
                const unsigned long long* src = gran + (size_t)(b + PC_LEADERS * (use ? m : 0)) * PC_RES_ROW + 2 * (work ? k : 0);
                wh[u] = pc_gran_load(src); wl[u] = pc_gran_load(src + 1);
            }
            bool ok = true, psn = false;
            v = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = j * Q + u;
                const bool use = work && u < Q && m < cnt;
                if (!use) continue;
                const unsigned th = (unsigned)(wh[u] >> 32), tl = (unsigned)(wl[u] >> 32);
                if (th != tag || tl != tag) { ok = false; psn = psn || th == poison || tl == poison; }
                const double val = pc_unhalf(wh[u], wl[u]);
                v = (k == SUM_GMAX) ? fmax(v, val) : v + val;
            }
            if (__ballot(psn) != 0ull) { bad = 1; break; }
            if (__ballot(!ok) == 0ull) break;
            if (spins >= spin_limit) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!bad) {
            const double s1 = __shfl(v, (lane & ~3) + 1), s2 = __shfl(v, (lane & ~3) + 2), s3 = __shfl(v, (lane & ~3) + 3);
            const double S = (k == SUM_GMAX) ? fmax(fmax(fmax(v, s1), s2), s3) : ((v + s1) + s2) + s3;   // (valid in lanes 4k)
            const double Sg = __shfl(S, 4 * (lane >> 1));        // granule lane g publishes sum g >> 1
            if (lane < 2 * NSUMS) {
                if (PEERS) {
                    for (int r = 0; r < peers->world; ++r)     // into every rank's area (this rank's slot there)
                        pc_gran_store(pc_peer_rows(peers->lead[r], peers->world, tag & 1u, peers->rank) + (size_t)b * PC_RES_ROW + lane, tag,
                                      pc_half(Sg, lane & 1));
                } else {
                    pc_gran_store(lead + (size_t)b * PC_RES_ROW + lane, tag, pc_half(Sg, lane & 1));
                }
            }
        }
        PC_RTL(5);
    }
    double total = 0.0;
    // ---- every block: lane (k, j) adds the leaders 8 j .. 8 j + 7 in order, then the four j's in order (PEERS: of every rank, rank
    //      by rank, from this rank's own area -- a rank that is late keeps everybody in its poll, bounded by the spin limit) ----
    const int n_src = PEERS ? peers->world : 1;
    for (int r = 0; r < n_src && !bad; ++r) {
        const unsigned long long* rows = PEERS ? pc_peer_rows(peers->lead[peers->rank], peers->world, tag & 1u, r) : lead;
        const int Lr = PEERS ? peers->L[r] : L;
        double t = 0.0;
        for (int spins = 0;; ++spins) {
            unsigned long long wh[8], wl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = 8 * j + u;
                const unsigned long long* src = rows + (size_t)(work && x < Lr ? x : 0) * PC_RES_ROW + 2 * (work ? k : 0);
                wh[u] = pc_gran_load(src); wl[u] = pc_gran_load(src + 1);
            }
            bool ok = true, psn = false;
            t = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x = 8 * j + u;
                if (!work || x >= Lr) continue;
                const unsigned th = (unsigned)(wh[u] >> 32), tl = (unsigned)(wl[u] >> 32);
                if (th != tag || tl != tag) { ok = false; psn = psn || th == poison || tl == poison; }
                const double S = pc_unhalf(wh[u], wl[u]);
                t = (k == SUM_GMAX) ? fmax(t, S) : t + S;
            }
            if (__ballot(psn) != 0ull) { bad = 1; break; }
            if (__ballot(!ok) == 0ull) break;
            if (spins >= spin_limit) { bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (bad) break;
        const double t1 = __shfl(t, (lane & ~3) + 1), t2 = __shfl(t, (lane & ~3) + 2), t3 = __shfl(t, (lane & ~3) + 3);
        const double Tr = (k == SUM_GMAX) ? fmax(fmax(fmax(t, t1), t2), t3) : ((t + t1) + t2) + t3;       // (valid in lanes 4k)
        total = r == 0 ? Tr : ((k == SUM_GMAX) ? fmax(total, Tr) : total + Tr);
    }
    if (bad) {
        // poison: whoever waits for this block (its leader; everybody -- on every rank -- if it is a leader) leaves at its next poll
        if (lane < 2 * NSUMS) {
            pc_gran_store(mine + lane, poison, 0u);
            if (b < L) {
                if (PEERS) {
                    for (int r = 0; r < peers->world; ++r)
                        pc_gran_store(pc_peer_rows(peers->lead[r], peers->world, tag & 1u, peers->rank) + (size_t)b * PC_RES_ROW + lane, poison, 0u);
                } else {
                    pc_gran_store(lead + (size_t)b * PC_RES_ROW + lane, poison, 0u);
                }
            }
        }
        return 1;
    }
    PC_RTL(6);
#pragma unroll
    for (int q = 0; q < PC_NSUM; ++q) tot[q] = q < NSUMS ? __shfl(total, 4 * q) : 0.0;
    return 0;
}

// A resident solve that ends without a result (its hand-off timed out: the grid was not co-resident; or it ran out of rounds) has
// written nothing.  ENQUEUED inside a sequence (raise_stall) it raises the device-side stall flag itself -- everything enqueued
// behind turns into no-ops and the host redoes the solve with launches at its checkpoint.  A launch the host POLLS behind --
// a batch solve, the redo of a stalled solve (psfm_solve_frame_resume) -- must not: it just leaves its control block "not done"
// and pc_finish_sync goes on with one launch per iteration (which return at once behind a raised flag).
__device__ __forceinline__ void pc_res_stall(const PcParams& P, int raise_stall)
{
    if (raise_stall && P.birth_frame && threadIdx.x == 0) *P.stall = P.frame + 1;
}

// (PcSlot and what happens to a slot -- pc_slot_fill / _start / _round / _refresh / _accept / _candidate: psfm_pc_resident.h, host-compilable)

template <int NS, bool PEERS = false>
__global__ __launch_bounds__(PC_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
void psfm_pc_resident_kernel(PcParams P, unsigned long long* gran, unsigned epoch, int spin_limit, int max_rounds, int quit_code,
                             int init_inside, int raise_stall, double* out_rows, PcPeers peers)
{
    // init_inside: iteration 0 (what psfm_pc_init_kernel does) is this launch's first round; else it runs behind that kernel.
    // When the loop ends with the solve done, every block writes its tracks back (what psfm_pc_writeback_kernel does) and the
    // control block says so.
    if (*P.stall) return;
    __shared__ PsfmSolveCtrl s_C;        // this block's copy of the control block (every block runs the same control step)
    __shared__ PcRound s_R;
    __shared__ double s_next[NS][8][PC_BLOCK];     // (u', d') at the candidate of the round
    __shared__ double s_ref[NS][4][PC_BLOCK];      // refs (r1, r2) of the slots
    __shared__ double s_blk[PC_NSUM];
    const int tid = threadIdx.x;
    const int nblk = (int)gridDim.x;
    const PcF2* F12 = (const PcF2*)P.flow12;
    const unsigned poison = (epoch << 12) | 0xfffu;
    const int* lst = P.list + (int64_t)blockIdx.x * P.list_pitch;
    PcSlot T[NS];
    int cnt;
    if (init_inside) {
        // ---- iteration 0: the block's list, the slots' tracks from their three buffered positions, the sums, Ceres' IterationZero ----
        if (blockIdx.x == 0 && tid == 0) {
            // (whatever gives up below must not leave an earlier solve's outcome to be read as this one's)
            P.ctrl->done = 0; P.ctrl->written = 0;
            if (P.sel) *P.sel = 0;      // (as pc_init_tracks)
        }
        const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
        cnt = pc_build_list(P, n);
        double acc[PC_NSUM];
#pragma unroll
        for (int q = 0; q < PC_NSUM; ++q) acc[q] = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (pc_slot_entry(k, tid) < cnt) {
                PcInit o;
                pc_init_entry(P, lst[pc_slot_entry(k, tid)], false, acc, o);
                pc_slot_fill(T[k], o.s, o.c, o.x, o.y);
                s_ref[k][0][tid] = o.r1.x; s_ref[k][1][tid] = o.r1.y; s_ref[k][2][tid] = o.r2.x; s_ref[k][3][tid] = o.r2.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int p = pc_stream_first(NS, tid); p < cnt; p += PC_BLOCK) {     // (streamed entries: their constants go to memory)
            PcInit o;
            pc_init_entry(P, lst[p], true, acc, o);
        }
        pc_block_sums<PC_NSUM>(acc, s_blk);
        if (tid < PSFM_WAVE) {
            double tot[PC_NSUM];
            const int bad = pc_res_allreduce<PC_NSUM, PEERS>(s_blk, gran, nblk, (epoch << 12) | 0xffeu, poison, spin_limit, false, tot, -1, &peers);
            if (tid == 0) {
                s_R.giveup = bad; s_R.accepted = 0;
                if (!bad) {
                    PsfmSolveCtrl& C = s_C;          // (in place, in LDS: a private copy would not stay in registers behind the memset)
                    pc_chain_control(C, tot, 0);
                    C.launches = 1;
                    s_R.done = C.done; s_R.cur = C.cur; s_R.kind = C.kind_next; s_R.mu = C.mu; s_R.a = C.dl_a; s_R.b = C.dl_b;
                }
            }
        }
        __syncthreads();
        if (s_R.giveup) { pc_res_stall(P, raise_stall); return; }
    } else {
        if (tid < PC_CTRL_WORDS) ((unsigned long long*)&s_C)[tid] = ((const unsigned long long*)P.ctrl)[tid];    // (written by the launch in front)
        __syncthreads();
        if (tid == 0) {
            s_R.done = s_C.done; s_R.cur = s_C.cur; s_R.kind = s_C.kind_next; s_R.mu = s_C.mu; s_R.a = s_C.dl_a; s_R.b = s_C.dl_b;
            s_R.accepted = 0; s_R.giveup = 0;
        }
        __syncthreads();
        cnt = P.list_n[blockIdx.x];
        if (!s_R.done) {
            // ---- the slots: constants and the start values from memory (once), the system at x0 for the mu in force ----
            // (an empty slot loads lane 0's state and computes on it; nothing of it is ever added or stored)
            PcTaps tp[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = pc_slot_entry(k, tid) < cnt ? lst[pc_slot_entry(k, tid)] : 0;
                const double2 r1 = P.ref1[i], r2 = P.ref2[i], js = P.jscale[i], p1 = P.x1a[i], p2 = P.x2a[i];
                T[k].s = P.scale[i];
                T[k].S0q = js.x; T[k].S1q = js.y;
                s_ref[k][0][tid] = r1.x; s_ref[k][1][tid] = r1.y; s_ref[k][2][tid] = r2.x; s_ref[k][3][tid] = r2.y;
                T[k].x[0] = p1.x; T[k].x[1] = p1.y; T[k].x[2] = p2.x; T[k].x[3] = p2.y;
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) tp[k] = pc_core_taps<PC_ITER_PAIR>(F12, P.H, P.W, T[k].x);
#pragma unroll
            for (int k = 0; k < NS; ++k)
                pc_slot_start(T[k], tp[k], s_ref[k][0][tid], s_ref[k][1][tid], s_ref[k][2][tid], s_ref[k][3][tid], s_R.mu);
        }
    }
    for (unsigned it = 0; it < (unsigned)max_rounds; ++it) {
        if (s_R.done) break;
        const double mu = s_R.mu, a = s_R.a, b = s_R.b;
        const int cur = s_R.cur;
        const bool refresh = s_R.kind != 0;
#ifdef PSFM_TIMELINE
        const int rtl = (epoch == PC_RTL_EPOCH && it >= PC_RTL_R0 && it < PC_RTL_R0 + 8u) ? (int)(it - PC_RTL_R0) : -1;
        if (rtl >= 0 && tid == 0 && blockIdx.x == 0) g_pc_tl_n = rtl + 1;
#else
        const int rtl = -1;
#endif
        PC_RTL(0);
        double acc[PC_NSUM];
#pragma unroll
        for (int q = 0; q < PC_NSUM; ++q) acc[q] = 0.0;
        if (!refresh) {
            // the candidate of (a, b), its cost and -- ahead of the decision -- the system there
            const double mu_next = pc_mu_next(mu);
            PcTaps tp[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {     // every slot's taps in flight together
                double xe[4];
                (void)pc_slot_candidate(T[k], a, b, xe);
                tp[k] = pc_core_taps<PC_ITER_PAIR>(F12, P.H, P.W, xe);
            }
            PC_RTL(1);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                // (a lane without a track in this slot skips it; the sums receive ONE term per track, in list order, like the launches')
                if (pc_slot_entry(k, tid) < cnt) {
                    // (pc_slot_round recomputes the candidate: the same bits, 8 registers fewer across the gather)
                    double nx[8];
                    pc_slot_round(T[k], tp[k], s_ref[k][0][tid], s_ref[k][1][tid], s_ref[k][2][tid], s_ref[k][3][tid], a, b, mu_next, acc, nx);
#pragma unroll
                    for (int q = 0; q < 8; ++q) s_next[k][q][tid] = nx[q];
                }
                __builtin_amdgcn_sched_barrier(0);      // (one slot at a time: interleaving them costs more registers than it hides)
            }
        } else {
            // the system at x for the mu an invalid step has raised: (u, d) of the slots are replaced (rare)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (pc_slot_entry(k, tid) < cnt)
                    pc_slot_refresh<PC_ITER_PAIR>(T[k], F12, P.H, P.W, s_ref[k][0][tid], s_ref[k][1][tid], s_ref[k][2][tid], s_ref[k][3][tid], mu, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // entries beyond the slots: streamed, as the launches do it
        if (cnt > NS * PC_BLOCK) pc_iter_tracks(P, pc_stream_first(NS, tid), cur, mu, a, b, refresh, acc);
        PC_RTL(2);
        pc_block_sums<PC_RES_SUMS>(acc, s_blk);
        PC_RTL(3);
        if (tid < PSFM_WAVE) {
            double tot[PC_NSUM];
            const bool quit = quit_code != 0 && (quit_code >> 16) == (int)blockIdx.x + 1 && (quit_code & 0xffff) == (int)it;
            const int bad = pc_res_allreduce<PC_RES_SUMS, PEERS>(s_blk, gran, nblk, (epoch << 12) | (it + 1u), poison, spin_limit, quit, tot, rtl, &peers);
            PC_RTL(9);
            // what the control step may need from the totals (pc_derive), one quantity per lane instead of one behind the other
            PcDerived D;
            {
                const int lane = tid;
                const double rad = lane == 0 ? tot[SUM_STEP2] : (lane == 1 ? tot[SUM_XN2] : (lane == 2 ? tot[SUM_G2] : tot[SUM_GN2]));
                const double num = lane == 4 ? tot[SUM_G2] : s_C.x_cost - tot[SUM_COST];
                const double den = lane == 4 ? tot[SUM_JG2] : s_C.mcc;
                const double res = lane < 4 ? sqrt(rad) : num / den;
                D.step_norm = __shfl(res, 0); D.x_norm = __shfl(res, 1); D.gnorm = __shfl(res, 2); D.gnn = __shfl(res, 3);
                D.alpha = __shfl(res, 4); D.rho = __shfl(res, 5);
            }
            PC_RTL(10);
            if (tid == 0) {
                if (bad) s_R.giveup = 1;
                else {
                    // (in place, in LDS; run by every lane of the wave on private copies instead -- uniform branches, no one-lane exec
                    // mask -- it takes the same 1.3 us: a dependent chain of ~150 f64 operations, a few of them square roots / quotients)
                    PsfmSolveCtrl& C = s_C;
                    const int cur0 = C.cur;
                    pc_chain_control_d(C, tot, D, 1);
                    C.launches += 1;
                    s_R.accepted = C.cur != cur0;
                    s_R.done = C.done; s_R.cur = C.cur; s_R.kind = C.kind_next; s_R.mu = C.mu; s_R.a = C.dl_a; s_R.b = C.dl_b;
                }
            }
            PC_RTL(7);
        }
        __syncthreads();
        PC_RTL(8);
        if (s_R.giveup) { pc_res_stall(P, raise_stall); return; }
        if (s_R.accepted) {
            // x <- the candidate (the same operations: the same bits), (u, d) <- what was solved there
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                double nx[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) nx[q] = s_next[k][q][tid];
                pc_slot_accept(T[k], a, b, nx);
            }
        }
    }
    if (!s_R.done) { pc_res_stall(P, raise_stall); return; }            // (ran out of rounds)
    // ---- write-back (block 0 also: statistics, the control block) ----
    const PsfmSolveCtrl C = s_C;
    const bool moved = pc_res_moved(C);                  // (a failed solve hands the parameters back as they came in)
    if (moved || out_rows) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (pc_slot_entry(k, tid) >= cnt) continue;
            const int i = lst[pc_slot_entry(k, tid)];
            double2 p1 = make_double2(T[k].x[0], T[k].x[1]), p2 = make_double2(T[k].x[2], T[k].x[3]);
            if (!moved) { p1 = P.x1a[i]; p2 = P.x2a[i]; }
            if (out_rows) {
                out_rows[4 * (int64_t)i + 0] = p1.x; out_rows[4 * (int64_t)i + 1] = p1.y;
                out_rows[4 * (int64_t)i + 2] = p2.x; out_rows[4 * (int64_t)i + 3] = p2.y;
            } else {
                P.x1a[i] = p1; P.x2a[i] = p2;
            }
        }
        const int src = pc_res_stream_source(C);
        const double2* xc1 = pc_buf1(P, src);
        const double2* xc2 = pc_buf2(P, src);
        for (int p = pc_stream_first(NS, tid); p < cnt; p += PC_BLOCK) {
            const int i = lst[p];
            const double2 p1 = xc1[i], p2 = xc2[i];
            if (out_rows) {
                out_rows[4 * (int64_t)i + 0] = p1.x; out_rows[4 * (int64_t)i + 1] = p1.y;
                out_rows[4 * (int64_t)i + 2] = p2.x; out_rows[4 * (int64_t)i + 3] = p2.y;
            } else if (src != 0) {
                P.x1a[i] = p1; P.x2a[i] = p2;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        pc_writeback_scalars(P, C);
        PsfmSolveCtrl Cw = C;
        Cw.written = 1;
        *P.ctrl = Cw;
    }
}

// ------------------------------------------------------------------------------------------------
// pc_fused: the WHOLE solve of a frame in one launch, speculating that every trust-region iteration takes the pure
// Gauss-Newton step at mu = min_mu and is accepted -- which is what Ceres does on every solve that converges without a
// rejection.  Under that assumption nothing a track computes depends on the global scalars: thread = track runs K
// iterations in registers (the residuals / Jacobian evaluated at a candidate for its cost ARE the next iteration's
// evaluation at x: K + 1 interpolations instead of 2 K), stores the iterate after m accepted steps in buffer m, and
// every iteration's 13 sums are reduced block -> group of PC_GROUP blocks -> grid (two tickets, fixed order).  The last
// block replays Ceres' control flow over iterations 1..K with those sums, exactly as the launch chain would have
// (pc_control_step), and stops believing the speculation at the first decision that is not "Gauss-Newton step
// accepted": termination -> done, the accepted iterate is buffer `cur` (PsfmCounters::sel tells the next chain step, which
// copies it into the log on its way); anything else (rejection, dogleg interpolation, invalid step, more than K
// iterations) raises the stall flag and the host redoes this solve with the launch chain from the untouched buffer 0.
// ------------------------------------------------------------------------------------------------
// Where iterate m (the position after m accepted steps) of a fused solve lives.  A solve that goes as speculated accepts
// e = k_first - 1 steps and terminates in iteration k_first, so iterate e is written straight into buffer 0 -- the log
// slabs, where the next chain step and finalize read -- and the values the solve started from are kept in buffer e
// instead; every other iterate m sits in buffer m.
__device__ __forceinline__ int pc_phys(int m, int e) { return m == e ? 0 : (m == 0 ? e : m); }

// Ceres' control flow replayed over the sums of `n_it` speculated iterations (thread 0 of the last block, or the control
// kernel of a track-sharded run): tot[j * PC_NSUM + k].  first: the launch that started the solve (iteration 1 ..), else
// a continuation behind `C.cur` accepted steps.  pc != NULL (device-paced sequence): the outcome also says what the NEXT
// launch does -- the next frame, or more iterations of this solve when every one so far was accepted and none terminated.
__device__ __forceinline__ void pc_fused_replay(const PcParams& P, const double* tot, int n_it, bool first, PsfmCounters* pc,
                                                int launch_id, const PsfmSolveCtrl* c_in = nullptr)
{
    PsfmSolveCtrl C = c_in ? *c_in : *P.ctrl;      // (c_in: the caller's copy of the control block, loaded with everything else it needs)
    const int base = first ? 0 : C.cur;
    const int k_first = first ? n_it : C.k_first;
    for (int j = 1; j <= n_it; ++j) {
        pc_control_step(C, tot + (j - 1) * PC_NSUM, first && j == 1, base + j);
        if (C.done) break;
        // iteration j + 1 was computed at iterate base + j with the Gauss-Newton step at min_mu: only valid behind an accepted step
        if (!(C.fresh_x && C.cur == base + j && C.dl_fixed == 0 && C.mu == 1e-8)) break;
    }
    C.k_first = k_first;
    C.launches += 1;
    *P.ctrl = C;
    const int e = k_first - 1;
    if (!C.done) {
        const bool all_accepted = C.fresh_x && C.cur == base + n_it && C.dl_fixed == 0 && C.mu == 1e-8;
        if (pc && all_accepted && C.cur + 1 <= PC_KMAX) {     // device-paced: the next launch continues this solve
            pc->pc_phase = 1;
            pc->pc_owner = launch_id + 1;
            return;
        }
        // not what was speculated: the launch chain redoes this solve from the values it started from,
        *P.sel = pc_phys(0, e);   // which psfm_solve_frame_resume first moves back into buffer 0
        *P.stall = P.frame + 1;
        return;
    }
    // (a failed solve hands the parameters back as they came in -- iterate 0 -- whatever it had accepted on the way: see the write-back)
    *P.sel = pc_phys(C.failed ? 0 : C.cur, e);
    if (P.stats_dev) {
        psfm_solve_stats st;
        st.iterations = C.iteration; st.successful_steps = C.successful;
        st.termination = C.n_tracks == 0 ? -1 : C.termination; st.dogleg_nonGN = C.nonGN;
        st.initial_cost = C.initial_cost; st.final_cost = C.x_cost;
        if (C.failed) st.termination = PSFM_TERM_FAILURE;
        P.stats_dev[P.frame] = st;
    }
    if (pc) { pc->pc_frame = P.frame + 1; pc->pc_phase = 0; pc->pc_owner = launch_id + 1; }
}

struct PcTrack {            // one track's solve state in registers
    double x[4], r[6], jac[4];
    double2 r1, r2;
    PcConst c;
    double iA22;            // 1 / (H22 (1 + mu)) at the mu the fused solve speculates
};

// (Round 4 tried a tap NEIGHBOURHOOD per track in LDS -- 3 rows x 4 columns of F12 around the start position, brought in by six
// global_load_lds_dwordx4 per lane at the setup, so that the K evaluations behind it take their taps from LDS instead of a dependent
// gather each: bit-identical, and 2 % SLOWER (frame kernel 66.7 vs 65.4 us at 1080p).  The taps of an iterate lie in the cache
// lines its predecessor touched: those gathers are L1 hits, not round trips to L2 / HBM.  profiles/EXPERIMENTS.md 6.5.)
// residuals and Jacobian of T at T.x
__device__ __forceinline__ void pc_track_eval(const PcParams& P, PcTrack& T)
{
    const PcTaps t = pc_core_taps<PC_FUSED_PAIR>((const PcF2*)P.flow12, P.H, P.W, T.x);
    pc_core_eval_taps(t, T.x, T.r1.x, T.r1.y, T.r2.x, T.r2.y, T.c.s, T.r, T.jac);
}

// sums of one trust-region iteration at T.x (already evaluated: T.r, T.jac) into v[]; the candidate goes to (xn1, xn2)[i]
// and becomes T.x, evaluated.  The same function as the launch chain's iteration, with (a, b) = (0, 1) at compile time.
__device__ __forceinline__ void pc_fused_iteration(const PcParams& P, PcTrack& T, double mu, double2* xn1, double2* xn2, int i,
                                                   double v[PC_NSUM])
{
    double xp[4];
    pc_core_iteration<true>(T.x, T.r, T.jac, T.c, mu, T.iA22, 0.0, 1.0, v, xp);
    if (xn1) {
        xn1[i] = make_double2(xp[0], xp[1]);
        xn2[i] = make_double2(xp[2], xp[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) T.x[k] = xp[k];
    pc_track_eval(P, T);
    v[SUM_COST] += pc_core_cost(T.r);
}

// Sums of one iteration over the 64 tracks of a WAVE, without a block barrier (the four waves of a block stay independent
// inside the iteration loop).  Round 4: the transposed exchange tree of psfm_pc_reduce.h inside each 16-lane row (registers, DPP),
// the four row totals of a slot through 450 bytes of LDS per wave, added in row order by lane k < 13.  Fixed order.  (Rounds 2-3
// parked every lane's 13 values in LDS -- 27 KB per block, the space the tap neighbourhoods below need; PC_WAVE_PARK keeps
// that form for A/B builds.)  LDS operations of one wave execute in order, so the wave only has to wait for its own stores.
#define PC_NW (PC_BLOCK / PSFM_WAVE)
#ifndef PC_WAVE_PARK
struct PcWaveRed {
    double row[PC_NW][4][PC_NSUM + 1];
    double wsum[PC_NW][PC_KMAX][PC_NSUM];
};
__device__ __forceinline__ void pc_wave_reduce(PcWaveRed& R, const double v[PC_NSUM], int iter)
{
    const int lane = threadIdx.x & (PSFM_WAVE - 1), w = threadIdx.x / PSFM_WAVE;
    int slot;
    const double g = pc_row_tree<PC_NSUM>(v, slot);
    R.row[w][lane >> 4][slot] = g;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < PC_NSUM) {
        const double q0 = R.row[w][0][lane], q1 = R.row[w][1][lane], q2 = R.row[w][2][lane], q3 = R.row[w][3][lane];
        R.wsum[w][iter][lane] = (lane == SUM_GMAX) ? fmax(fmax(fmax(q0, q1), q2), q3) : ((q0 + q1) + q2) + q3;
    }
    // (the next iteration's row stores come behind these loads in the wave's LDS queue)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#else
#define PC_WROW 66          // row pitch in doubles: 2-way instead of 13-way bank conflicts in the quarter sums
struct PcWaveRed {
    double park[PC_NW][PC_NSUM][PC_WROW];
    double quarter[PC_NW][PC_NSUM][4];
    double wsum[PC_NW][PC_KMAX][PC_NSUM];
};
__device__ __forceinline__ void pc_wave_reduce(PcWaveRed& R, const double v[PC_NSUM], int iter)
{
    const int lane = threadIdx.x & (PSFM_WAVE - 1), w = threadIdx.x / PSFM_WAVE;
#pragma unroll
    for (int k = 0; k < PC_NSUM; ++k) R.park[w][k][lane] = v[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (the max slot on its own branch: one select per addition otherwise -- 4 instructions per step instead of 1)
    const int k = lane >> 2, q = lane & 3;
    if (k < PC_NSUM && k != SUM_GMAX) {
        double t = R.park[w][k][q];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += R.park[w][k][4 * j + q];
        R.quarter[w][k][q] = t;
    } else if (k == SUM_GMAX) {
        double t = R.park[w][SUM_GMAX][q];
#pragma unroll
        for (int j = 1; j < 16; ++j) t = fmax(t, R.park[w][SUM_GMAX][4 * j + q]);
        R.quarter[w][SUM_GMAX][q] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < PC_NSUM) {
        const double q0 = R.quarter[w][lane][0], q1 = R.quarter[w][lane][1], q2 = R.quarter[w][lane][2], q3 = R.quarter[w][lane][3];
        R.wsum[w][iter][lane] = (lane == SUM_GMAX) ? fmax(fmax(fmax(q0, q1), q2), q3) : ((q0 + q1) + q2) + q3;
    }
    // (the next iteration's park stores come behind these loads in the wave's LDS queue)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif

// drain this block's write-through stores, then take a ticket: true for the last of `members` arrivals (which also
// resets the counter for the next launch)
__device__ __forceinline__ bool pc_last_arrival(unsigned* ticket, unsigned members)
{
    __shared__ int s_last2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last2 = (t == members - 1u);
        if (s_last2) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last2 != 0;
}

// (one LDS instance for every body that reduces per wave: a kernel that contains two of them must not pay for two)
__device__ __forceinline__ PcWaveRed& pc_shared_red()
{
    __shared__ PcWaveRed s_red_storage;
    return s_red_storage;
}

// refs / scale (trajectory.py:173-183, as in psfm_pc_init_kernel) and the Jacobi scaling of a track: from its p0 and
// the values x0 its solve starts from.  Leaves T.x = x0 evaluated (T.r, T.jac); returns the cost at x0.
__device__ __forceinline__ double pc_track_setup(const PcParams& P, PcTrack& T, double2 p0, double2 p1, double2 p2, double mu)
{
    const PsfmTaps t = psfm_taps((float)p0.x, (float)p0.y, P.cw, P.ch, P.H, P.W);
    const PsfmTapIdx k = psfm_tap_idx(P.H, P.W, t);
    const float2 f01 = psfm_sample_flow(P.flow01, k, t);
    const float2 f02 = psfm_sample_flow(P.flow02, k, t);
    const float o02 = psfm_sample_mask(P.occ02, k, t);
    const float nrm = sqrtf(__fadd_rn(__fmul_rn(f02.x, f02.x), __fmul_rn(f02.y, f02.y)));
    const float sf = __fmul_rn(__fsub_rn(1.0f, o02), nrm < 20.0f ? 1.0f : 0.0f);
    T.r1 = make_double2(p0.x + (double)f01.x, p0.y + (double)f01.y);
    T.r2 = make_double2(p0.x + (double)f02.x, p0.y + (double)f02.y);
    T.x[0] = p1.x; T.x[1] = p1.y; T.x[2] = p2.x; T.x[3] = p2.y;
    T.c.s = (double)sf;
    pc_track_eval(P, T);
    // Jacobi scaling from the Jacobian at x0 (what psfm_pc_init_kernel stores in P.jscale)
    T.c = pc_core_const((double)sf, T.jac);
    T.iA22 = pc_core_iA22(T.c, mu);
    return pc_core_cost(T.r);
}

// The block's sums of n_it iterations (the waves' sums, in wave order) -> its partial row; then the two tickets: the
// last block of each group of PC_GROUP consecutive blocks adds the group's rows, the last group adds the group sums in
// group order.  Returns the totals ([n_it][PC_NSUM], in LDS) in the LAST block of the launch, NULL in every other block.
// n_active: the blocks of the launch that come here.
__device__ __forceinline__ const double* pc_grid_totals(const PcParams& P, PcWaveRed& s_red, int n_it, int n_active)
{
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid < n_it * PC_NSUM) {   // the block's sums: its waves in order
        const int j = tid / PC_NSUM, k = tid - j * PC_NSUM;
        double t = s_red.wsum[0][j][k];
#pragma unroll
        for (int w = 1; w < PC_NW; ++w) t = (k == SUM_GMAX) ? fmax(t, s_red.wsum[w][j][k]) : t + s_red.wsum[w][j][k];
        __hip_atomic_store(&P.partials[((int64_t)blockIdx.x * PC_KMAX + j) * PC_NSUM + k], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- level 1 ----
    const int n_groups = (n_active + PC_GROUP - 1) / PC_GROUP;
    const int grp = blockIdx.x / PC_GROUP;
    const int members = min(PC_GROUP, n_active - grp * PC_GROUP);
    if (!pc_last_arrival(P.gticket + 1 + grp, (unsigned)members)) return nullptr;
    __shared__ double s_half[2][PC_KMAX * PC_NSUM];
    __shared__ double s_tot[PC_KMAX][PC_NSUM];
    const int slot = tid & 127, half = tid >> 7;      // slot = j * PC_NSUM + k
    const int nslot = n_it * PC_NSUM;
    {
        double t = 0.0;
        if (slot < nslot) {
            const int j = slot / PC_NSUM, k = slot - j * PC_NSUM;
            double pv[PC_GROUP / 2];
#pragma unroll
            for (int u = 0; u < PC_GROUP / 2; ++u) {
                const int b = half * (PC_GROUP / 2) + u;
                pv[u] = b < members ? __hip_atomic_load(&P.partials[((int64_t)(grp * PC_GROUP + b) * PC_KMAX + j) * PC_NSUM + k],
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                    : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PC_GROUP / 2; ++u) t = (k == SUM_GMAX) ? fmax(t, pv[u]) : t + pv[u];
            s_half[half][slot] = t;
        }
        __syncthreads();
        if (tid < nslot) {
            const int j = tid / PC_NSUM, k = tid - j * PC_NSUM;
            const double a = s_half[0][tid], b = s_half[1][tid];
            __hip_atomic_store(&P.gpart[((int64_t)grp * PC_KMAX + j) * PC_NSUM + k], (k == SUM_GMAX) ? fmax(a, b) : a + b,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- level 2 ----
    if (!pc_last_arrival(P.gticket, (unsigned)n_groups)) return nullptr;
    {
        double t = 0.0;
        if (slot < nslot) {
            const int j = slot / PC_NSUM, k = slot - j * PC_NSUM;
            for (int g0 = half * 16; g0 < n_groups; g0 += 32) {
                double pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    pv[u] = g0 + u < n_groups ? __hip_atomic_load(&P.gpart[((int64_t)(g0 + u) * PC_KMAX + j) * PC_NSUM + k],
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                              : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) t = (k == SUM_GMAX) ? fmax(t, pv[u]) : t + pv[u];
            }
            s_half[half][slot] = t;
        }
        __syncthreads();
        if (tid < nslot) {
            const int j = tid / PC_NSUM, k = tid - j * PC_NSUM;
            const double a = s_half[0][tid], b = s_half[1][tid];
            s_tot[j][k] = (k == SUM_GMAX) ? fmax(a, b) : a + b;
        }
        __syncthreads();
    }
    return &s_tot[0][0];
}

// The solve of lane i = blockIdx.x * PC_BLOCK + threadIdx.x (if `part`) from its three buffered positions, the sums of
// the launch and -- in its last block -- the control flow.  n_active: the blocks of the launch that come here (every one
// of them, whatever its lanes do: the tickets count them).  snap_next / n_lanes_live: the merged frame kernel leaves the
// lane count for the next frame's launch.  pc / launch_id: device-paced sequence (see pc_fused_replay); there the K-th
// candidate is stored too (a continuation launch starts from it).
__device__ __forceinline__ void pc_fused_body(const PcParams& P, bool part, double2 p0, double2 p1, double2 p2, int n_active,
                                              int* snap_next, const int* n_lanes_live, PsfmCounters* pc, int launch_id, int lane0)
{
    const int K = P.K;
    const int tid = threadIdx.x;
    const int i = lane0 + tid;          // (lane0: first lane of the block's tile -- XCD-banded in the merged kernels, psfm_xcd_tile)
    const double mu = 1e-8;
    const int e = K - 1;            // (pc_phys: iterate e goes straight into the log slabs, the start values into buffer e)
    PcWaveRed& s_red = pc_shared_red();
    PcTrack T;
    double c0 = 0.0;
    if (part) {
        if (e != 0) { pc_buf1(P, e)[i] = p1; pc_buf2(P, e)[i] = p2; }
        c0 = pc_track_setup(P, T, p0, p1, p2, mu);
    }
    for (int j = 0; j < K; ++j) {
        double v[PC_NSUM];
#pragma unroll
        for (int q = 0; q < PC_NSUM; ++q) v[q] = 0.0;
        if (part) {
            // (host-paced: the K-th candidate is never read -- were it accepted, the solve would not be over and is redone)
            const bool keep = j + 1 < K || (pc != nullptr && K < PC_KMAX);
            const int m = pc_phys(j + 1, e);
            pc_fused_iteration(P, T, mu, keep ? pc_buf1(P, m) : nullptr, keep ? pc_buf2(P, m) : nullptr, i, v);
            if (j == 0) { v[SUM_CNT] = 1.0; v[SUM_COST0] = c0; }
        }
        pc_wave_reduce(s_red, v, j);
    }
    const double* tot = pc_grid_totals(P, s_red, K, n_active);
    if (!tot) return;
    if (P.export_sums) {      // track-sharded: the totals of THIS process; the control step follows the ranks' exchange
        if (tid < K * PC_NSUM) P.export_sums[tid] = tot[tid];
        if (tid == 0 && snap_next) *snap_next = *n_lanes_live;   // (merged frame launch: every block is past its chain step)
        return;
    }
    if (tid != 0) return;
    if (snap_next) *snap_next = *n_lanes_live;   // (every block is past its chain step: the count is final for this launch)
    pc_fused_replay(P, tot, K, true, pc, launch_id);
}

// Device-paced sequence: MORE iterations of the solve of P.frame, behind a launch whose K iterations were all accepted
// without terminating it.  The track state is rebuilt from p0 and the start values (kept in buffer k_first - 1), the
// iterate is read from where the previous launch left it, up to two more iterations are speculated the same way.
__device__ __forceinline__ void pc_more_body(const PcParams& P, int n_active, PsfmCounters* pc, int launch_id)
{
    const PsfmSolveCtrl C0 = *P.ctrl;
    const int e = C0.k_first - 1, base = C0.cur;
    const int n_it = min(2, PC_KMAX - base);
    const int n = min(*P.n_lanes_ptr, P.n_rows);
    const int tid = threadIdx.x;
    const int i = blockIdx.x * PC_BLOCK + tid;
    const double mu = 1e-8;
    PcWaveRed& s_red = pc_shared_red();
    const bool part = pc_participates(P, i, n);
    PcTrack T;
    if (part) {
        const double2 p0 = P.p0[i];
        const double2 s1 = pc_buf1(P, pc_phys(0, e))[i], s2 = pc_buf2(P, pc_phys(0, e))[i];      // the start values
        (void)pc_track_setup(P, T, p0, s1, s2, mu);
        const double2 c1 = pc_buf1(P, pc_phys(base, e))[i], c2 = pc_buf2(P, pc_phys(base, e))[i];  // the current iterate
        T.x[0] = c1.x; T.x[1] = c1.y; T.x[2] = c2.x; T.x[3] = c2.y;
        pc_track_eval(P, T);
    }
    for (int j = 0; j < n_it; ++j) {
        double v[PC_NSUM];
#pragma unroll
        for (int q = 0; q < PC_NSUM; ++q) v[q] = 0.0;
        if (part) {
            const int m = pc_phys(base + j + 1, e);
            const bool keep = base + j + 1 <= PC_KMAX;
            pc_fused_iteration(P, T, mu, keep ? pc_buf1(P, m) : nullptr, keep ? pc_buf2(P, m) : nullptr, i, v);
        }
        pc_wave_reduce(s_red, v, j);
    }
    const double* tot = pc_grid_totals(P, s_red, n_it, n_active);
    if (!tot || tid != 0) return;
    pc_fused_replay(P, tot, n_it, false, pc, launch_id);
}

template <int WAVES>
__global__ __launch_bounds__(PC_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void psfm_pc_fused_kernel(PcParams P)
{
    if (*P.stall) return;
    P.K = P.K < 1 ? 1 : (P.K > PC_KMAX ? PC_KMAX : P.K);      // (the host clamps it too: stated here, the range spares the iteration
                                                             // loop 37 spilled VGPRs at 4 waves per SIMD)
    const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
    const int n_active = (n + PC_BLOCK - 1) / PC_BLOCK;        // blocks with a lane below the high-water mark
    if ((int)blockIdx.x >= n_active) return;
    const int i = blockIdx.x * PC_BLOCK + threadIdx.x;
    // (state loaded alongside the birth frame that decides whether the lane takes part: one round trip less)
    double2 p0 = make_double2(0.0, 0.0), p1 = p0, p2 = p0;
    if (i < n) { p0 = P.p0[i]; p1 = P.x1a[i]; p2 = P.x2a[i]; }
    pc_fused_body(P, pc_participates(P, i, n), p0, p1, p2, n_active, nullptr, nullptr, nullptr, 0, (int)blockIdx.x * PC_BLOCK);
}

// ------------------------------------------------------------------------------------------------
// The merged frame kernel of track_optimize (track_optimize.py:31-50, one loop iteration = ONE launch): the chain step
// of the frame for the block's lanes (births, step, marks, records -- psfm_chain_step_body) and, for the tracks that
// survive it with three buffered points, the whole path-consistency solve (pc_fused_body) -- their tail and next position
// go from the step straight into the solve, the taps of both share a round trip, one launch boundary and one pass over
// the lane state disappear.  Every block below the lane snapshot / the grid takes part in the solve's tickets.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(PC_BLOCK) __attribute__((amdgpu_waves_per_eu(PSFM_FRAME_WAVES, PSFM_FRAME_WAVES)))
void psfm_frame_kernel(PsfmChainArgs a, PcParams P)
{
    P.K = P.K < 1 ? 1 : (P.K > PC_KMAX ? PC_KMAX : P.K);      // (clamped by the host already: the stated range keeps the loop from spilling)
    PsfmChainOut o;
    if (!psfm_chain_step_body<R, true, true>(a, o)) return;
    // (clamped to the launch's grid: a lane table that ran full leaves a snapshot beyond it -- the host ends such a run at its next
    // checkpoint, PSFM_ERR_CAPACITY; until then no launch may count on blocks that do not exist)
    const int n_active = min((max(a.ctr->n_lanes_snap[a.frame & 1], a.Gband) + PC_BLOCK - 1) / PC_BLOCK, (int)gridDim.x);
    pc_fused_body(P, o.solve, o.p0, o.p1, o.p2, n_active, &a.ctr->n_lanes_snap[(a.frame + 1) & 1], &a.ctr->n_lanes, nullptr, 0, o.tile);
}

// What changes from one frame to the next in the solver's arguments (frame mode)
__host__ __device__ inline void pc_params_rebase(PcParams& P, const PsfmSeqStride& st, int64_t occ2_stride, int f)
{
    const int64_t df = (int64_t)f - P.frame;
    P.p0 += df * st.cap; P.x1a += df * st.cap; P.x2a += df * st.cap;
    P.flow01 += df * st.flow; P.flow12 += df * st.flow; P.flow02 += df * st.flow;
    P.occ02 += df * occ2_stride;
    P.frame = f;
    P.max_birth = f - 1;
}

// ------------------------------------------------------------------------------------------------
// The device-paced sequence of track_optimize: every launch is the same kernel with the same arguments (those of frame
// 1 + the strides of the per-frame stacks) and does "the next thing" the device-side program counter names --
//   phase 0  the frame kernel of pc_frame (chain step + fused solve with solve_K iterations), or
//   phase 1  more iterations of that frame's solve, when all of them were accepted and none terminated it
// -- and its control thread moves the counter on.  The host only keeps launches in the queue and looks at the counter
// at its checkpoints: a solve that needs one iteration more than the sequence's usual costs one more launch instead of
// a drained queue, a host-side redo and the no-op launches behind a stall flag.  (Solves that reject a step or leave the
// Gauss-Newton path still raise the stall flag and are redone by the launch chain.)
// launch_id: a block that is dispatched after the control thread has moved the counter on (only blocks beyond the lane
// snapshot can be) sees pc_owner != launch_id and leaves.
// ------------------------------------------------------------------------------------------------
// what one block of a device-paced launch does for ITS sequence (psfm_seq_kernel: the launch's only sequence; psfm_seq_batch_kernel:
// sequence blockIdx.y of the batch)
template <int R>
__device__ __forceinline__ void psfm_seq_body(PsfmChainArgs a, PcParams P, const PsfmSeqStride st, int64_t occ2_stride, int n_flows, int launch_id)
{
    PsfmCounters* ctr = a.ctr;
    if (ctr->stall || ctr->pc_owner != launch_id) return;
    const int f = ctr->pc_frame, phase = ctr->pc_phase;
    if (f >= n_flows) return;
    psfm_chain_args_rebase(a, st, f);
    pc_params_rebase(P, st, occ2_stride, f);
    const int k = ctr->solve_K;
    P.K = k < 1 ? 1 : (k > PC_KMAX ? PC_KMAX : k);
    const int n_active = min((max(ctr->n_lanes_snap[f & 1], a.Gband) + PC_BLOCK - 1) / PC_BLOCK, (int)gridDim.x);   // (as in psfm_frame_kernel)
    if (phase == 0) {
        PsfmChainOut o;
        if (!psfm_chain_step_body<R, true, true>(a, o)) return;
        pc_fused_body(P, o.solve, o.p0, o.p1, o.p2, n_active, &ctr->n_lanes_snap[(f + 1) & 1], &ctr->n_lanes, ctr, launch_id, o.tile);
    } else {
        if ((int)blockIdx.x >= n_active) return;
        pc_more_body(P, n_active, ctr, launch_id);
    }
}

template <int R, int WAVES>
__global__ __launch_bounds__(PC_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void psfm_seq_kernel(PsfmChainArgs a, PcParams P, PsfmSeqStride st, int64_t occ2_stride, int n_flows, int launch_id)
{
    psfm_seq_body<R>(a, P, st, occ2_stride, n_flows, launch_id);
}

// ------------------------------------------------------------------------------------------------
// B same-shape sequences per launch (psfm_connect_batch; run_particlesfm.py:168-176 walks a directory of sequences): a frame of a
// DAVIS / Sintel / ScanNet-sized sequence is a few hundred blocks -- 10-40 % of the device's block slots -- for one dependent chain
// of round trips.  blockIdx.y = sequence: every sequence keeps its own context (lanes, log, counters, tickets, control block), its
// own device-side program counter and K, so the sequences advance independently -- one may be continuing a solve while its
// neighbours take their next frame, a stalled or finished one turns its blocks into no-ops -- and the launches fill the machine.
// gridDim.x is a multiple of 8: block (x, y) sits on XCD x % 8, as psfm_xcd_tile assumes.
// ------------------------------------------------------------------------------------------------
struct PsfmBatchSeqOpt { PsfmChainArgs a; PcParams P; PsfmSeqStride st; int64_t occ2_stride; int n_flows; int pad; };

template <int R, int WAVES>
__global__ __launch_bounds__(PC_BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void psfm_seq_batch_kernel(const PsfmBatchSeqOpt* __restrict__ seqs, int launch_id)
{
    const PsfmBatchSeqOpt& q = seqs[blockIdx.y];
    // (the launch may cover fewer lanes than the table has: say so if it is not enough -- psfm_batch.hip runs the batch again then)
    if (blockIdx.x == 0 && threadIdx.x == 0 && q.a.ctr->n_lanes > (int)(gridDim.x * PC_BLOCK)) atomicOr(&q.a.ctr->overflow, 16);
    psfm_seq_body<R>(q.a, q.P, q.st, q.occ2_stride, q.n_flows, launch_id);
}

// behind the last solve of every sequence of a batch (what psfm_solve_flush does for one): the accepted iterate into the log
__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_flush_batch_kernel(const PsfmBatchSeqOpt* __restrict__ seqs, int clear_sel)
{
    const PsfmBatchSeqOpt& q = seqs[blockIdx.y];
    if (q.n_flows < 2) return;
    PcParams P = q.P;
    pc_params_rebase(P, q.st, q.occ2_stride, q.n_flows - 1);
    if (*P.stall) return;
    if (clear_sel) { if (blockIdx.x == 0 && threadIdx.x == 0) *P.sel = 0; return; }
    const int m = *P.sel;
    if (m == 0) return;
    const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
    for (int i = blockIdx.x * PC_BLOCK + threadIdx.x; i < n; i += gridDim.x * PC_BLOCK) {
        if (pc_participates(P, i, n)) {
            P.x1a[i] = pc_buf1(P, m)[i];
            P.x2a[i] = pc_buf2(P, m)[i];
        }
    }
}

// per-sequence device-side program counters of a batch, set by the host at a checkpoint (a redone solve, an adapted K):
// v[i] = {pc_frame, pc_phase, pc_owner, solve_K}; pc_frame -1 = leave sequence i alone, -2 = its K only (the device may be inside a
// solve of that sequence: continuation launches do not read K)
struct PsfmBatchPc { int v[PSFM_BATCH_MAX][4]; };
__global__ void psfm_batch_set_pc_kernel(const PsfmBatchSeqOpt* __restrict__ seqs, PsfmBatchPc pc, int n_seq)
{
    const int i = threadIdx.x;
    if (i >= n_seq || pc.v[i][0] == -1) return;
    PsfmCounters* ctr = seqs[i].a.ctr;
    if (pc.v[i][0] == -2) { ctr->solve_K = pc.v[i][3]; return; }
    ctr->pc_frame = pc.v[i][0]; ctr->pc_phase = pc.v[i][1]; ctr->pc_owner = pc.v[i][2]; ctr->solve_K = pc.v[i][3];
}

// a checkpoint of the batch in ONE device-to-host copy: every sequence's counters and the statistics of the solves of frames
// [lo[i], lo[i] + win) packed into out[i * row_bytes ..]
struct PsfmBatchWin { int lo[PSFM_BATCH_MAX]; };
__global__ __launch_bounds__(64) void psfm_batch_pack_kernel(const PsfmBatchSeqOpt* __restrict__ seqs, PsfmBatchWin w, int win, int row_bytes,
                                                            char* __restrict__ out)
{
    const PsfmBatchSeqOpt& q = seqs[blockIdx.x];
    char* dst = out + (size_t)blockIdx.x * row_bytes;
    const int* src = (const int*)q.a.ctr;
    for (int k = threadIdx.x; k < (int)(sizeof(PsfmCounters) / 4); k += 64) ((int*)dst)[k] = src[k];
    const int lo = w.lo[blockIdx.x];
    const int n = min(win, q.n_flows + 1 - lo);
    if (!q.P.stats_dev || n <= 0) return;
    const int* ssrc = (const int*)(q.P.stats_dev + lo);
    int* sdst = (int*)(dst + sizeof(PsfmCounters));
    for (int k = threadIdx.x; k < n * (int)(sizeof(psfm_solve_stats) / 4); k += 64) sdst[k] = ssrc[k];
}


// Track-sharded runs: the control step on the totals over all ranks (every rank runs it on the same numbers).
// mode 0: replay over the K iterations of a fused launch; 1 / 2: one step behind pc_init / pc_iter.
// stall_host (mode 0, may be NULL): the stall flag as this control step leaves it, written straight into pinned host memory, where
// psfm_shard_peek_stall reads it without a synchronisation (round 3 sent a 4-byte copy behind every control launch: one more
// command on the stream per frame)
__global__ __launch_bounds__(128) void psfm_pc_control_kernel(PcParams P, const double* totals, int K, int mode, int* stall_host)
{
    if (blockIdx.x != 0) return;
    if (mode == 0) {
        // one round trip for everything the replay reads -- the stall flag, the control block, the K x 13 totals -- instead of three
        // dependent ones by a single lane (7.7 us per frame of a sharded sequence went into this launch)
        __shared__ double s_tot[PC_KMAX * PC_NSUM];
        __shared__ PsfmSolveCtrl s_ctrl;
        __shared__ int s_stall;
        const int tid = threadIdx.x;
        const int kk = K < 1 ? 1 : (K > PC_KMAX ? PC_KMAX : K);
        if (tid < kk * PC_NSUM) s_tot[tid] = totals[tid];
        static_assert(sizeof(PsfmSolveCtrl) % 4 == 0 && PC_KMAX * PC_NSUM <= 128, "control kernel: one load per thread");
        for (int q = tid; q < (int)(sizeof(PsfmSolveCtrl) / 4); q += 128) ((unsigned*)&s_ctrl)[q] = ((const unsigned*)P.ctrl)[q];
        if (tid == 0) s_stall = *P.stall;
        __syncthreads();
        if (tid != 0) return;
        if (!s_stall) pc_fused_replay(P, s_tot, kk, true, nullptr, 0, &s_ctrl);
        if (stall_host) __hip_atomic_store(stall_host, *P.stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x != 0) return;
    PsfmSolveCtrl C = *P.ctrl;
    if (mode == 2 && C.done) return;
    pc_chain_control(C, totals, mode == 1 ? 0 : 1);
    C.launches += 1;
    *P.ctrl = C;
}

// Copy the accepted iterate of the last fused solve from buffer *sel into the log (what the next chain step does on
// its way; needed behind the LAST solve of a sequence, whose positions no chain step picks up).
__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_flush_kernel(PcParams P)
{
    if (*P.stall) return;
    const int m = *P.sel;
    if (m == 0) return;
    const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
    const int i = blockIdx.x * PC_BLOCK + threadIdx.x;
    if (pc_participates(P, i, n)) {
        P.x1a[i] = pc_buf1(P, m)[i];
        P.x2a[i] = pc_buf2(P, m)[i];
    }
}
__global__ void psfm_pc_clear_sel_kernel(int* sel, const int* stall) { if (!*stall) *sel = 0; }
__global__ void psfm_pc_raise_stall_kernel(int* stall, int frame) { if (!*stall) *stall = frame + 1; }

// what thread 0 of block 0 leaves behind a finished solve of the chain: the frame's statistics, PsfmCounters::sel, the lane snapshot
__device__ __forceinline__ void pc_writeback_scalars(const PcParams& P, const PsfmSolveCtrl& C)
{
    if (P.stats_dev) {
        psfm_solve_stats st;
        st.iterations = C.iteration; st.successful_steps = C.successful;
        st.termination = C.n_tracks == 0 ? -1 : C.termination; st.dogleg_nonGN = C.nonGN;
        st.initial_cost = C.initial_cost; st.final_cost = C.x_cost;
        if (C.failed) st.termination = PSFM_TERM_FAILURE;
        P.stats_dev[P.frame] = st;
    }
    if (P.sel) {
        *P.sel = 0;                                   // the iterate is (being) copied into buffer 0 right here
        if (P.n_lanes_snap) P.n_lanes_snap[(P.frame + 1) & 1] = *P.n_lanes_ptr;   // (no chain step is running: the lane count is final)
    }
}

// final: if the accepted iterate lives in the scratch pair, copy it back (frame mode: into the log); C: the final control block
__device__ __forceinline__ void pc_writeback_tracks(const PcParams& P, const PsfmSolveCtrl& C, double* out_rows)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) pc_writeback_scalars(P, C);
    const int n = P.n_lanes_ptr ? min(*P.n_lanes_ptr, P.n_rows) : P.n_rows;
    // A failed solve is not an error (the reference ignores Ceres' FAILURE, trajectory_optimize.cpp:81-82) and Ceres
    // hands the parameters back as they came in (solver.cc Minimize(), Summary::IsSolutionUsable() is false): buffer 0,
    // which the chain never writes before this point, whatever was accepted on the way -- non-finite residuals at
    // iteration 0, five invalid steps in a row or a system that lost definiteness behind accepted steps alike.
    const int cur = C.failed ? 0 : C.cur;
    if (cur == 0 && !out_rows) return;
    const double2* xc1 = pc_buf1(P, cur);
    const double2* xc2 = pc_buf2(P, cur);
    for (int i = blockIdx.x * PC_BLOCK + threadIdx.x; i < n; i += gridDim.x * PC_BLOCK) {
        if (!pc_participates(P, i, n)) continue;
        const double2 p1 = xc1[i], p2 = xc2[i];
        if (out_rows) {
            out_rows[4 * (int64_t)i + 0] = p1.x; out_rows[4 * (int64_t)i + 1] = p1.y;
            out_rows[4 * (int64_t)i + 2] = p2.x; out_rows[4 * (int64_t)i + 3] = p2.y;
        } else {
            P.x1a[i] = p1; P.x2a[i] = p2;
        }
    }
}

__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_writeback_kernel(PcParams P, double* out_rows)
{
    if (*P.stall) return;
    const PsfmSolveCtrl C = *P.ctrl;
    if (!C.done) {
        // the unrolled iterations did not suffice: poison everything that was enqueued behind this solve; the
        // host resumes it (psfm_track polls `stall` at its checkpoints) and re-enqueues the later frames
        if (blockIdx.x == 0 && threadIdx.x == 0) *P.stall = P.frame + 1;
        return;
    }
    if (C.written) return;       // (the persistent solve has written back itself)
    pc_writeback_tracks(P, C, out_rows);
}

// batch mode: split (n,4) rows into the (x1, x2) pair and (n,2) refs into double2 arrays
__global__ __launch_bounds__(PC_BLOCK) void psfm_pc_load_rows_kernel(const double* __restrict__ uv12, int64_t n,
                                                                     double2* __restrict__ x1, double2* __restrict__ x2)
{
    const int64_t i = (int64_t)blockIdx.x * PC_BLOCK + threadIdx.x;
    if (i >= n) return;
    x1[i] = make_double2(uv12[4 * i], uv12[4 * i + 1]);
    x2[i] = make_double2(uv12[4 * i + 2], uv12[4 * i + 3]);
}

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
static int pc_blocks(const psfm_ctx* c, int n_rows_upper)
{
    // grid of the launch chain's grid-stride kernels; PSFM_PC_BLOCKS (<= PC_MAX_BLOCKS) overrides for measurements.  A context with a
    // resident budget (psfm_ctx_set_resident_budget: its resident solves share the device with other contexts') stays inside it
    static const int limit_env = getenv("PSFM_PC_BLOCKS") ? atoi(getenv("PSFM_PC_BLOCKS")) : 0;
    int limit = limit_env >= 1 && limit_env <= PC_MAX_BLOCKS ? limit_env : PC_CHAIN_BLOCKS;
    if (c->resident_budget > 0 && c->resident_budget < limit) limit = c->resident_budget;
    int n_blocks = (n_rows_upper + PC_BLOCK - 1) / PC_BLOCK;
    if (n_blocks > limit) n_blocks = limit;
    return n_blocks < 1 ? 1 : n_blocks;
}

// sol_ctrl layout: [PsfmSolveCtrl][ticket u32][zero word used as the `stall` flag of batch solves]
static psfm_status pc_setup(psfm_ctx* c, PcParams& P, hipStream_t s)
{
    psfm_status rc;
    if ((rc = c->sol_partials.ensure(sizeof(double) * PC_MAX_BLOCKS * PC_NSUM)) != PSFM_OK) return rc;
    const bool fresh = c->sol_ctrl.p == nullptr;
    if ((rc = c->sol_ctrl.ensure(sizeof(PsfmSolveCtrl) + 64)) != PSFM_OK) return rc;
    if (fresh) PSFM_HIP(hipMemsetAsync(c->sol_ctrl.p, 0, sizeof(PsfmSolveCtrl) + 64, s));
    P.partials = c->sol_partials.as<double>();
    P.ctrl = c->sol_ctrl.as<PsfmSolveCtrl>();
    P.ticket = (unsigned*)((char*)c->sol_ctrl.p + sizeof(PsfmSolveCtrl));
    if (!P.stall) P.stall = (int*)((char*)c->sol_ctrl.p + sizeof(PsfmSolveCtrl) + 16);
    // the blocks' lists (pc_build_list): block b of the chain's grid sees the lane chunks b, b + n_blocks, ...
    const int n_blocks = pc_blocks(c, P.n_rows);
    const int chunks = (P.n_rows + PC_BLOCK - 1) / PC_BLOCK;
    // (a banded block takes every (n_blocks / 8)-th chunk of a band of ceil(chunks / 8): at most one chunk more than in block order)
    P.list_pitch = ((chunks + n_blocks - 1) / n_blocks + 1) * PC_BLOCK;
    {
        static const int band = getenv("PSFM_PC_BAND") ? atoi(getenv("PSFM_PC_BAND")) : 1;      // (0: chunks in block order; measurements)
        P.list_banded = band;
    }
    if ((rc = c->sol_list.ensure(sizeof(int) * ((size_t)n_blocks * P.list_pitch + PC_MAX_BLOCKS))) != PSFM_OK) return rc;
    P.list_n = c->sol_list.as<int>();
    P.list = P.list_n + PC_MAX_BLOCKS;
    return PSFM_OK;
}

static void pc_fill_stats(const PsfmSolveCtrl* h, psfm_solve_stats* st)
{
    st->iterations = h->iteration;
    st->successful_steps = h->successful;
    st->termination = h->n_tracks == 0 ? -1 : h->termination;   // -1: nothing to solve (the reference would raise here)
    st->dogleg_nonGN = h->nonGN;
    st->initial_cost = h->initial_cost;
    st->final_cost = h->x_cost;
}

// Keep launching iterations, polling `done` between chunks, until the solve terminates; then write back.
// resident: a resident launch is what has been enqueued -- not done behind it means its hand-off gave up (it has written nothing: the
// control block is as iteration 0 left it, the launches below take the solve from there); a call that sees that twice stops trying
static psfm_status pc_finish_sync(psfm_ctx* c, PcParams& P, int n_blocks, double* out_rows, psfm_solve_stats* st, hipStream_t s,
                                  bool resident = false)
{
    PsfmSolveCtrl* hctrl = (PsfmSolveCtrl*)((char*)c->host_pinned + 512 + sizeof(PsfmShard) * PSFM_NSHARD);
    int launched = 0, chunk = 6;
    for (;;) {
        PSFM_HIP(hipMemcpyAsync(hctrl, P.ctrl, sizeof(PsfmSolveCtrl), hipMemcpyDeviceToHost, s));
        PSFM_HIP(hipStreamSynchronize(s));
        if (hctrl->done) break;
        if (resident && launched == 0) c->pc_giveups += 1;
        if (launched > 2 * 200 + 64) { psfm_set_error("path-consistency solver did not terminate"); return PSFM_ERR_SOLVER; }
        for (int k = 0; k < chunk; ++k) hipLaunchKernelGGL(psfm_pc_iter_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
        c->n_iter_launches += chunk;
        PSFM_HIP(hipGetLastError());
        launched += chunk;
        chunk = chunk < 32 ? chunk * 2 : 32;
    }
    hipLaunchKernelGGL(psfm_pc_writeback_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, out_rows);
    PSFM_HIP(hipGetLastError());
    if (st) pc_fill_stats(hctrl, st);
    // (a failed solve -- invalid steps / a system that is not positive definite / non-finite residuals -- is not an error: like the reference,
    // which ignores Ceres' FAILURE at trajectory_optimize.cpp:81-82, the parameters stay as they came in; stats say 5)
    return PSFM_OK;
}

static psfm_status pc_workspace(psfm_ctx* c, int64_t rows, int n_buf)
{
    psfm_status rc;
    // iterate buffers 1..n_buf (two double2 arrays each); ref1, ref2, jscale (double2 each) + scale (double)
    if ((rc = c->sol_x.ensure(sizeof(double2) * rows * 2 * n_buf)) != PSFM_OK) return rc;
    if ((rc = c->sol_state.ensure(sizeof(double2) * rows * 3 + sizeof(double) * rows)) != PSFM_OK) return rc;
    return PSFM_OK;
}

static psfm_status pc_frame_params(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                   const float* flow02, const uint8_t* occ02, int frame, PcParams& P, hipStream_t s)
{
    psfm_status rc;
    if ((rc = pc_workspace(c, d.cap, PC_KMAX)) != PSFM_OK) return rc;
    if ((rc = c->sol_stats.ensure(sizeof(psfm_solve_stats) * (size_t)(d.n_flows + 1))) != PSFM_OK) return rc;
    memset(&P, 0, sizeof(P));
    P.H = d.H; P.W = d.W; P.cw = d.cw; P.ch = d.ch;
    P.birth_frame = c->birth_frame.as<int>();
    P.max_birth = frame - 1;    // three buffered positions: times frame-1, frame, frame+1
    PsfmCounters* ctr = c->counters.as<PsfmCounters>();
    P.n_lanes_ptr = &ctr->n_lanes;
    P.stall = &ctr->stall;
    P.n_rows = (int)d.cap;
    double2* lg = c->log.as<double2>();
    P.p0 = lg + (int64_t)(frame - 1) * d.cap;
    P.x1a = lg + (int64_t)frame * d.cap;
    P.x2a = lg + (int64_t)(frame + 1) * d.cap;
    P.xs = c->sol_x.as<double2>(); P.xs_stride = d.cap;
    P.sel = &ctr->sel;
    P.n_lanes_snap = ctr->n_lanes_snap;
    P.ref1 = c->sol_state.as<double2>();
    P.ref2 = P.ref1 + d.cap;
    P.jscale = P.ref2 + d.cap;
    P.scale = (double*)(P.jscale + d.cap);
    P.flow12 = (const float2*)flow12; P.flow01 = (const float2*)flow01; P.flow02 = (const float2*)flow02;
    P.occ02 = occ02;
    P.frame = frame;
    P.stats_dev = c->sol_stats.as<psfm_solve_stats>();
    return pc_setup(c, P, s);
}

// The trust-region loop of the launch chain as ONE persistent launch behind pc_init (psfm_pc_resident_kernel), when this call
// has the device to itself and the grid is co-resident; returns false when it is not used (the caller launches iterations).
// PSFM_PC_PERSIST=0 keeps one launch per iteration (measurements, tests).
template <int NS>
static int pc_resident_capacity(psfm_ctx* c)
{
    if (c->pc_persist_blocks[NS] < 0) {
        c->pc_persist_blocks[NS] = 0;
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, psfm_pc_resident_kernel<NS, false>, PC_BLOCK, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, c->device) == hipSuccess)
            c->pc_persist_blocks[NS] = per_cu * prop.multiProcessorCount;
    }
    return c->pc_persist_blocks[NS];
}

// peers != nullptr: the launch of ONE rank of a solve that spans several (PcPeers); `peer_epoch` is the epoch all ranks tag this
// solve's granules with, the member rows live in front of the leader area of the context's peer buffer (never in sol_bar, whose
// granules carry the context's own launch counter)
static bool pc_persist_enqueue(psfm_ctx* c, const PcParams& P, int n_blocks, double* out_rows, bool init_inside, bool raise_stall, hipStream_t s,
                               const PcPeers* peers = nullptr, unsigned peer_epoch = 0)
{
    const char* env = getenv("PSFM_PC_PERSIST");          // (read per call: the tests switch it inside one process)
    if ((env && atoi(env) == 0) || !c->pc_persist_ok || c->pc_giveups >= 2 || P.export_sums) return false;
    if (n_blocks > PC_RES_BLOCKS) return false;
    // slots per thread: what the longest possible list needs, at most PC_RES_NS_MAX (longer lists are streamed behind the slots)
    int ns = P.list_pitch / PC_BLOCK - 1;           // (the pitch has one chunk of head-room, see pc_setup)
    if (getenv("PSFM_PC_SLOTS")) ns = atoi(getenv("PSFM_PC_SLOTS"));      // (measurements, tests: fewer slots = a streamed tail)
    ns = ns < 1 ? 1 : (ns > PC_RES_NS_MAX ? PC_RES_NS_MAX : ns);
    const int capacity = ns == 1 ? pc_resident_capacity<1>(c) : (ns == 2 ? pc_resident_capacity<2>(c) : pc_resident_capacity<3>(c));
    if (n_blocks > capacity) return false;                 // (every block must be resident at once)
    // granules: one row per block + one per leader; zeroed once -- tags carry the launch epoch, so what an earlier launch left
    // never matches (the 20-bit epoch wraps after a million solves: cleared again then)
    const size_t gbytes = sizeof(unsigned long long) * PC_RES_ROW * (PC_RES_BLOCKS + 2 * PC_LEADERS);      // (two sets of leader rows)
    const bool fresh = c->sol_bar.bytes < gbytes;
    if (c->sol_bar.ensure(gbytes) != PSFM_OK) return false;
    c->pc_epoch += 1;
    if (fresh || c->pc_epoch >= (1u << 20)) {
        if (hipMemsetAsync(c->sol_bar.p, 0, gbytes, s) != hipSuccess) return false;
        c->pc_epoch = 1;
    }
    // polls of (s_sleep 1 + a batch of uncached loads, ~1.5 us): ~10 ms
    const int spin_limit = getenv("PSFM_PC_SPIN") ? atoi(getenv("PSFM_PC_SPIN")) : 8000;      // (read per call: a test forces the give-up with 0)
    // tests: "block,round" makes that block give up in that round (what a grid that is not co-resident looks like to the others)
    int quit_code = 0;
    if (const char* q = getenv("PSFM_PC_QUIT")) {
        int qb = 0, qr = 0;
        if (sscanf(q, "%d,%d", &qb, &qr) == 2 && qb >= 0 && qr >= 0) quit_code = ((qb + 1) << 16) | (qr & 0xffff);
    }
    unsigned long long* gran = c->sol_bar.as<unsigned long long>();
    const int max_rounds = 2 * 200 + 64;
    PcPeers pp;
    memset(&pp, 0, sizeof(pp));
    // (the write-back is in the launch too)
    if (peers) {
        pp = *peers;
        // ranks start this launch up to a host-side scheduling delay apart: the polls wait longer before they give a solve up (~100 ms)
        const int spin_peer = getenv("PSFM_PC_SPIN") ? spin_limit : 80000;
        unsigned long long* members = c->peer_area.as<unsigned long long>();      // [PC_RES_BLOCKS rows], in front of the leader area
        const unsigned ep = peer_epoch & 0xfffffu;
        if (ns == 1) hipLaunchKernelGGL((psfm_pc_resident_kernel<1, true>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, members, ep, spin_peer, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
        else if (ns == 2) hipLaunchKernelGGL((psfm_pc_resident_kernel<2, true>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, members, ep, spin_peer, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
        else hipLaunchKernelGGL((psfm_pc_resident_kernel<3, true>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, members, ep, spin_peer, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
    }
    else if (ns == 1) hipLaunchKernelGGL((psfm_pc_resident_kernel<1, false>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, gran, c->pc_epoch, spin_limit, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
    else if (ns == 2) hipLaunchKernelGGL((psfm_pc_resident_kernel<2, false>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, gran, c->pc_epoch, spin_limit, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
    else hipLaunchKernelGGL((psfm_pc_resident_kernel<3, false>), dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, gran, c->pc_epoch, spin_limit, max_rounds, quit_code, init_inside ? 1 : 0, raise_stall ? 1 : 0, out_rows, pp);
    c->n_resident += 1;
    return true;
}

// ---- ONE solve over several ranks (psfm_shard.hip) ----
size_t psfm_peer_area_bytes(void) { return sizeof(unsigned long long) * PC_RES_ROW * ((size_t)PC_RES_BLOCKS + 2 * PSFM_MAX_PEERS * PC_LEADERS); }
size_t psfm_peer_lead_offset(void) { return sizeof(unsigned long long) * PC_RES_ROW * (size_t)PC_RES_BLOCKS; }
int psfm_peer_leaders(int n_blocks) { return n_blocks < PC_LEADERS ? n_blocks : PC_LEADERS; }
int psfm_solve_blocks(psfm_ctx* c, const PsfmTrackDims& d) { return pc_blocks(c, (int)d.cap); }

// This rank's launch of the solve of `frame` over all ranks of c->peers: iteration 0, the trust-region loop with the cross-rank
// all-reduce, the write-back -- enqueued, nothing read back.  There is no launch-chain form of it (launches cannot add sums across
// ranks): a launch that cannot be made, or that gives up, leaves the stall flag raised, every rank's polls run into this rank's
// poison (or their limit), and the caller redoes the solve in the exchange form at its checkpoint.
psfm_status psfm_solve_frame_enqueue_peer(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                          const float* flow02, const uint8_t* occ02, int frame, unsigned epoch, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = pc_blocks(c, (int)d.cap);
    PcPeers pp;
    memset(&pp, 0, sizeof(pp));
    pp.world = c->peer_world; pp.rank = c->peer_rank;
    for (int r = 0; r < c->peer_world; ++r) { pp.L[r] = c->peer_L[r]; pp.lead[r] = (unsigned long long*)c->peer_lead[r]; }
    if (pp.L[pp.rank] != psfm_peer_leaders(n_blocks)) {
        psfm_set_error("psfm_shard_solve_peer: this rank's launch has %d blocks, the ranks were told %d leaders", n_blocks, pp.L[pp.rank]);
        return PSFM_ERR_ARG;
    }
    const bool saved_ok = c->pc_persist_ok;
    const int saved_giveups = c->pc_giveups;
    c->pc_persist_ok = true; c->pc_giveups = 0;          // (the caller decided for all ranks at once: psfm_shard_peer_connect)
    const bool ok = pc_persist_enqueue(c, P, n_blocks, nullptr, true, true, s, &pp, epoch);
    c->pc_persist_ok = saved_ok; c->pc_giveups = saved_giveups;
    if (!ok) {
        // (the other ranks must not wait for sums that will never come: raise the flag, they give up at their spin limit)
        hipLaunchKernelGGL(psfm_pc_raise_stall_kernel, dim3(1), dim3(1), 0, s, P.stall, frame);
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// Enqueue one frame's solve with NO host synchronisation: the resident solve (iteration 0, the trust-region loop and the
// write-back in ONE launch) when the call has the device to itself, else pc_init + `unroll` launches of one iteration each; a
// solve that is not done behind them -- the loop gave up on its hand-off, the launches did not suffice -- has its write-back
// kernel raise the device-side stall flag, which turns everything enqueued behind it into no-ops; the host redoes it at its
// checkpoint.
psfm_status psfm_solve_frame_enqueue(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                     const float* flow02, const uint8_t* occ02, int frame, int unroll, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = pc_blocks(c, (int)d.cap);
    // iteration 0, then the loop and the write-back in one launch when possible (a loop that gave up on its barrier leaves the
    // control block "not done": the write-back kernel behind it then raises the stall flag and the host redoes the solve at its
    // checkpoint), else `unroll` launches of one iteration each + write-back
    // (PSFM_PC_INIT_INSIDE=0: iteration 0 as its own launch in front of the resident solve -- measurements, tests)
    const bool inside = !(getenv("PSFM_PC_INIT_INSIDE") && atoi(getenv("PSFM_PC_INIT_INSIDE")) == 0);
    if (!inside || !pc_persist_enqueue(c, P, n_blocks, nullptr, true, true, s)) {
        hipLaunchKernelGGL(psfm_pc_init_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
        if (!pc_persist_enqueue(c, P, n_blocks, nullptr, false, true, s)) {
            for (int k = 0; k < unroll; ++k) hipLaunchKernelGGL(psfm_pc_iter_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
            c->n_iter_launches += unroll;
            // (the resident solve writes back, or raises the stall flag, itself)
            hipLaunchKernelGGL(psfm_pc_writeback_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P, (double*)nullptr);
        }
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// The stalled solve of `frame`: clear the flag, iterate to termination with host polling, write back.
psfm_status psfm_solve_frame_resume(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                    const float* flow02, const uint8_t* occ02, int frame, psfm_solve_stats* st,
                                    int try_fused_k, bool chain_stalled, hipStream_t s)
{
    // chain_stalled: the solve that raised the flag was a launch-chain solve already -- it ran out of unrolled launches or its
    // resident launch gave up (the grid was not co-resident: another process on the device); this redo uses launches, and a
    // call that sees it happen twice stops trying the resident form
    if (chain_stalled && c->pc_persist_ok) c->pc_giveups += 1;
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    PSFM_HIP(hipMemsetAsync(P.stall, 0, sizeof(int), s));
    // from the top: the launch chain never writes buffer 0 (the log slabs) before its write-back; a fused solve that gave
    // up left the values it started from in the iterate buffer PsfmCounters::sel names -- back into the log first
    const int nb_all = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    hipLaunchKernelGGL(psfm_pc_flush_kernel, dim3(nb_all), dim3(PC_BLOCK), 0, s, P);
    hipLaunchKernelGGL(psfm_pc_clear_sel_kernel, dim3(1), dim3(1), 0, s, P.sel, (const int*)P.stall);
    if (try_fused_k > 0) {
        // most redos are solves that needed one or two iterations more than the sequence's usual: the fused solve with
        // try_fused_k iterations settles those in one launch; anything else falls through to the chain
        psfm_status rc2 = psfm_solve_frame_fused(c, d, flow01, flow12, flow02, occ02, frame, try_fused_k, s);
        if (rc2 != PSFM_OK) return rc2;
        PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
        psfm_solve_stats* hs = (psfm_solve_stats*)((char*)c->host_pinned + 288);
        PSFM_HIP(hipMemcpyAsync(hc, c->counters.p, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
        PSFM_HIP(hipMemcpyAsync(hs, c->sol_stats.as<psfm_solve_stats>() + frame, sizeof(psfm_solve_stats), hipMemcpyDeviceToHost, s));
        PSFM_HIP(hipStreamSynchronize(s));
        if (!hc->stall) { if (st) *st = *hs; return PSFM_OK; }
        PSFM_HIP(hipMemsetAsync(P.stall, 0, sizeof(int), s));
        hipLaunchKernelGGL(psfm_pc_flush_kernel, dim3(nb_all), dim3(PC_BLOCK), 0, s, P);
        hipLaunchKernelGGL(psfm_pc_clear_sel_kernel, dim3(1), dim3(1), 0, s, P.sel, (const int*)P.stall);
    }
    const int n_blocks = pc_blocks(c, (int)d.cap);
    hipLaunchKernelGGL(psfm_pc_init_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
    // (the host polls behind this launch: a resident solve that gives up must leave the stall flag alone -- see pc_res_stall)
    bool resident = false;
    if (chain_stalled || !(resident = pc_persist_enqueue(c, P, n_blocks, nullptr, false, false, s))) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(psfm_pc_iter_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
        c->n_iter_launches += 4;
    }
    PSFM_HIP(hipGetLastError());
    return pc_finish_sync(c, P, n_blocks, nullptr, st, s, resident);
}

int psfm_solve_kmax(void) { return PC_KMAX; }
int psfm_resident_blocks(psfm_ctx* c) { return pc_resident_capacity<PC_RES_NS_MAX>(c); }

// the chain steps of track_optimize capture the iterate-buffer pointer: allocate before the first one is enqueued
psfm_status psfm_solve_prepare(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s)
{
    psfm_status rc;
    if ((rc = pc_workspace(c, d.cap, PC_KMAX)) != PSFM_OK) return rc;
    if ((rc = c->sol_stats.ensure(sizeof(psfm_solve_stats) * (size_t)(d.n_flows + 1))) != PSFM_OK) return rc;
    // The fused solve's tickets, zero at the start of every sequence.  Every launch leaves them at zero -- unless a sequence runs into
    // its lane capacity: its launches then count on more blocks than the grid has, never see their last arrival, and the tickets would
    // stay mid-count for every later sequence of this context (round 5's stress: a capacity retry that made "no progress").
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    const int n_groups = (n_blocks + PC_GROUP - 1) / PC_GROUP;
    const size_t tbytes = 4096 * sizeof(unsigned), gbytes = sizeof(double) * (size_t)n_groups * PC_KMAX * PC_NSUM;
    if (n_groups + 1 > 4096) { psfm_set_error("psfm_track: lane table too large for the fused solve"); return PSFM_ERR_ARG; }
    if ((rc = c->sol_fused.ensure(tbytes + gbytes)) != PSFM_OK) return rc;
    PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, tbytes, s));
    return PSFM_OK;
}

// One frame's solve as ONE launch that speculates K Gauss-Newton iterations (psfm_pc_fused_kernel).  No host
// synchronisation: a solve that does not go as speculated raises the stall flag like a launch chain that ran out of
// iterations.  The accepted iterate stays in an iterate buffer; the next chain step (or psfm_solve_flush) moves it.
psfm_status psfm_solve_frame_fused(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                   const float* flow02, const uint8_t* occ02, int frame, int K, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    const int n_groups = (n_blocks + PC_GROUP - 1) / PC_GROUP;
    if ((rc = c->sol_partials.ensure(sizeof(double) * (size_t)n_blocks * PC_KMAX * PC_NSUM)) != PSFM_OK) return rc;
    // tickets first (their place must not depend on the shape: they are zeroed once and every launch leaves them at zero)
    const size_t tbytes = 4096 * sizeof(unsigned), gbytes = sizeof(double) * (size_t)n_groups * PC_KMAX * PC_NSUM;
    if (n_groups + 1 > 4096) { psfm_set_error("psfm_track: lane table too large for the fused solve"); return PSFM_ERR_ARG; }
    void* before = c->sol_fused.p;
    if ((rc = c->sol_fused.ensure(tbytes + gbytes)) != PSFM_OK) return rc;
    if (c->sol_fused.p != before) PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, tbytes, s));
    P.partials = c->sol_partials.as<double>();
    P.gticket = c->sol_fused.as<unsigned>();
    P.gpart = (double*)((char*)c->sol_fused.p + tbytes);
    P.K = K < 1 ? 1 : (K > PC_KMAX ? PC_KMAX : K);
    static const int waves = getenv("PSFM_FUSED_WAVES") ? atoi(getenv("PSFM_FUSED_WAVES")) : 4;   // (round 3: 124 VGPRs, nothing spilled, once the kernel states the range of K)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    c->prof.kernel_span(PSFM_PROF_SOLVER, &e0, &e1, true);   // (profiling on: exact begin / end of every fused launch)
    if (waves == 3) hipExtLaunchKernelGGL(psfm_pc_fused_kernel<3>, dim3(n_blocks), dim3(PC_BLOCK), 0, s, e0, e1, 0, P);
    else hipExtLaunchKernelGGL(psfm_pc_fused_kernel<4>, dim3(n_blocks), dim3(PC_BLOCK), 0, s, e0, e1, 0, P);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ---- track-sharded runs (psfm_shard.hip): one launch of the solve of `frame` that only exports its sums
//      (kind 0 fused with K iterations, 1 pc_init, 2 pc_iter), the control step on the combined totals, the write-back ----
psfm_status psfm_solve_export(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12, const float* flow02,
                              const uint8_t* occ02, int frame, int kind, int K, double* sums_out, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    P.export_sums = sums_out;
    if (kind == 0) {
        const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
        const int n_groups = (n_blocks + PC_GROUP - 1) / PC_GROUP;
        if ((rc = c->sol_partials.ensure(sizeof(double) * (size_t)n_blocks * PC_KMAX * PC_NSUM)) != PSFM_OK) return rc;
        const size_t tbytes = 4096 * sizeof(unsigned), gbytes = sizeof(double) * (size_t)n_groups * PC_KMAX * PC_NSUM;
        if (n_groups + 1 > 4096) { psfm_set_error("lane table too large for the fused solve"); return PSFM_ERR_ARG; }
        void* before = c->sol_fused.p;
        if ((rc = c->sol_fused.ensure(tbytes + gbytes)) != PSFM_OK) return rc;
        if (c->sol_fused.p != before) PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, tbytes, s));
        P.partials = c->sol_partials.as<double>();
        P.gticket = c->sol_fused.as<unsigned>();
        P.gpart = (double*)((char*)c->sol_fused.p + tbytes);
        P.K = K < 1 ? 1 : (K > PC_KMAX ? PC_KMAX : K);
        hipLaunchKernelGGL(psfm_pc_fused_kernel<4>, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
    } else {
        const int n_blocks = pc_blocks(c, (int)d.cap);
        if (kind == 1) hipLaunchKernelGGL(psfm_pc_init_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
        else hipLaunchKernelGGL(psfm_pc_iter_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_solve_control(psfm_ctx* c, const PsfmTrackDims& d, int frame, int kind, int K, const double* totals, hipStream_t s,
                               int* stall_host)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, nullptr, nullptr, nullptr, nullptr, frame, P, s);
    if (rc != PSFM_OK) return rc;
    hipLaunchKernelGGL(psfm_pc_control_kernel, dim3(1), dim3(128), 0, s, P, totals, K, kind, kind == 0 ? stall_host : nullptr);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// solver state after a control step: done flag, stall flag, statistics (synchronises)
psfm_status psfm_solve_state(psfm_ctx* c, int* done, int* stall, psfm_solve_stats* st, hipStream_t s)
{
    PsfmSolveCtrl* hctrl = (PsfmSolveCtrl*)((char*)c->host_pinned + 512 + sizeof(PsfmShard) * PSFM_NSHARD);
    PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
    PSFM_HIP(hipMemcpyAsync(hctrl, c->sol_ctrl.p, sizeof(PsfmSolveCtrl), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipMemcpyAsync(hc, c->counters.p, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    if (done) *done = hctrl->done;
    if (stall) *stall = hc->stall;
    if (st) pc_fill_stats(hctrl, st);
    if (st && hctrl->failed) st->termination = PSFM_TERM_FAILURE;
    return PSFM_OK;
}

// redo of a fused solve that gave up: the values it started from back into the log, stall flag cleared
psfm_status psfm_solve_restore(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, nullptr, nullptr, nullptr, nullptr, frame, P, s);
    if (rc != PSFM_OK) return rc;
    PSFM_HIP(hipMemsetAsync(P.stall, 0, sizeof(int), s));
    const int nb = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    hipLaunchKernelGGL(psfm_pc_flush_kernel, dim3(nb), dim3(PC_BLOCK), 0, s, P);
    hipLaunchKernelGGL(psfm_pc_clear_sel_kernel, dim3(1), dim3(1), 0, s, P.sel, (const int*)P.stall);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_solve_writeback(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, nullptr, nullptr, nullptr, nullptr, frame, P, s);
    if (rc != PSFM_OK) return rc;
    hipLaunchKernelGGL(psfm_pc_writeback_kernel, dim3(pc_blocks(c, (int)d.cap)), dim3(PC_BLOCK), 0, s, P, (double*)nullptr);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// One loop iteration of track_optimize as ONE launch: chain step of `frame` + the fused solve of its tracks
// (psfm_frame_kernel).  flow12 = the frame's own forward flow, occ = its occlusion map.
psfm_status psfm_launch_frame(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12, const float* flow02,
                              const uint8_t* occ, const uint8_t* occ02, int frame, int K, hipStream_t s, double* export_sums)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, flow01, flow12, flow02, occ02, frame, P, s);
    if (rc != PSFM_OK) return rc;
    P.export_sums = export_sums;      // track-sharded runs: the K x 13 sums of this process instead of the control step
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    const int n_groups = (n_blocks + PC_GROUP - 1) / PC_GROUP;
    if ((rc = c->sol_partials.ensure(sizeof(double) * (size_t)n_blocks * PC_KMAX * PC_NSUM)) != PSFM_OK) return rc;
    const size_t tbytes = 4096 * sizeof(unsigned), gbytes = sizeof(double) * (size_t)n_groups * PC_KMAX * PC_NSUM;
    if (n_groups + 1 > 4096) { psfm_set_error("psfm_track: lane table too large for the fused solve"); return PSFM_ERR_ARG; }
    void* before = c->sol_fused.p;
    if ((rc = c->sol_fused.ensure(tbytes + gbytes)) != PSFM_OK) return rc;
    if (c->sol_fused.p != before) PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, tbytes, s));
    P.partials = c->sol_partials.as<double>();
    P.gticket = c->sol_fused.as<unsigned>();
    P.gpart = (double*)((char*)c->sol_fused.p + tbytes);
    P.K = K < 1 ? 1 : (K > PC_KMAX ? PC_KMAX : K);
    PsfmChainArgs a;
    psfm_fill_chain_args(c, d, flow12, occ, frame, a, s);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    c->prof.kernel_span(PSFM_PROF_SOLVER, &e0, &e1, true);
    const dim3 grid((unsigned)n_blocks), block(PC_BLOCK);
    switch (d.ratio) {
        case 1: hipExtLaunchKernelGGL(psfm_frame_kernel<1>, grid, block, 0, s, e0, e1, 0, a, P); break;
        case 2: hipExtLaunchKernelGGL(psfm_frame_kernel<2>, grid, block, 0, s, e0, e1, 0, a, P); break;
        case 4: hipExtLaunchKernelGGL(psfm_frame_kernel<4>, grid, block, 0, s, e0, e1, 0, a, P); break;
        default: hipExtLaunchKernelGGL(psfm_frame_kernel<0>, grid, block, 0, s, e0, e1, 0, a, P); break;
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// The arguments every launch of a device-paced sequence carries: those of frame 1 + the strides of the per-frame stacks
// (psfm_seq_kernel gets them as kernel arguments, psfm_seq_batch_kernel from its sequence's row of the batch table).
static psfm_status pc_seq_args(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch,
                               const float* flows_f2, const uint8_t* occ_s2, PcParams& P, PsfmChainArgs& a, PsfmSeqStride& st, hipStream_t s)
{
    const int64_t Pix = (int64_t)d.H * d.W;
    psfm_status rc = pc_frame_params(c, d, flows, flows + Pix * 2, flows_f2, occ_s2, 1, P, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    const int n_groups = (n_blocks + PC_GROUP - 1) / PC_GROUP;
    if ((rc = c->sol_partials.ensure(sizeof(double) * (size_t)n_blocks * PC_KMAX * PC_NSUM)) != PSFM_OK) return rc;
    const size_t tbytes = 4096 * sizeof(unsigned), gbytes = sizeof(double) * (size_t)n_groups * PC_KMAX * PC_NSUM;
    if (n_groups + 1 > 4096) { psfm_set_error("psfm_track: lane table too large for the fused solve"); return PSFM_ERR_ARG; }
    void* before = c->sol_fused.p;
    if ((rc = c->sol_fused.ensure(tbytes + gbytes)) != PSFM_OK) return rc;
    if (c->sol_fused.p != before) PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, tbytes, s));
    P.partials = c->sol_partials.as<double>();
    P.gticket = c->sol_fused.as<unsigned>();
    P.gpart = (double*)((char*)c->sol_fused.p + tbytes);
    psfm_fill_chain_args_nolaunch(c, d, flows + Pix * 2, occ + occ_pitch, 1, a);
    a.owner_clear = 1;
    st.flow = Pix; st.occ = occ_pitch; st.cap = d.cap;          // (float2 / byte / double2 elements per frame)
    return PSFM_OK;
}

// ---- batch (psfm_batch.hip): rows of the table, the batched launches ----
size_t psfm_batch_seq_opt_bytes(void) { return sizeof(PsfmBatchSeqOpt); }
psfm_status psfm_batch_fill_seq_opt(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch,
                                    const float* flows_f2, const uint8_t* occ_s2, void* row_host, hipStream_t s)
{
    PsfmBatchSeqOpt* q = (PsfmBatchSeqOpt*)row_host;
    memset(q, 0, sizeof(*q));
    psfm_status rc = pc_seq_args(c, d, flows, occ, occ_pitch, flows_f2, occ_s2, q->P, q->a, q->st, s);
    if (rc != PSFM_OK) return rc;
    q->occ2_stride = (int64_t)d.H * d.W;
    q->n_flows = d.n_flows;
    return PSFM_OK;
}

psfm_status psfm_launch_seq_batch(psfm_ctx* owner, const void* tab_dev, int n_seq, int ratio, int64_t cap_max, int n_launches, int launch_id0,
                                  hipStream_t s)
{
    const unsigned gx = (unsigned)(((cap_max + PC_BLOCK - 1) / PC_BLOCK + 7) / 8 * 8);
    const dim3 grid(gx, (unsigned)n_seq), block(PC_BLOCK);
    const PsfmBatchSeqOpt* tab = (const PsfmBatchSeqOpt*)tab_dev;
    for (int k = 0; k < n_launches; ++k) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        owner->prof.kernel_span(PSFM_PROF_SOLVER, &e0, &e1, true);
        const int id = launch_id0 + k;
        switch (ratio) {
            case 1: hipExtLaunchKernelGGL((psfm_seq_batch_kernel<1, PSFM_SEQ_WAVES_DEFAULT>), grid, block, 0, s, e0, e1, 0, tab, id); break;
            case 2: hipExtLaunchKernelGGL((psfm_seq_batch_kernel<2, PSFM_SEQ_WAVES_DEFAULT>), grid, block, 0, s, e0, e1, 0, tab, id); break;
            case 4: hipExtLaunchKernelGGL((psfm_seq_batch_kernel<4, PSFM_SEQ_WAVES_DEFAULT>), grid, block, 0, s, e0, e1, 0, tab, id); break;
            default: hipExtLaunchKernelGGL((psfm_seq_batch_kernel<0, PSFM_SEQ_WAVES_DEFAULT>), grid, block, 0, s, e0, e1, 0, tab, id); break;
        }
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_launch_flush_batch(const void* tab_dev, int n_seq, int64_t cap_max, hipStream_t s)
{
    int64_t nb = (cap_max + PC_BLOCK - 1) / PC_BLOCK;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(psfm_pc_flush_batch_kernel, dim3((unsigned)nb, (unsigned)n_seq), dim3(PC_BLOCK), 0, s, (const PsfmBatchSeqOpt*)tab_dev, 0);
    hipLaunchKernelGGL(psfm_pc_flush_batch_kernel, dim3(1, (unsigned)n_seq), dim3(PC_BLOCK), 0, s, (const PsfmBatchSeqOpt*)tab_dev, 1);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_launch_batch_set_pc(const void* tab_dev, const int (*v)[4], int n_seq, hipStream_t s)
{
    PsfmBatchPc pc;
    memset(&pc, 0, sizeof(pc));
    for (int i = 0; i < PSFM_BATCH_MAX; ++i) pc.v[i][0] = -1;
    for (int i = 0; i < n_seq; ++i) for (int k = 0; k < 4; ++k) pc.v[i][k] = v[i][k];
    hipLaunchKernelGGL(psfm_batch_set_pc_kernel, dim3(1), dim3(PSFM_BATCH_MAX), 0, s, (const PsfmBatchSeqOpt*)tab_dev, pc, n_seq);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

size_t psfm_batch_pack_row_bytes(int win) { return sizeof(PsfmCounters) + sizeof(psfm_solve_stats) * (size_t)win; }
psfm_status psfm_launch_batch_pack(const void* tab_dev, const int* lo, int n_seq, int win, char* out_dev, hipStream_t s)
{
    PsfmBatchWin w;
    memset(&w, 0, sizeof(w));
    for (int i = 0; i < n_seq; ++i) w.lo[i] = lo[i];
    hipLaunchKernelGGL(psfm_batch_pack_kernel, dim3((unsigned)n_seq), dim3(64), 0, s, (const PsfmBatchSeqOpt*)tab_dev, w, win,
                       (int)psfm_batch_pack_row_bytes(win), out_dev);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// Device-paced sequence: `n_launches` launches of psfm_seq_kernel (ids launch_id0 ..), every one with the arguments of
// frame 1 and the strides; what each of them does is decided on the device (PsfmCounters::pc_*).
psfm_status psfm_launch_seq(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch,
                            const float* flows_f2, const uint8_t* occ_s2, int n_launches, int launch_id0, hipStream_t s)
{
    const int64_t Pix = (int64_t)d.H * d.W;
    PcParams P;
    PsfmChainArgs a;
    PsfmSeqStride st;
    psfm_status rc = pc_seq_args(c, d, flows, occ, occ_pitch, flows_f2, occ_s2, P, a, st, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    if (getenv("PSFM_SEQ_CHECK")) {     // the device-side rebase against the host's own per-frame arguments
        const int fs[] = {2, 3, 17, 254, 255, 256, 509, 510};
        for (int f : fs) {
            if (f >= d.n_flows) continue;
            PsfmChainArgs r = a, h;
            psfm_chain_args_rebase(r, st, f);
            // (no stream: the host version may launch the stamp-wrap clear, which the device-paced form does in-kernel)
            psfm_fill_chain_args_nolaunch(c, d, flows + (size_t)f * Pix * 2, occ + (size_t)f * occ_pitch, f, h);
            PcParams rp = P, hp;
            pc_params_rebase(rp, st, Pix, f);
            if ((rc = pc_frame_params(c, d, flows + (size_t)(f - 1) * Pix * 2, flows + (size_t)f * Pix * 2, flows_f2 + (size_t)(f - 1) * Pix * 2,
                                      occ_s2 + (size_t)(f - 1) * Pix, f, hp, s)) != PSFM_OK) return rc;
            const bool ok = r.flow == h.flow && r.occ == h.occ && r.log_cur == h.log_cur && r.log_next == h.log_next && r.log_prev == h.log_prev &&
                            r.blocked_cur == h.blocked_cur && r.blocked_prev == h.blocked_prev && r.stamp_cur == h.stamp_cur &&
                            r.stamp_prev == h.stamp_prev && r.surv_cur == h.surv_cur && r.surv_prev == h.surv_prev && r.sh_pop == h.sh_pop &&
                            r.sh_push == h.sh_push && r.free_pop == h.free_pop && r.free_push == h.free_push && r.frame == h.frame &&
                            rp.p0 == hp.p0 && rp.x1a == hp.x1a && rp.x2a == hp.x2a && rp.flow01 == hp.flow01 && rp.flow12 == hp.flow12 &&
                            rp.flow02 == hp.flow02 && rp.occ02 == hp.occ02 && rp.frame == hp.frame && rp.max_birth == hp.max_birth;
            if (!ok) { psfm_set_error("psfm_seq: rebased arguments of frame %d differ from the host's", f); return PSFM_ERR_ARG; }
        }
    }
    const dim3 grid((unsigned)n_blocks), block(PC_BLOCK);
    // waves per SIMD of the frame kernel: 3 (<= 168 VGPRs; leaves the register file room for the background flow_check's
    // block beside it) or 4 (<= 128 VGPRs: the solver core fits since round 3); PSFM_SEQ_WAVES overrides for measurements
    static const int seq_waves = getenv("PSFM_SEQ_WAVES") ? atoi(getenv("PSFM_SEQ_WAVES")) : PSFM_SEQ_WAVES_DEFAULT;
    for (int k = 0; k < n_launches; ++k) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        c->prof.kernel_span(PSFM_PROF_SOLVER, &e0, &e1, true);
        const int id = launch_id0 + k;
#define PSFM_SEQ_LAUNCH(R_, W_) hipExtLaunchKernelGGL((psfm_seq_kernel<R_, W_>), grid, block, 0, s, e0, e1, 0, a, P, st, Pix, d.n_flows, id)
        if (seq_waves == 4) {
            switch (d.ratio) {
                case 1: PSFM_SEQ_LAUNCH(1, 4); break;
                case 2: PSFM_SEQ_LAUNCH(2, 4); break;
                case 4: PSFM_SEQ_LAUNCH(4, 4); break;
                default: PSFM_SEQ_LAUNCH(0, 4); break;
            }
        } else {
            switch (d.ratio) {
                case 1: PSFM_SEQ_LAUNCH(1, 3); break;
                case 2: PSFM_SEQ_LAUNCH(2, 3); break;
                case 4: PSFM_SEQ_LAUNCH(4, 3); break;
                default: PSFM_SEQ_LAUNCH(0, 3); break;
            }
        }
#undef PSFM_SEQ_LAUNCH
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// Behind the last solve of a sequence: the accepted iterate into the log (no chain step follows to do it).
psfm_status psfm_solve_flush(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s)
{
    PcParams P;
    psfm_status rc = pc_frame_params(c, d, nullptr, nullptr, nullptr, nullptr, frame, P, s);
    if (rc != PSFM_OK) return rc;
    const int n_blocks = (int)((d.cap + PC_BLOCK - 1) / PC_BLOCK);
    hipLaunchKernelGGL(psfm_pc_flush_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
    hipLaunchKernelGGL(psfm_pc_clear_sel_kernel, dim3(1), dim3(1), 0, s, P.sel, (const int*)P.stall);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// path_consistency_cost.h:42-59 as ceres::AutoDiffCostFunction<PathConsistencyError, 6, 4>::Evaluate sees it, one residual block per
// thread: the six residuals and the 6 x 4 Jacobian (row-major, like Ceres' jacobians[0]) at uv12 -- through the SAME pc_core_eval the
// solver kernels call, written out in full so that a test can hold it against the functor differentiated by an independent mechanism
// (tests/test_pc_eval_autograd.py: f64 torch.autograd of the reference's formulas as written).
__global__ void __launch_bounds__(PC_BLOCK) psfm_pc_eval_kernel(const double* __restrict__ uv12, const double* __restrict__ ref1,
                                                                const double* __restrict__ ref2, const double* __restrict__ scale,
                                                                const float2* __restrict__ flow12, long long n, int W, int H,
                                                                double* __restrict__ res, double* __restrict__ jac)
{
    const long long i = (long long)blockIdx.x * PC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double x[4] = {uv12[4 * i], uv12[4 * i + 1], uv12[4 * i + 2], uv12[4 * i + 3]};
    const double s = scale[i];
    double r[6], j[4];
    pc_core_eval<true>((const PcF2*)flow12, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, j);
    if (res) for (int k = 0; k < 6; ++k) res[6 * i + k] = r[k];
    if (jac) {
        double* J = jac + 24 * i;
        for (int k = 0; k < 24; ++k) J[k] = 0.0;
        J[0] = 1.0; J[5] = 1.0; J[10] = s; J[15] = s;              // d r0 / d x0, d r1 / d x1, d r2 / d x2, d r3 / d x3
        J[16] = j[0]; J[17] = j[1]; J[18] = 1.0;                   // r4 = (x2 - x0) - F12.x(row = x1, col = x0)
        J[20] = j[2]; J[21] = j[3]; J[23] = 1.0;                   // r5 = (x3 - x1) - F12.y
    }
}

psfm_status psfm_launch_pc_eval(const double* uv12, const double* ref1, const double* ref2, const double* scale, const float* flow12,
                                int64_t n, int w, int h, double* res, double* jac, hipStream_t s)
{
    if (n == 0) return PSFM_OK;
    const unsigned nb = (unsigned)((n + PC_BLOCK - 1) / PC_BLOCK);
    hipLaunchKernelGGL(psfm_pc_eval_kernel, dim3(nb), dim3(PC_BLOCK), 0, s, uv12, ref1, ref2, scale, (const float2*)flow12,
                       (long long)n, w, h, res, jac);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_solve_batch(psfm_ctx* c, const double* uv12, const double* ref1, const double* ref2,
                             const double* scale, const float* flow12, int64_t n, int w, int h, double* out,
                             psfm_solve_stats* st, hipStream_t s)
{
    if (n == 0) return PSFM_OK;
    if (n > 0x3fffffff) { psfm_set_error("psfm_optimize_location: n too large"); return PSFM_ERR_ARG; }
    psfm_status rc;
    if ((rc = pc_workspace(c, n, 2)) != PSFM_OK) return rc;
    if ((rc = c->sol_misc.ensure(sizeof(double2) * n * 2)) != PSFM_OK) return rc;
    PcParams P;
    memset(&P, 0, sizeof(P));
    P.H = h; P.W = w;
    P.cw = (float)((double)(w - 1) / 2.0); P.ch = (float)((double)(h - 1) / 2.0);
    P.n_rows = (int)n;
    P.x1a = c->sol_misc.as<double2>();
    P.x2a = P.x1a + n;
    P.xs = c->sol_x.as<double2>(); P.xs_stride = n;
    P.ref1 = c->sol_state.as<double2>();
    P.ref2 = P.ref1 + n;
    P.jscale = P.ref2 + n;
    P.scale = (double*)(P.jscale + n);
    P.flow12 = (const float2*)flow12;
    if ((rc = pc_setup(c, P, s)) != PSFM_OK) return rc;
    const int n_blocks = pc_blocks(c, (int)n);
    const unsigned nb = (unsigned)((n + PC_BLOCK - 1) / PC_BLOCK);
    hipLaunchKernelGGL(psfm_pc_load_rows_kernel, dim3(nb), dim3(PC_BLOCK), 0, s, uv12, n, P.x1a, P.x2a);
    PSFM_HIP(hipMemcpyAsync(P.ref1, ref1, sizeof(double2) * n, hipMemcpyDeviceToDevice, s));
    PSFM_HIP(hipMemcpyAsync(P.ref2, ref2, sizeof(double2) * n, hipMemcpyDeviceToDevice, s));
    PSFM_HIP(hipMemcpyAsync(P.scale, scale, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(psfm_pc_init_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
    bool resident = false;
    if (!(resident = pc_persist_enqueue(c, P, n_blocks, out, false, false, s))) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(psfm_pc_iter_kernel, dim3(n_blocks), dim3(PC_BLOCK), 0, s, P);
        c->n_iter_launches += 4;
    }
    PSFM_HIP(hipGetLastError());
    rc = pc_finish_sync(c, P, n_blocks, out, st, s, resident);
    if (rc != PSFM_OK) return rc;
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}
