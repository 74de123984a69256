// psfm_pc_reduce.h -- the block-level reduction of the path-consistency solver's 13 sums (psfm_solver.hip), in a header of its
// own so that scripts/micro/dpp_check.hip can run exactly this code against integer-valued data on the device.
#pragma once
#include <hip/hip_runtime.h>
#include "psfm_pc_core.h"

#ifndef PC_BLOCK
#define PC_BLOCK 256
#endif
#ifndef PSFM_WAVE
#define PSFM_WAVE 64
#endif

// Block reduction of acc[NS_] (slot SUM_GMAX by max, the others by sum) in a fixed order, mostly in registers.  Inside a
// 16-lane row the twelve slots that are ADDED are reduced "transposed": in the exchange over lane bit 0 a lane keeps six of its
// twelve values and receives the partner's versions of those six (it sends the other six), over bit 1 three of six, over bit 2
// two of four (the fourth being the running maximum of SUM_GMAX), over bit 3 one of two -- 12 + 2 exchanged values per lane
// instead of 13 x 4, every sum formed in exactly one place (fixed order), and afterwards lane (b0, b1, b2, b3) of a row holds the
// row's total of list entry 6 b0 + 3 b1 + 2 b2 + b3 (2 b2 + b3 = 3: the row's maximum).  The 4 rows x 4 waves = 16 row totals
// per slot go through 1.7 KB of LDS and are added in order by one thread per slot.  (Round 3 parked every thread's accumulators
// in LDS -- 27 KB, which the resident solve needs for its tracks; a ds_bpermute tree took ~4 us of every launch's tail; a plain
// DPP tree per sum with v_readlane for the rows was 480 instructions per wave, as much as the arithmetic of three tracks.)
// The list of added slots is the same for every NS_ (absent slots are zeros), so the launch chain (PC_NSUM slots) and a round
// of the resident solve (the first PC_RES_SUMS) give the same bits.  out[0 .. NS_): LDS, valid for every thread after the call.
// scripts/micro/dpp_check.hip checks the exchange patterns on the device with integer-valued data.
template <int BIT>
__device__ __forceinline__ double pc_xchg(double v)      // the value of the lane whose id differs in bit BIT (of the 16-lane row)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
    if (BIT == 0)      { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
    else if (BIT == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
    else if (BIT == 2) { lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F); }                                          // bit mode: lane ^ 4
    else               { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 0xf, false); } // row_ror:8 (= lane ^ 8)
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
#define PC_ADDED 12      // slots that are added: all of PC_NSUM but SUM_GMAX, in slot order
__host__ __device__ constexpr int pc_added_slot(int i) { return i < SUM_GMAX ? i : i + 1; }
// The exchange tree over the 16 lanes of a row: returns the row total this lane ends up holding and the slot it belongs to
// (PC_NSUM: the spare column -- the rows' maximum lands in four lanes, one of them keeps it under SUM_GMAX).
template <int NS_>
__device__ __forceinline__ double pc_row_tree(const double* acc, int& slot)
{
    static_assert(PC_NSUM == PC_ADDED + 1 && SUM_GMAX == 5, "the exchange tree below is laid out for 12 added slots + the maximum");
    const int lane = threadIdx.x & (PSFM_WAVE - 1);
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    double c[6], e[4], f[2];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const double lo = pc_added_slot(q) < NS_ ? acc[pc_added_slot(q)] : 0.0, hi = pc_added_slot(q + 6) < NS_ ? acc[pc_added_slot(q + 6)] : 0.0;
        c[q] = (b0 ? hi : lo) + pc_xchg<0>(b0 ? lo : hi);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) e[q] = (b1 ? c[q + 3] : c[q]) + pc_xchg<1>(b1 ? c[q] : c[q + 3]);
    {
        double m = SUM_GMAX < NS_ ? acc[SUM_GMAX] : 0.0;
        m = fmax(m, pc_xchg<0>(m));
        e[3] = fmax(m, pc_xchg<1>(m));
    }
    f[0] = (b2 ? e[2] : e[0]) + pc_xchg<2>(b2 ? e[0] : e[2]);
    {
        const double keep = b2 ? e[3] : e[1], recv = pc_xchg<2>(b2 ? e[1] : e[3]);
        f[1] = b2 ? fmax(keep, recv) : keep + recv;
    }
    double g;
    {
        const double keep = b3 ? f[1] : f[0], recv = pc_xchg<3>(b3 ? f[0] : f[1]);
        g = (b2 && b3) ? fmax(keep, recv) : keep + recv;
    }
    const int t = (b2 ? 2 : 0) + (b3 ? 1 : 0);
    const int idx = (b0 ? 6 : 0) + (b1 ? 3 : 0) + t;
    slot = t == 3 ? ((b0 || b1) ? PC_NSUM : SUM_GMAX) : (idx < SUM_GMAX ? idx : idx + 1);
    return g;
}

template <int NS_>
__device__ __forceinline__ void pc_block_sums(const double* acc, double* out)
{
    __shared__ double s_row[4 * (PC_BLOCK / PSFM_WAVE)][PC_NSUM + 1];
    const int tid = threadIdx.x;
    int slot;
    const double g = pc_row_tree<NS_>(acc, slot);
    s_row[tid >> 4][slot] = g;
    __syncthreads();
    // the 16 row totals of a slot, in row order: the added slots by the first lanes of wave 0, the maximum by the first lane of wave 1 --
    // one wave adds, another takes maxima (round 6: one loop that formed BOTH for every row and selected was 105 instructions on the
    // critical path of every launch / round; the order of the additions is unchanged)
    if (tid < NS_ && tid != SUM_GMAX) {
        double v = s_row[0][tid];
#pragma unroll
        for (int q = 1; q < 4 * (PC_BLOCK / PSFM_WAVE); ++q) v = v + s_row[q][tid];
        out[tid] = v;
    } else if (tid == PSFM_WAVE && SUM_GMAX < NS_) {
        double v = s_row[0][SUM_GMAX];
#pragma unroll
        for (int q = 1; q < 4 * (PC_BLOCK / PSFM_WAVE); ++q) v = fmax(v, s_row[q][SUM_GMAX]);
        out[SUM_GMAX] = v;
    }
    __syncthreads();
}
