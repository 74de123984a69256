// psfm_sort.hip -- the finalize's record sort: stable LSD radix sort of (32-bit key, lane) pairs, 8 bits per pass.
//
// The trajectory ids of the reference are the rank of a track by (death step, birth frame, birth grid index)
// (trajectory.py:129-158): psfm_finalize.hip packs the three into one key and sorts the records.  At the
// headline shape that is 2.07 M pairs, 32 key bits.  rocPRIM's device sort spends 4 passes of 30 us, a 16 us histogram and
// NINE 5-us fill launches on it (profiles/r06/r06_w_*): ~190 us of a 590-us finalize for 33 MB that cross the memory
// twice per pass (7 us at the copy ceiling).  This file is the same algorithm laid out for this size on this device:
//
//   per pass   H  one block per GROUP of 8 tiles of 4096 keys (two waves per tile): the digit counts of the tiles (LDS adds, one
//                 counter row per tile) as exclusive prefixes over the tiles of the group -- one row of 256 per tile -- and
//                 their sum as the group's row
//              S  one block per tile: where digit d of tile b goes = the digits below d (all tiles) + digit d of the tiles
//                 in front of b -- the group rows (64 in flight at once, behind the tile's own loads) + the tile's own prefix
//                 row; ranks by wave-wide digit matching (8 ballots per round, an LDS counter row per wave: no LDS atomics);
//                 the tile is put in order in LDS and leaves as runs of equal digits
//   Two launches per pass, no atomics on global memory, no fills, nothing to zero between passes: 5.7 + 17.6 us per pass at
//   the headline shape (four passes 93 us) against rocPRIM's 45.  Measured on the way (profiles/r06/r06_v_*,
//   scripts/micro/sort_whatif.sh): every tile adding its counts to group rows and totals with global adds (130 k adds on
//   4 352 addresses) made H a 19-us kernel; one block per tile + the group step by the block that finishes last in its group
//   (write-through rows, a ticket, L2-bypassing loads) put a 9-us tail on H, 42 us with release / acquire fences in it (a
//   release writes back an L2 full of the previous pass's output); one block per tile + the group step as a launch of its
//   own 4.7 + 4.3 us -- any launch costs 4.4 us here (a kernel that only loads its tile), which is what decides between
//   these forms, not the 16 MB a pass moves.
//
// Tiles are read "wave-striped" (wave w, round i, lane l <-> element 1024 w + 64 i + l), so the order (wave, round, lane)
// in which a digit's elements are ranked IS their order in memory: every pass is stable, which is what LSD needs.
// 506 tiles at the headline shape: two resident blocks per CU in S, no look-back chain, no device-wide ordering between tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "psfm_device.h"
#include "psfm_internal.h"

#define PS_BLOCK 256
#define PS_ITEMS 16
#define PS_TILE (PS_BLOCK * PS_ITEMS)
#define PS_NW (PS_BLOCK / PSFM_WAVE)
#define PS_WSPAN (PS_ITEMS * PSFM_WAVE)      // elements of a tile one wave reads
#define PS_GROUP 8                           // tiles per group row
#define PS_DIGITS 256
#define PS_GCHUNK 64                         // group rows the scatter kernel has in flight at once
#ifndef PS_WHATIF
#define PS_WHATIF 0      // timing builds (scripts/micro/sort_whatif.sh): 1 H without the match loop, 2 S without the rank loop, 4 S without its stores,
#endif                  // 8 H leaves after its loads, 16 S leaves after its loads (results are wrong with any of them)

// lanes of the wave whose digit equals mine (every lane of the wave takes part)
__device__ __forceinline__ unsigned long long ps_match8(unsigned d)
{
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const bool on = (d >> bit) & 1u;
        const unsigned long long m = __ballot(on);
        peers &= on ? m : ~m;
    }
    return peers;
}

// H.  Block = group of PS_GROUP tiles (PS_HWAVES waves on every tile): the digit counts of every tile of the group as exclusive prefixes
// over the tiles of the group -> rows of cnt; their sum -> the group's row of grp.  (One block per TILE + a launch that turns 32 rows
// into prefixes and a group row: 4.7 + 4.3 us.)
#define PS_HWAVES 2
#ifndef PS_HREP
#define PS_HREP 4                             // copies of a tile's counter row (a power of two)
#endif
#define PS_HBLOCK (PS_GROUP * PS_HWAVES * PSFM_WAVE)
#define PS_HITEMS (PS_TILE / (PS_HWAVES * PSFM_WAVE))
__global__ __launch_bounds__(PS_HBLOCK) void psfm_sort_hist_kernel(const unsigned* __restrict__ keys, int64_t n, int shift,
                                                                   unsigned* __restrict__ cnt, unsigned* __restrict__ grp, int nb)
{
    __shared__ unsigned s_h[PS_GROUP][PS_HREP][PS_DIGITS];
    const int tid = threadIdx.x, lane = tid & (PSFM_WAVE - 1), wave = tid / PSFM_WAVE;
    for (int q = tid; q < PS_GROUP * PS_HREP * PS_DIGITS; q += PS_HBLOCK) (&s_h[0][0][0])[q] = 0u;
    const int t = wave / PS_HWAVES, part = wave % PS_HWAVES;      // this wave's tile of the group, and which part of it
    const int b = blockIdx.x * PS_GROUP + t;
    const int64_t tile0 = (int64_t)b * PS_TILE;
    const int nv = b < nb ? (int)(n - tile0 < PS_TILE ? n - tile0 : PS_TILE) : 0;
    unsigned k[PS_HITEMS];
#pragma unroll
    for (int i = 0; i < PS_HITEMS; ++i) {                  // all loads in flight before the first add
        const int off = (part * PS_HITEMS + i) * PSFM_WAVE + lane;
        k[i] = off < nv ? keys[tile0 + off] : 0u;
    }
    __syncthreads();
    if (PS_WHATIF & 8) { unsigned x = 0; for (int i = 0; i < PS_HITEMS; ++i) x ^= k[i]; if (x == 0x12345u) cnt[tid] = x; return; }
#pragma unroll
    for (int i = 0; i < PS_HITEMS; ++i) {
        if (PS_WHATIF & 1) { if (k[i] == 0x12345u) s_h[t][0][0] = 1; continue; }
        const int off = (part * PS_HITEMS + i) * PSFM_WAVE + lane;
        // one LDS add per key, nobody waits for it.  Lanes that meet on a counter are served one after the other: the records' top
        // digit is skewed (a quarter of them die in the last frame) and made this kernel 13 us on real keys against 5.7 on random ones
        // -> PS_HREP copies of the row, picked by lane, and one add for a round whose keys all agree.  (Matching the lanes by digit
        // first and adding once per digit -- the scatter kernel's way -- is 65 instructions per key on the quarter of the CUs this
        // launch occupies.)
        const bool ok = off < nv;
        const unsigned d = (k[i] >> shift) & 255u;
        const unsigned long long valid = __ballot(ok);
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        if (__ballot(ok && d == d0) == valid) {
            if (lane == 0 && valid) atomicAdd(&s_h[t][0][d0], (unsigned)__popcll(valid));
        } else if (ok) {
            atomicAdd(&s_h[t][lane & (PS_HREP - 1)][d], 1u);
        }
    }
    __syncthreads();
    if (tid < PS_DIGITS) {
        unsigned run = 0;
#pragma unroll
        for (int q = 0; q < PS_GROUP; ++q) {
            unsigned c = 0;
#pragma unroll
            for (int rep = 0; rep < PS_HREP; ++rep) c += s_h[q][rep][tid];
            if (blockIdx.x * PS_GROUP + q < nb) cnt[(int64_t)(blockIdx.x * PS_GROUP + q) * PS_DIGITS + tid] = run;
            run += c;
        }
        grp[blockIdx.x * PS_DIGITS + tid] = run;
    }
}

// S.  Block = tile.
#ifndef PS_CHAINS
#define PS_CHAINS 1      // (2 and 4 measured: 15.9 / 16.1 us against 15.7 -- the chain is not what the kernel waits for)
#endif
// the rounds of a wave in PS_CHAINS runs, each with a counter row of its own: their read-then-write LDS chains (one round trip per
// round) run side by side
#define PS_CROUNDS (PS_ITEMS / PS_CHAINS)
__global__ __launch_bounds__(PS_BLOCK) void psfm_sort_scatter_kernel(const unsigned* __restrict__ kin, const int* __restrict__ vin,
                                                                     unsigned* __restrict__ kout, int* __restrict__ vout, int64_t n,
                                                                     int shift, const unsigned* __restrict__ cnt,
                                                                     const unsigned* __restrict__ grp, int ngrp)
{
    __shared__ unsigned s_k[PS_TILE];
    __shared__ int s_v[PS_TILE];
    __shared__ unsigned s_wc[PS_NW][PS_CHAINS][PS_DIGITS];   // per wave and run of rounds: digit counts, then where that share of a digit starts in the tile
    __shared__ int s_delta[PS_DIGITS];               // digit -> (position in the output) - (position in the tile)
    __shared__ unsigned s_wsum[2][PS_NW];
    const int tid = threadIdx.x, lane = tid & (PSFM_WAVE - 1), wave = tid / PSFM_WAVE;
    const int b = blockIdx.x;
    const int64_t tile0 = (int64_t)b * PS_TILE;
    const int nv = (int)(n - tile0 < PS_TILE ? n - tile0 : PS_TILE);
#pragma unroll
    for (int w = 0; w < PS_NW; ++w)
#pragma unroll
        for (int ch = 0; ch < PS_CHAINS; ++ch) s_wc[w][ch][tid] = 0u;
    unsigned k[PS_ITEMS];
    int v[PS_ITEMS];
#pragma unroll
    for (int i = 0; i < PS_ITEMS; ++i) {
        const int off = wave * PS_WSPAN + i * PSFM_WAVE + lane;
        const bool ok = off < nv;
        k[i] = ok ? kin[tile0 + off] : 0xffffffffu;      // padding: digit 255 in every pass, behind every key of the tile
        v[i] = ok ? vin[tile0 + off] : 0;
    }
    // digit `tid` of the tiles in front of this one (whole groups; the tiles of its own group: H left that prefix in the tile's row),
    // and of all tiles -- behind the tile's own loads in program order, so that those are in flight while the rows arrive
    unsigned before = cnt[(int64_t)b * PS_DIGITS + tid], all = 0;
    {
        const int g = b / PS_GROUP;
        for (int q0 = 0; q0 < ngrp; q0 += PS_GCHUNK) {      // (one round of loads per chunk: 64 group rows = 2 M keys in one round)
            unsigned row[PS_GCHUNK];
#pragma unroll
            for (int u = 0; u < PS_GCHUNK; ++u) row[u] = q0 + u < ngrp ? grp[(q0 + u) * PS_DIGITS + tid] : 0u;
#pragma unroll
            for (int u = 0; u < PS_GCHUNK; ++u) { all += row[u]; if (q0 + u < g) before += row[u]; }
        }
    }
    __syncthreads();
    if (PS_WHATIF & 16) { unsigned x = before ^ all; for (int i = 0; i < PS_ITEMS; ++i) x ^= k[i] ^ (unsigned)v[i]; if (x == 0x12345u) kout[tid] = x; return; }
    // rank of every element among the elements of its digit in its wave and run, in (round, lane) order
    unsigned r[PS_ITEMS];
    {
        const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < PS_CROUNDS; ++i) {
            unsigned d[PS_CHAINS], old[PS_CHAINS];
            unsigned long long peers[PS_CHAINS];
#pragma unroll
            for (int ch = 0; ch < PS_CHAINS; ++ch) {
                if (PS_WHATIF & 2) k[ch * PS_CROUNDS + i] &= ~(255u << shift);      // (everything digit 0; only with 4: no stores)
                d[ch] = (k[ch * PS_CROUNDS + i] >> shift) & 255u;
                peers[ch] = (PS_WHATIF & 2) ? ~0ull : ps_match8(d[ch]);
            }
            // every lane reads its digit's counter, the first of the peers writes it back: plain LDS accesses -- the wave runs them in
            // program order, and the compiler keeps a load behind a store it may alias (through a volatile generic pointer these were
            // flat loads / stores with a wait after each: the rank loop then took 5.7 us of the kernel's 16)
#pragma unroll
            for (int ch = 0; ch < PS_CHAINS; ++ch) old[ch] = s_wc[wave][ch][d[ch]];
#pragma unroll
            for (int ch = 0; ch < PS_CHAINS; ++ch) {
                const unsigned long long below = peers[ch] & lower;
                r[ch * PS_CROUNDS + i] = old[ch] + (unsigned)__popcll(below);
                if (below == 0ull) s_wc[wave][ch][d[ch]] = old[ch] + (unsigned)__popcll(peers[ch]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // thread = digit: the tile's run of the digit starts behind the smaller digits; the shares of the waves and their runs of rounds
    // follow each other inside it
    {
        unsigned c[PS_NW][PS_CHAINS], mine = 0;
#pragma unroll
        for (int w = 0; w < PS_NW; ++w)
#pragma unroll
            for (int ch = 0; ch < PS_CHAINS; ++ch) { c[w][ch] = s_wc[w][ch][tid]; mine += c[w][ch]; }
        unsigned inc_t = mine, inc_a = all;               // inclusive scans over the digits: tile counts, global counts
#pragma unroll
        for (int o = 1; o < PSFM_WAVE; o <<= 1) {
            const unsigned a = __shfl_up(inc_t, o), g2 = __shfl_up(inc_a, o);
            if (lane >= o) { inc_t += a; inc_a += g2; }
        }
        if (lane == PSFM_WAVE - 1) { s_wsum[0][wave] = inc_t; s_wsum[1][wave] = inc_a; }
        __syncthreads();
        unsigned base_t = 0, base_a = 0;
#pragma unroll
        for (int w = 0; w < PS_NW; ++w) if (w < wave) { base_t += s_wsum[0][w]; base_a += s_wsum[1][w]; }
        const unsigned dstart = base_t + inc_t - mine;        // first position of the digit in the ordered tile
        const unsigned gstart = base_a + inc_a - all;         // first position of the digit in the output
        unsigned run = dstart;
#pragma unroll
        for (int w = 0; w < PS_NW; ++w)
#pragma unroll
            for (int ch = 0; ch < PS_CHAINS; ++ch) { s_wc[w][ch][tid] = run; run += c[w][ch]; }
        s_delta[tid] = (int)(gstart + before) - (int)dstart;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PS_ITEMS; ++i) {
        const unsigned d = (k[i] >> shift) & 255u;
        const unsigned q = s_wc[wave][i / PS_CROUNDS][d] + r[i];
        s_k[q] = k[i];
        s_v[q] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PS_ITEMS; ++j) {
        const int q = j * PS_BLOCK + tid;
        if (q < nv) {
            const unsigned key = s_k[q];
            const int64_t pos = (int64_t)s_delta[(key >> shift) & 255u] + q;
            if (PS_WHATIF & 4) { if (key == 0x12345u && pos == 77) kout[0] = key; continue; }
            kout[pos] = key;
            vout[pos] = s_v[q];
        }
    }
}

int psfm_sort_pairs32_passes(unsigned end_bit) { return (int)((end_bit + 7u) / 8u); }

// Sorts n (key, value) pairs by the key bits [0, end_bit).  `half0` / `half1`: two buffers of n entries each; the input sits in
// half0 when the number of passes is even, in half1 when it is odd (psfm_sort_pairs32_passes), and the result always ends in half0.
// n < 2^31.  Uses c->sort_tmp.
psfm_status psfm_sort_pairs32(psfm_ctx* c, unsigned* k_half0, int* v_half0, unsigned* k_half1, int* v_half1, int64_t n,
                              unsigned end_bit, hipStream_t s)
{
    const int passes = psfm_sort_pairs32_passes(end_bit);
    if (n <= 0 || passes == 0) return PSFM_OK;
    const int64_t nb = (n + PS_TILE - 1) / PS_TILE;
    const int64_t ngrp = (nb + PS_GROUP - 1) / PS_GROUP;
    const size_t cnt_bytes = sizeof(unsigned) * PS_DIGITS * (size_t)nb;
    const size_t grp_bytes = sizeof(unsigned) * PS_DIGITS * (size_t)ngrp;
    psfm_status st = c->sort_tmp.ensure(cnt_bytes + grp_bytes);
    if (st != PSFM_OK) return st;
    unsigned* cnt = c->sort_tmp.as<unsigned>();
    unsigned* grp = (unsigned*)((char*)c->sort_tmp.p + cnt_bytes);
    unsigned* kin = (passes & 1) ? k_half1 : k_half0;
    int* vin = (passes & 1) ? v_half1 : v_half0;
    unsigned* kout = (passes & 1) ? k_half0 : k_half1;
    int* vout = (passes & 1) ? v_half0 : v_half1;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(psfm_sort_hist_kernel, dim3((unsigned)ngrp), dim3(PS_HBLOCK), 0, s, (const unsigned*)kin, n, 8 * p, cnt, grp, (int)nb);
        hipLaunchKernelGGL(psfm_sort_scatter_kernel, dim3((unsigned)nb), dim3(PS_BLOCK), 0, s, (const unsigned*)kin, (const int*)vin, kout,
                           vout, n, 8 * p, (const unsigned*)cnt, (const unsigned*)grp, (int)ngrp);
        unsigned* tk = kin; kin = kout; kout = tk;
        int* tv = vin; vin = vout; vout = tv;
    }
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}
