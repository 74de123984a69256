// psfm_finalize.hip -- K7: trajectory ids, lengths and the id-ordered CSR result.
//
// Reference semantics: full_trajs lists, for every step f ascending, the tracks that
// failed at f in active-list order, then all still-active tracks (trajectory.py:138-147,154-158); the
// active list is always sorted by (birth_frame, birth grid index).  Hence
//     id = rank under the key (last_valid_time, birth_frame, birth_grid_index)
// and the saved id is that rank (main_connect_point_trajectories.py:56-60).  The frame loop recorded one
// (key, lane) pair per finished track; here the survivors are appended, the pairs are radix-sorted by
// key (rocPRIM device radix sort on the used key bits only), lengths are scanned into offsets, and the
// frame-major log is transposed into a track-major (n_points,2) f64 array through LDS tiles so that
// both the slab reads (consecutive lanes) and the result writes (consecutive times) are coalesced.
#include <cstring>
#include <stdio.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "psfm_device.h"
#include "psfm_internal.h"

#define PSFM_BLOCK 256

__device__ __forceinline__ void psfm_collect_alive_body(
    const int* __restrict__ birth_frame, const int* __restrict__ birth_idx, PsfmCounters* __restrict__ ctr,
    PsfmShard* __restrict__ shards, unsigned long long* __restrict__ fin_keys, int* __restrict__ fin_lanes, int cap,
    int shard_cap, int last_time, int shift_b, int shift_d)
{
    __shared__ int s_cnt[PSFM_BLOCK / PSFM_WAVE];
    __shared__ int s_base;
    const int n_lanes = min(ctr->n_lanes, cap);
    if ((int)(blockIdx.x * PSFM_BLOCK) >= n_lanes) return;
    const int i = blockIdx.x * PSFM_BLOCK + threadIdx.x;
    int bf = -1;
    if (i < n_lanes) bf = birth_frame[i];
    // bf >= 0: still active at the end (clear_active, trajectory.py:154-158) -> last time = n_flows;
    // bf <= -2: marked dead by the LAST chain_step launch (record not yet written) -> last time = n_flows - 1
    const bool pending = bf <= -2;
    const bool alive = (bf >= 0) | pending;
    const int my_last = pending ? last_time - 1 : last_time;
    if (pending) bf = -2 - bf;
    const unsigned long long am = __ballot(alive);
    const int lane = psfm_lane_id(), wave = threadIdx.x / PSFM_WAVE;
    if (lane == 0) s_cnt[wave] = __popcll(am);
    __syncthreads();
    const int shard = blockIdx.x % PSFM_NSHARD;
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < PSFM_BLOCK / PSFM_WAVE; ++w) tot += s_cnt[w];
        s_base = tot > 0 ? atomicAdd(&shards[shard].fin_cnt, tot) : 0;
    }
    __syncthreads();
    if (alive) {
        int r = s_base + psfm_rank_in(am);
        for (int w = 0; w < wave; ++w) r += s_cnt[w];
        if (r < shard_cap) {
            const int64_t o = (int64_t)shard * shard_cap + r;
            fin_keys[o] = ((unsigned long long)my_last << shift_d) | ((unsigned long long)bf << shift_b) |
                          (unsigned long long)birth_idx[i];
            fin_lanes[o] = i;
        } else {
            atomicOr(&ctr->overflow, 2);
        }
    }
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_collect_alive_kernel(
    const int* __restrict__ birth_frame, const int* __restrict__ birth_idx, PsfmCounters* __restrict__ ctr,
    PsfmShard* __restrict__ shards, unsigned long long* __restrict__ fin_keys, int* __restrict__ fin_lanes, int cap,
    int shard_cap, int last_time, int shift_b, int shift_d)
{
    psfm_collect_alive_body(birth_frame, birth_idx, ctr, shards, fin_keys, fin_lanes, cap, shard_cap, last_time, shift_b, shift_d);
}

// The sort key (last << shift_d | birth << shift_b | idx) has birth <= last: the pair (last, birth) is re-coded as the
// triangular index last*(last+1)/2 + birth, which orders like the pair and needs ~1 bit less.  When that makes the key
// fit 32 bits (100 frames at 1080p/r=2: 13 + 19) the radix sort runs on 4-byte keys in 4 passes instead of 8-byte
// keys in 5.
struct PsfmKeyFmt { int shift_b, shift_d; int use32; };
__device__ __forceinline__ unsigned psfm_key32(unsigned long long k, const PsfmKeyFmt& f)
{
    const unsigned last = (unsigned)(k >> f.shift_d);
    const unsigned birth = (unsigned)((k >> f.shift_b) & ((1ull << (f.shift_d - f.shift_b)) - 1ull));
    const unsigned idx = (unsigned)(k & ((1ull << f.shift_b) - 1ull));
    return ((last * (last + 1u) / 2u + birth) << f.shift_b) | idx;
}
__device__ __forceinline__ void psfm_put_key(unsigned long long* keys, int64_t pos, unsigned long long k, const PsfmKeyFmt& f)
{
    if (f.use32) ((unsigned*)keys)[pos] = psfm_key32(k, f);
    else keys[pos] = k;
}

// gather the per-shard record slices into one contiguous (key, lane) array for the sort
struct PsfmShardOffsets { int count[PSFM_NSHARD]; int64_t start[PSFM_NSHARD]; };
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_compact_shards_kernel(
    const unsigned long long* __restrict__ fin_keys, const int* __restrict__ fin_lanes, int shard_cap,
    PsfmShardOffsets so, unsigned long long* __restrict__ keys, int* __restrict__ lanes, PsfmKeyFmt fmt)
{
    const int shard = blockIdx.y;
    const int n = so.count[shard];
    for (int i = blockIdx.x * PSFM_BLOCK + threadIdx.x; i < n; i += gridDim.x * PSFM_BLOCK) {
        psfm_put_key(keys, so.start[shard] + i, fin_keys[(int64_t)shard * shard_cap + i], fmt);
        lanes[so.start[shard] + i] = fin_lanes[(int64_t)shard * shard_cap + i];
    }
}

// sorted record i (== trajectory id) -> birth, length (i32) and length as i64 for the offset scan
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_decode_kernel(const unsigned long long* __restrict__ keys, int64_t n,
                                                                 int shift_b, int shift_d, int* __restrict__ birth,
                                                                 int* __restrict__ len, int64_t* __restrict__ len64)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i > n) return;
    if (i == n) { len64[i] = 0; return; }
    const unsigned long long k = keys[i];
    const int last = (int)(k >> shift_d);
    const int b = (int)((k >> shift_b) & ((1ull << (shift_d - shift_b)) - 1ull));
    birth[i] = b;
    len[i] = last - b + 1;
    len64[i] = (int64_t)(last - b + 1);
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_decode32_kernel(const unsigned* __restrict__ keys, int64_t n, int shift_b,
                                                                   int* __restrict__ birth, int* __restrict__ len,
                                                                   int64_t* __restrict__ len64)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i > n) return;
    if (i == n) { len64[i] = 0; return; }
    const unsigned tri = keys[i] >> shift_b;
    // last = the largest l with l*(l+1)/2 <= tri
    int l = (int)((sqrtf(8.0f * (float)tri + 1.0f) - 1.0f) * 0.5f);
    while (l > 0 && (unsigned)l * (unsigned)(l + 1) / 2u > tri) --l;
    while ((unsigned)(l + 1) * (unsigned)(l + 2) / 2u <= tri) ++l;
    const int b = (int)(tri - (unsigned)l * (unsigned)(l + 1) / 2u);
    birth[i] = b;
    len[i] = l - b + 1;
    len64[i] = (int64_t)(l - b + 1);
}

// ---- offsets without a scan over the trajectories ----
// Sorted records with the same (last, birth) -- a GROUP, numbered g = last (last + 1) / 2 + birth, the order of the sort -- have the same
// length, so the CSR offset of trajectory id is  goff[g] + (id - gstart[g]) * len(g):  one pass over the sorted keys marks where every
// group starts, ONE block turns the (n_flows + 2)(n_flows + 3) / 2 starts into counts and offsets, and the gather computes birth /
// length / offset of its 64 trajectories from their keys (and stores them for psfm_result_*).  That replaces the decode kernel and a
// three-kernel scan over n + 1 int64 (2 M entries at the headline shape: 45 us of launches for what 5 253 groups say).  Sequences
// with more groups than the plan block holds in LDS (n_flows > 178) keep the decode + scan form.
#define PSFM_PLAN_MAX_GROUPS 16384
#define PSFM_PLAN_BLOCK 1024
struct PsfmPlan {
    const int* gstart; const int64_t* goff;      // gstart == nullptr: birth / len / off are read from the arrays below instead
    const void* keys; int use32, shift_b, shift_d;
    int* birth_w; int* len_w; int64_t* off_w;    // where the gather stores what it computed (the result arrays)
};
__device__ __forceinline__ void psfm_tri_invert(unsigned tri, int* last, int* birth)
{
    // last = the largest l with l*(l+1)/2 <= tri
    int l = (int)((sqrtf(8.0f * (float)tri + 1.0f) - 1.0f) * 0.5f);
    while (l > 0 && (unsigned)l * (unsigned)(l + 1) / 2u > tri) --l;
    while ((unsigned)(l + 1) * (unsigned)(l + 2) / 2u <= tri) ++l;
    *last = l;
    *birth = (int)(tri - (unsigned)l * (unsigned)(l + 1) / 2u);
}
__device__ __forceinline__ void psfm_key_fields(const void* keys, int64_t id, int use32, int shift_b, int shift_d, int* last, int* birth, int* idx)
{
    if (use32) {
        const unsigned k = ((const unsigned*)keys)[id];
        psfm_tri_invert(k >> shift_b, last, birth);
        *idx = (int)(k & ((1u << shift_b) - 1u));
    } else {
        const unsigned long long k = ((const unsigned long long*)keys)[id];
        *last = (int)(k >> shift_d);
        *birth = (int)((k >> shift_b) & ((1ull << (shift_d - shift_b)) - 1ull));
        *idx = (int)(k & ((1ull << shift_b) - 1ull));
    }
}
__device__ __forceinline__ int psfm_key_group(const void* keys, int64_t id, int use32, int shift_b, int shift_d)
{
    if (use32) return (int)(((const unsigned*)keys)[id] >> shift_b);      // the 32-bit key carries the group number itself
    const unsigned long long k = ((const unsigned long long*)keys)[id];
    const unsigned last = (unsigned)(k >> shift_d);
    const unsigned birth = (unsigned)((k >> shift_b) & ((1ull << (shift_d - shift_b)) - 1ull));
    return (int)(last * (last + 1u) / 2u + birth);
}

// marks[g] = the first sorted record of group g (the array arrives filled with -1: groups nobody marks are empty)
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_group_bounds_kernel(const void* __restrict__ keys, int64_t n, int use32, int shift_b,
                                                                       int shift_d, int* __restrict__ gstart)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int g = psfm_key_group(keys, i, use32, shift_b, shift_d);
    const int gp = i > 0 ? psfm_key_group(keys, i - 1, use32, shift_b, shift_d) : -1;
    if (g != gp) gstart[g] = (int)i;
}

// ONE block: marks -> counts (the next non-empty group's start, or n, ends a group) -> goff[g] = points of all groups before g.
// The marks go back to -1 on the way (what the next call's bounds kernel expects); the gather reads the starts from gstart.
__global__ __launch_bounds__(PSFM_PLAN_BLOCK) void psfm_group_plan_kernel(int* __restrict__ marks, int* __restrict__ gstart,
                                                                          int64_t* __restrict__ goff, int ng, int64_t n,
                                                                          int64_t* __restrict__ off_n)
{
    __shared__ int s_g[PSFM_PLAN_MAX_GROUPS];          // starts, then counts
    __shared__ int s_wmin[PSFM_PLAN_BLOCK / PSFM_WAVE];
    __shared__ long long s_wsum[PSFM_PLAN_BLOCK / PSFM_WAVE];
    constexpr int NW = PSFM_PLAN_BLOCK / PSFM_WAVE;
    constexpr int INF = 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & (PSFM_WAVE - 1), wave = tid / PSFM_WAVE;
    for (int k = tid; k < ng; k += PSFM_PLAN_BLOCK) {
        const int v = marks[k];
        s_g[k] = v;
        gstart[k] = v;
        if (v >= 0) marks[k] = -1;
    }
    __syncthreads();
    const int per = (ng + PSFM_PLAN_BLOCK - 1) / PSFM_PLAN_BLOCK;
    const int g0 = min(tid * per, ng), g1 = min(g0 + per, ng);
    int first = INF;
    for (int g = g1 - 1; g >= g0; --g) if (s_g[g] >= 0) first = s_g[g];
    // the next non-empty start BEHIND this thread's groups: a suffix minimum (starts grow with g), wave by wave
    int sm = first;
#pragma unroll
    for (int o = 1; o < PSFM_WAVE; o <<= 1) {
        const int v = __shfl_down(sm, o);
        if (lane + o < PSFM_WAVE && v < sm) sm = v;
    }
    if (lane == 0) s_wmin[wave] = sm;
    int nxt = __shfl_down(sm, 1);
    if (lane == PSFM_WAVE - 1) nxt = INF;
    __syncthreads();
    for (int w = wave + 1; w < NW; ++w) nxt = min(nxt, s_wmin[w]);
    if (nxt == INF) nxt = (int)n;
    int last = 0, birth = 0;
    if (g0 < g1) psfm_tri_invert((unsigned)g0, &last, &birth);
    // counts (walking back), then this thread's points (walking forward with (last, birth) of every group)
    for (int g = g1 - 1; g >= g0; --g) {
        const int st = s_g[g];
        if (st >= 0) { s_g[g] = nxt - st; nxt = st; } else s_g[g] = 0;
    }
    long long pts = 0;
    {
        int l = last, b = birth;
        for (int g = g0; g < g1; ++g) {
            pts += (long long)s_g[g] * (long long)(l - b + 1);
            if (++b > l) { ++l; b = 0; }
        }
    }
    long long inc = pts;                                  // inclusive prefix sum over the threads, wave by wave
#pragma unroll
    for (int o = 1; o < PSFM_WAVE; o <<= 1) {
        const long long v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == PSFM_WAVE - 1) s_wsum[wave] = inc;
    __syncthreads();
    long long base = 0, total = 0;
    for (int w = 0; w < NW; ++w) { const long long v = s_wsum[w]; if (w < wave) base += v; total += v; }
    long long run = base + inc - pts;
    {
        int l = last, b = birth;
        for (int g = g0; g < g1; ++g) {
            goff[g] = run;
            run += (long long)s_g[g] * (long long)(l - b + 1);
            if (++b > l) { ++l; b = 0; }
        }
    }
    if (tid == 0) *off_n = total;      // off[n] = all points
}

// what the gather knows of its track `id`, in two steps.  First what its READS need -- birth and length, from the key alone (or the
// decoded arrays); then, with the first chunk's loads already in flight, what its WRITES need -- the offset, two dependent look-ups in
// the plan -- and the three values stored for psfm_result_*.
__device__ __forceinline__ void psfm_track_header(const PsfmPlan& pl, const int* __restrict__ birth, const int* __restrict__ len,
                                                  int64_t id, int* b, int* l, int* idx)
{
    if (pl.gstart) {
        int last;
        psfm_key_fields(pl.keys, id, pl.use32, pl.shift_b, pl.shift_d, &last, b, idx);
        *l = last - *b + 1;
    } else {
        *b = birth[id]; *l = len[id]; *idx = -1;
    }
}
__device__ __forceinline__ int64_t psfm_track_offset(const PsfmPlan& pl, const int64_t* __restrict__ off, int64_t id, int b, int l)
{
    if (!pl.gstart) return off[id];
    const int last = b + l - 1;
    const int g = last * (last + 1) / 2 + b;
    const int64_t o = pl.goff[g] + (id - (int64_t)pl.gstart[g]) * (int64_t)l;
    pl.birth_w[id] = b; pl.len_w[id] = l; pl.off_w[id] = o;
    return o;
}

// Transpose gather: a block owns TILE_J consecutive ids and walks time in chunks of TILE_K steps.
#ifndef TILE_J
#define TILE_J 64
#endif
#ifndef TILE_K
#define TILE_K 32
#endif
__device__ __forceinline__ void psfm_gather_body(const double2* __restrict__ log, int64_t cap,
                                                                 const int* __restrict__ lanes,
                                                                 const int* __restrict__ birth,
                                                                 const int* __restrict__ len,
                                                                 const int64_t* __restrict__ off, int64_t n,
                                                                 double2* __restrict__ out, const PsfmPlan& pl)
{
    __shared__ double2 tile[TILE_J][TILE_K + 1];
    __shared__ int s_lane[TILE_J], s_birth[TILE_J], s_len[TILE_J];
    __shared__ int64_t s_off[TILE_J];
    __shared__ int s_maxlen;
    const int tid = threadIdx.x;
    const int64_t id0 = (int64_t)blockIdx.x * TILE_J;
    int hb = 0, hl = 0;
    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (tid < TILE_J) {
        const int64_t id = id0 + tid;
        const bool ok = id < n;
        int b = 0, l = 0, idx;
        const int my_lane = ok ? lanes[id] : 0;
        if (ok) psfm_track_header(pl, birth, len, id, &b, &l, &idx);
        hb = b; hl = l;
        s_lane[tid] = my_lane;
        s_birth[tid] = b;
        s_len[tid] = l;
        if (ok) atomicMax(&s_maxlen, l);
    }
    __syncthreads();
    const int maxlen = s_maxlen;
    const int j = tid & (TILE_J - 1);   // track within the tile (read phase: lanes fastest)
    const int q = tid / TILE_J;         // 0..3
    constexpr int NLD = TILE_K * TILE_J / PSFM_BLOCK;   // tile entries a thread loads per chunk
    const int lj = s_len[j];
    const int64_t col = s_lane[j];
    const int bj = s_birth[j];
    // read: for a fixed time, 64 consecutive ids -> (mostly) consecutive lanes of one slab; the loads of chunk c+1 are
    // issued before chunk c is written out, so the read latency hides behind the writes
    double2 r[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int t = q + u * (PSFM_BLOCK / TILE_J);
        r[u] = t < lj ? log[(int64_t)(bj + t) * cap + col] : make_double2(0.0, 0.0);
    }
    if (tid < TILE_J) s_off[tid] = id0 + tid < n ? psfm_track_offset(pl, off, id0 + tid, hb, hl) : 0;   // (seen behind the loop's first barrier)
    for (int k0 = 0; k0 < maxlen; k0 += TILE_K) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) tile[j][q + u * (PSFM_BLOCK / TILE_J)] = r[u];
        __syncthreads();
        if (k0 + TILE_K < maxlen) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int t = k0 + TILE_K + q + u * (PSFM_BLOCK / TILE_J);
                r[u] = t < lj ? log[(int64_t)(bj + t) * cap + col] : make_double2(0.0, 0.0);
            }
        }
        // write: for a fixed track, TILE_K consecutive times -> contiguous run of the result
        const int kk = tid & (TILE_K - 1);
        for (int jj = tid / TILE_K; jj < TILE_J; jj += PSFM_BLOCK / TILE_K) {
            const int t = k0 + kk;
            if (t < s_len[jj]) out[s_off[jj] + t] = tile[jj][kk];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_gather_kernel(const double2* __restrict__ log, int64_t cap,
                                                                 const int* __restrict__ lanes,
                                                                 const int* __restrict__ birth,
                                                                 const int* __restrict__ len,
                                                                 const int64_t* __restrict__ off, int64_t n,
                                                                 double2* __restrict__ out, PsfmPlan pl)
{
    psfm_gather_body(log, cap, lanes, birth, len, off, n, out, pl);
}

// The persistent loop logs the sampled flow of every survived step instead of positions (half the bytes, written once,
// read once): a trajectory is its birth grid point plus the running f64 sum of its column -- the additions the loop
// itself performed, p(t+1) = p(t) + (double)flow, re-run here in the same order (one thread walks one trajectory's
// chunk serially through LDS; a parallel scan would round differently whenever an addition is inexact).
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_gather_delta_kernel(const float2* __restrict__ dlog, int64_t cap,
                                                                       const int* __restrict__ lanes,
                                                                       const int* __restrict__ birth,
                                                                       const int* __restrict__ len,
                                                                       const int64_t* __restrict__ off, int64_t n,
                                                                       const void* __restrict__ keys, int use32, int shift_b,
                                                                       int GW, int ratio, double2* __restrict__ out, PsfmPlan pl)
{
    __shared__ float2 dtile[TILE_J][TILE_K + 1];
    __shared__ double2 tile[TILE_J][TILE_K + 1];
    __shared__ double2 s_pos[TILE_J];
    __shared__ int s_lane[TILE_J], s_birth[TILE_J], s_len[TILE_J];
    __shared__ int64_t s_off[TILE_J];
    __shared__ int s_maxlen;
    const int tid = threadIdx.x;
    const int64_t id0 = (int64_t)blockIdx.x * TILE_J;
    int hb = 0, hl = 0;
    if (tid == 0) s_maxlen = 0;
    __syncthreads();
    if (tid < TILE_J) {
        const int64_t id = id0 + tid;
        const bool ok = id < n;
        int b = 0, l = 0, idx = -1;
        const int my_lane = ok ? lanes[id] : 0;
        if (ok) psfm_track_header(pl, birth, len, id, &b, &l, &idx);
        hb = b; hl = l;
        s_lane[tid] = my_lane;
        s_birth[tid] = b;
        s_len[tid] = l;
        if (ok) {
            atomicMax(&s_maxlen, l);
            if (idx < 0) {
                const unsigned long long mask = (1ull << shift_b) - 1ull;
                idx = (int)((use32 ? (unsigned long long)((const unsigned*)keys)[id] : ((const unsigned long long*)keys)[id]) & mask);
            }
            const int gy = idx / GW, gx = idx - gy * GW;
            s_pos[tid] = make_double2((double)(gx * ratio), (double)(gy * ratio));   // trajectory.py:110-115
        }
    }
    __syncthreads();
    const int maxlen = s_maxlen;
    const int j = tid & (TILE_J - 1);
    const int q = tid / TILE_J;
    constexpr int NLD = TILE_K * TILE_J / PSFM_BLOCK;   // tile entries a thread loads per chunk
    const int lj = s_len[j];
    const int64_t col = s_lane[j];
    const int bj = s_birth[j];
    // point t >= 1 of a trajectory born at b needs the flow of step b + t - 1 (slab b + t - 1, its column)
    float2 r[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int t = q + u * (PSFM_BLOCK / TILE_J);
        r[u] = (t >= 1 && t < lj) ? dlog[(int64_t)(bj + t - 1) * cap + col] : make_float2(0.f, 0.f);
    }
    if (tid < TILE_J) s_off[tid] = id0 + tid < n ? psfm_track_offset(pl, off, id0 + tid, hb, hl) : 0;   // (seen behind the loop's first barrier)
    for (int k0 = 0; k0 < maxlen; k0 += TILE_K) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) dtile[j][q + u * (PSFM_BLOCK / TILE_J)] = r[u];
        __syncthreads();
        // the next chunk's flows travel while this one is accumulated and written
        if (k0 + TILE_K < maxlen) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int t = k0 + TILE_K + q + u * (PSFM_BLOCK / TILE_J);
                r[u] = (t < lj) ? dlog[(int64_t)(bj + t - 1) * cap + col] : make_float2(0.f, 0.f);
            }
        }
        // accumulate: one thread per trajectory, in time order (entries outside [1, len) are zero-filled and skipped)
        if (tid < TILE_J) {
            double2 pos = s_pos[tid];
            const int l = s_len[tid];
            float2 f[TILE_K];
#pragma unroll
            for (int k = 0; k < TILE_K; ++k) f[k] = dtile[tid][k];
#pragma unroll
            for (int k = 0; k < TILE_K; ++k) {
                const int t = k0 + k;
                if (t >= 1 && t < l) { pos.x = pos.x + (double)f[k].x; pos.y = pos.y + (double)f[k].y; }
                tile[tid][k] = pos;
            }
            s_pos[tid] = pos;
        }
        __syncthreads();
        // write: for a fixed track, TILE_K consecutive times -> contiguous run of the result
        const int kk = tid & (TILE_K - 1);
        for (int jj = tid / TILE_K; jj < TILE_J; jj += PSFM_BLOCK / TILE_K) {
            const int t = k0 + kk;
            if (t < s_len[jj]) out[s_off[jj] + t] = tile[jj][kk];
        }
        __syncthreads();
    }
}

// key format of a sequence: 32-bit triangular keys when they fit
static PsfmKeyFmt psfm_key_fmt(const PsfmTrackDims& d, unsigned* end_bit)
{
    PsfmKeyFmt f;
    f.shift_b = d.shift_b; f.shift_d = d.shift_d;
    const long long lmax = (long long)d.n_flows + 1;            // last valid time <= n_flows
    const long long tri_max = lmax * (lmax + 1) / 2 + lmax;
    int tri_bits = 1;
    while ((1ll << tri_bits) <= tri_max) ++tri_bits;
    f.use32 = (tri_bits + d.shift_b <= 32) ? 1 : 0;
    int tbits = 1;
    while ((1ll << tbits) < (long long)d.n_flows + 2) ++tbits;
    *end_bit = f.use32 ? (unsigned)(tri_bits + d.shift_b) : (unsigned)(d.shift_d + tbits);
    return f;
}

// Which half of sort_keys / sort_lanes the compaction fills: the second (rocPRIM sorts second half -> first half), unless this file's
// own sort runs an even number of passes -- it ping-pongs between the halves and has to END in the first.
static bool psfm_own_sort(const PsfmKeyFmt& fmt, int64_t n)
{
    static const int on = getenv("PSFM_FIN_SORT") ? atoi(getenv("PSFM_FIN_SORT")) : 1;      // 0: rocPRIM's device sort (A/B, tests)
    return on && fmt.use32 && n < (1ll << 31);
}
static bool psfm_records_in_first_half(const PsfmTrackDims& d, int64_t n)
{
    unsigned end_bit = 0;
    const PsfmKeyFmt fmt = psfm_key_fmt(d, &end_bit);
    return psfm_own_sort(fmt, n) && (psfm_sort_pairs32_passes(end_bit) % 2 == 0);
}

// common tail: (key, lane) records already compacted into the halves of sort_keys / sort_lanes psfm_records_in_first_half() names
// (keys in the format psfm_key_fmt() chose: 8-byte or 4-byte entries, a half = n entries of that width)
static psfm_status psfm_finalize_sorted(psfm_ctx* c, const PsfmTrackDims& d, int64_t n, int64_t npts, bool delta_log, hipStream_t s)
{
    psfm_status st;
    unsigned end_bit = 0;
    const PsfmKeyFmt fmt = psfm_key_fmt(d, &end_bit);
    int* l_in = c->sort_lanes.as<int>() + n;
    int* l_out = c->sort_lanes.as<int>();
    size_t tmp_bytes = 0, scan_bytes = 0;
    PSFM_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                     (size_t)(n + 1), rocprim::plus<int64_t>(), s));
    if (psfm_own_sort(fmt, n)) {
        if ((st = c->sort_tmp.ensure(scan_bytes)) != PSFM_OK) return st;      // (before the sort's launches: growing the buffer frees it)
        if ((st = psfm_sort_pairs32(c, c->sort_keys.as<unsigned>(), l_out, c->sort_keys.as<unsigned>() + n, l_in, n, end_bit, s)) != PSFM_OK) return st;
    } else if (fmt.use32) {
        unsigned* k_in = c->sort_keys.as<unsigned>() + n;
        unsigned* k_out = c->sort_keys.as<unsigned>();
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)n, 0u, end_bit, s));
        if ((st = c->sort_tmp.ensure(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)n, 0u, end_bit, s));
    } else {
        unsigned long long* k_in = c->sort_keys.as<unsigned long long>() + n;
        unsigned long long* k_out = c->sort_keys.as<unsigned long long>();
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)n, 0u, end_bit, s));
        if ((st = c->sort_tmp.ensure(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)n, 0u, end_bit, s));
    }
    // birth / length / offset of every trajectory: from the group plan (few groups), else decode + scan over the trajectories
    if ((st = c->res_birth.ensure(sizeof(int) * n)) != PSFM_OK) return st;
    if ((st = c->res_len.ensure(sizeof(int) * n)) != PSFM_OK) return st;
    if ((st = c->res_off.ensure(sizeof(int64_t) * (n + 1))) != PSFM_OK) return st;
    static const int plan_on = getenv("PSFM_FIN_PLAN") ? atoi(getenv("PSFM_FIN_PLAN")) : 1;      // 0: decode + scan always (A/B, tests)
    const long long lmax = (long long)d.n_flows + 1;
    const long long ng = lmax * (lmax + 1) / 2 + lmax + 1;      // group numbers last (last + 1) / 2 + birth, birth <= last <= lmax
    PsfmPlan pl;
    pl.gstart = nullptr; pl.goff = nullptr; pl.keys = (const void*)c->sort_keys.p; pl.use32 = fmt.use32; pl.shift_b = d.shift_b; pl.shift_d = d.shift_d;
    pl.birth_w = c->res_birth.as<int>(); pl.len_w = c->res_len.as<int>(); pl.off_w = c->res_off.as<int64_t>();
    if (plan_on && ng <= PSFM_PLAN_MAX_GROUPS && n < (1ll << 31)) {
        if ((st = c->scan_tmp.ensure((sizeof(int64_t) + sizeof(int)) * (size_t)ng)) != PSFM_OK) return st;
        int64_t* goff = c->scan_tmp.as<int64_t>();
        int* gstart = (int*)(goff + ng);
        // the marks: -1 everywhere between two calls (the plan kernel puts back what the bounds kernel marked); filled here only
        // the first time, after a call that did not get as far, or when the buffer is new
        const void* had = c->fin_marks.p;
        if ((st = c->fin_marks.ensure(sizeof(int) * (size_t)PSFM_PLAN_MAX_GROUPS)) != PSFM_OK) return st;
        if (!c->fin_marks_clean || had != c->fin_marks.p)
            PSFM_HIP(hipMemsetAsync(c->fin_marks.p, 0xff, sizeof(int) * (size_t)PSFM_PLAN_MAX_GROUPS, s));
        c->fin_marks_clean = false;
        hipLaunchKernelGGL(psfm_group_bounds_kernel, dim3((unsigned)((n + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK), 0, s,
                           (const void*)c->sort_keys.p, n, fmt.use32, d.shift_b, d.shift_d, c->fin_marks.as<int>());
        hipLaunchKernelGGL(psfm_group_plan_kernel, dim3(1), dim3(PSFM_PLAN_BLOCK), 0, s, c->fin_marks.as<int>(), gstart, goff, (int)ng, n,
                           c->res_off.as<int64_t>() + n);
        PSFM_HIP(hipGetLastError());
        c->fin_marks_clean = true;
        pl.gstart = gstart; pl.goff = goff;
    } else {
        if ((st = c->scan_tmp.ensure(sizeof(int64_t) * (n + 1))) != PSFM_OK) return st;
        if (fmt.use32)
            hipLaunchKernelGGL(psfm_decode32_kernel, dim3((unsigned)((n + 1 + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK), 0, s,
                               c->sort_keys.as<unsigned>(), n, d.shift_b, c->res_birth.as<int>(), c->res_len.as<int>(),
                               c->scan_tmp.as<int64_t>());
        else
            hipLaunchKernelGGL(psfm_decode_kernel, dim3((unsigned)((n + 1 + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK), 0, s,
                               c->sort_keys.as<unsigned long long>(), n, d.shift_b, d.shift_d, c->res_birth.as<int>(),
                               c->res_len.as<int>(), c->scan_tmp.as<int64_t>());
        PSFM_HIP(hipGetLastError());
        PSFM_HIP(rocprim::exclusive_scan(c->sort_tmp.p, scan_bytes, c->scan_tmp.as<int64_t>(), c->res_off.as<int64_t>(),
                                         (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), s));
    }
    c->res_n_points = npts;
    // transpose the frame-major log into the id-ordered CSR
    if ((st = c->res_xy.ensure(sizeof(double2) * (size_t)(npts > 0 ? npts : 1))) != PSFM_OK) return st;
    if (delta_log)
        hipLaunchKernelGGL(psfm_gather_delta_kernel, dim3((unsigned)((n + TILE_J - 1) / TILE_J)), dim3(PSFM_BLOCK), 0, s,
                           c->log.as<float2>(), d.cap, c->sort_lanes.as<int>(), c->res_birth.as<int>(),
                           c->res_len.as<int>(), c->res_off.as<int64_t>(), n, (const void*)c->sort_keys.p, fmt.use32,
                           d.shift_b, d.GW, d.ratio, c->res_xy.as<double2>(), pl);
    else
        hipLaunchKernelGGL(psfm_gather_kernel, dim3((unsigned)((n + TILE_J - 1) / TILE_J)), dim3(PSFM_BLOCK), 0, s,
                           c->log.as<double2>(), d.cap, c->sort_lanes.as<int>(), c->res_birth.as<int>(),
                           c->res_len.as<int>(), c->res_off.as<int64_t>(), n, c->res_xy.as<double2>(), pl);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

psfm_status psfm_finalize(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s)
{
    PsfmCounters* ctr = c->counters.as<PsfmCounters>();
    PsfmShard* shards = c->shards.as<PsfmShard>();
    // 1. survivors -> records with last_valid_time = n_flows
    hipLaunchKernelGGL(psfm_collect_alive_kernel, dim3((unsigned)((d.cap + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                       0, s, c->birth_frame.as<int>(), c->birth_idx.as<int>(), ctr, shards,
                       c->fin_keys.as<unsigned long long>(), c->fin_lanes.as<int>(), (int)d.cap, d.shard_cap,
                       d.n_flows, d.shift_b, d.shift_d);
    PSFM_HIP(hipGetLastError());
    PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
    PsfmShard* hs = (PsfmShard*)((char*)c->host_pinned + 512);
    PSFM_HIP(hipMemcpyAsync(hc, ctr, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipMemcpyAsync(hs, shards, sizeof(PsfmShard) * PSFM_NSHARD, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    PsfmShardOffsets so;
    int64_t n = 0, npts = 0;
    bool over = hc->overflow != 0;
    for (int k = 0; k < PSFM_NSHARD; ++k) npts += hs[k].points;   // every log write was counted by its block
    for (int k = 0; k < PSFM_NSHARD; ++k) {
        if (hs[k].fin_cnt > d.shard_cap) over = true;
        so.count[k] = hs[k].fin_cnt > d.shard_cap ? d.shard_cap : hs[k].fin_cnt;
        so.start[k] = n;
        n += so.count[k];
    }
    if (over || hc->n_lanes > d.cap) {
        psfm_set_error("capacity exceeded: lanes used %d of %lld, trajectory records %lld of %lld (%d shards); "
                       "raise psfm_ctx_set_capacity", hc->n_lanes, (long long)d.cap, (long long)n,
                       (long long)d.traj_cap, PSFM_NSHARD);
        return PSFM_ERR_CAPACITY;
    }
    c->res_n_traj = n;
    c->res_n_points = 0;
    if (n == 0) return PSFM_OK;
    // 2. compact the shards, then sort (key, lane) by key over the used bits
    psfm_status st;
    if ((st = c->sort_keys.ensure(sizeof(unsigned long long) * n * 2)) != PSFM_OK) return st;
    if ((st = c->sort_lanes.ensure(sizeof(int) * n * 2)) != PSFM_OK) return st;
    unsigned end_bit_unused = 0;
    const PsfmKeyFmt fmt = psfm_key_fmt(d, &end_bit_unused);
    const int64_t half = psfm_records_in_first_half(d, n) ? 0 : n;
    unsigned long long* kdst = fmt.use32 ? (unsigned long long*)(c->sort_keys.as<unsigned>() + half) : c->sort_keys.as<unsigned long long>() + half;
    hipLaunchKernelGGL(psfm_compact_shards_kernel, dim3(64, PSFM_NSHARD), dim3(PSFM_BLOCK), 0, s,
                       c->fin_keys.as<unsigned long long>(), c->fin_lanes.as<int>(), d.shard_cap, so,
                       kdst, c->sort_lanes.as<int>() + half, fmt);
    PSFM_HIP(hipGetLastError());
    return psfm_finalize_sorted(c, d, n, npts, false, s);
}

// ---- persistent frame loop: records sit in one private segment per block (+ a shared tail) ----
// Block b of the copy = segment b of the loop (b == nblk: the shared tail); where it goes in the sort's input is the sum of the counts
// in front of it, which every block adds up for itself from seg_info (a few KB out of the L2s) -- no table from the host.
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_compact_segments_kernel(
    const unsigned long long* __restrict__ fin_keys, const int* __restrict__ fin_lanes, const int2* __restrict__ seg_info, int nblk,
    int seg_cap, const PsfmCounters* __restrict__ ctr, int spill_cap, unsigned long long* __restrict__ keys, int* __restrict__ lanes,
    PsfmKeyFmt fmt)
{
    __shared__ long long s_part[PSFM_BLOCK / PSFM_WAVE];
    const int b = blockIdx.x, tid = threadIdx.x;
    long long mine = 0;
    for (int k = tid; k < b; k += PSFM_BLOCK) mine += seg_info[k].x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((tid & (PSFM_WAVE - 1)) == 0) s_part[tid / PSFM_WAVE] = mine;
    __syncthreads();
    long long dst = 0;
    for (int w = 0; w < PSFM_BLOCK / PSFM_WAVE; ++w) dst += s_part[w];
    const int count = b < nblk ? seg_info[b].x : min(ctr->spill_cnt, spill_cap);
    const long long src = (long long)b * seg_cap;
    // four records per thread in flight (a segment holds ~2 000 at the headline shape: two rounds instead of eight)
    for (int i0 = 0; i0 < count; i0 += 4 * PSFM_BLOCK) {
        unsigned long long k[4];
        int l[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PSFM_BLOCK + tid;
            k[u] = i < count ? fin_keys[src + i] : 0ull;
            l[u] = i < count ? fin_lanes[src + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PSFM_BLOCK + tid;
            if (i < count) {
                psfm_put_key(keys, dst + i, k[u], fmt);
                lanes[dst + i] = l[u];
            }
        }
    }
}

psfm_status psfm_finalize_persist(psfm_ctx* c, const PsfmTrackDims& d, bool* fallback, hipStream_t s)
{
    *fallback = false;
    const int nseg = d.nblk + 1;
    const size_t need = sizeof(int2) * (size_t)d.nblk;
    if (c->host_seg_bytes < need) {
        if (c->host_seg) (void)hipHostFree(c->host_seg);
        c->host_seg = nullptr; c->host_seg_bytes = 0;
        PSFM_HIP(hipHostMalloc(&c->host_seg, need, hipHostMallocDefault));
        c->host_seg_bytes = need;
    }
    int2* hinfo = (int2*)c->host_seg;
    PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
    PSFM_HIP(hipMemcpyAsync(hc, c->counters.p, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipMemcpyAsync(hinfo, c->seg_info.p, sizeof(int2) * (size_t)d.nblk, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    if (hc->pad[0]) fprintf(stderr, "psfm persist check: site %d index %d\n", hc->pad[0], hc->pad[1]);
    if (hc->overflow & (4 | 8)) {   // more tracks than resident lanes / a barrier gave up: per-frame launches take over
        *fallback = true;
        return PSFM_OK;
    }
    int64_t n = 0, npts = 0;
    for (int b = 0; b < d.nblk; ++b) {
        n += hinfo[b].x;
        npts += hinfo[b].y;
    }
    const int spilled = hc->spill_cnt < d.spill_cap ? hc->spill_cnt : d.spill_cap;
    n += spilled;
    if (hc->overflow != 0) {
        psfm_set_error("capacity exceeded (persistent loop): overflow bits %d, trajectory records %lld (%d spilled of %d); "
                       "raise psfm_ctx_set_capacity", hc->overflow, (long long)n, hc->spill_cnt, d.spill_cap);
        return PSFM_ERR_CAPACITY;
    }
    c->res_n_traj = n;
    c->res_n_points = 0;
    if (n == 0) return PSFM_OK;
    psfm_status st;
    if ((st = c->sort_keys.ensure(sizeof(unsigned long long) * n * 2)) != PSFM_OK) return st;
    if ((st = c->sort_lanes.ensure(sizeof(int) * n * 2)) != PSFM_OK) return st;
    unsigned end_bit_unused = 0;
    const PsfmKeyFmt fmt = psfm_key_fmt(d, &end_bit_unused);
    const int64_t half = psfm_records_in_first_half(d, n) ? 0 : n;
    unsigned long long* kdst = fmt.use32 ? (unsigned long long*)(c->sort_keys.as<unsigned>() + half) : c->sort_keys.as<unsigned long long>() + half;
    hipLaunchKernelGGL(psfm_compact_segments_kernel, dim3((unsigned)nseg), dim3(PSFM_BLOCK), 0, s,
                       c->fin_keys.as<unsigned long long>(), c->fin_lanes.as<int>(), (const int2*)c->seg_info.as<int2>(), d.nblk, d.seg_cap,
                       (const PsfmCounters*)c->counters.as<PsfmCounters>(), d.spill_cap, kdst, c->sort_lanes.as<int>() + half, fmt);
    PSFM_HIP(hipGetLastError());
    return psfm_finalize_sorted(c, d, n, npts, true, s);   // the persistent loop logs flows, not positions
}

// ------------------------------------------------------------------------------------------------
// Segmented finalize of a BATCH (psfm_connect_batch): what psfm_finalize does for one sequence, for B same-shape sequences with ONE
// host synchronisation, ONE radix sort and ONE scan.  A frame of a small sequence leaves a few ten thousand records: per
// sequence the finalize is a dozen launch-latency-bound kernels and a host round trip -- as much time as its whole frame loop
// takes inside a batch.  Here the records of all sequences go into one array under the key (sequence : key), are sorted once, and
// the decode / offset / gather kernels scatter into every context's own result buffers (so psfm_result_* and the consumers work
// on a batch member as on any other context).
// ------------------------------------------------------------------------------------------------
struct PsfmFinSeq {
    // the sequence's tables (known before the synchronisation)
    const int* birth_frame; const int* birth_idx; PsfmCounters* ctr; PsfmShard* shards;
    unsigned long long* fin_keys; int* fin_lanes;
    const double2* log; int64_t cap;
    int shard_cap, last_time;
    // behind the synchronisation: where its records sit in the combined arrays, where its result goes
    int64_t rec_start, n;
    int* res_birth; int* res_len; int64_t* res_off; double2* res_xy;
};

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_collect_alive_batch_kernel(const PsfmFinSeq* __restrict__ T, int shift_b, int shift_d)
{
    const PsfmFinSeq& q = T[blockIdx.y];
    psfm_collect_alive_body(q.birth_frame, q.birth_idx, q.ctr, q.shards, q.fin_keys, q.fin_lanes, (int)q.cap, q.shard_cap, q.last_time,
                            shift_b, shift_d);
}

// per sequence: {records, trajectory points, overflow flags, lanes used} -> counts[seq * 4 ..] (one block of one wave per sequence)
__global__ __launch_bounds__(PSFM_WAVE) void psfm_batch_counts_kernel(const PsfmFinSeq* __restrict__ T, long long* __restrict__ counts)
{
    const PsfmFinSeq& q = T[blockIdx.x];
    const int k = threadIdx.x;
    static_assert(PSFM_NSHARD == PSFM_WAVE, "one lane per shard");
    const int fc = q.shards[k].fin_cnt;
    long long n = fc > q.shard_cap ? q.shard_cap : fc;
    long long pts = q.shards[k].points;
    int over = fc > q.shard_cap ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o); pts += __shfl_down(pts, o); over |= __shfl_down(over, o); }
    if (k == 0) {
        counts[blockIdx.x * 4 + 0] = n;
        counts[blockIdx.x * 4 + 1] = pts;
        counts[blockIdx.x * 4 + 2] = (long long)(q.ctr->overflow | (over ? 2 : 0));
        counts[blockIdx.x * 4 + 3] = (long long)q.ctr->n_lanes;
    }
}

// combined key = (sequence << seq_shift) | key of the sequence (32- or 64-bit format, psfm_key_fmt)
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_compact_shards_batch_kernel(const PsfmFinSeq* __restrict__ T, unsigned long long* __restrict__ keys,
                                                                               int* __restrict__ lanes, PsfmKeyFmt fmt, int seq_shift)
{
    __shared__ int s_cnt[PSFM_NSHARD];
    const int shard = blockIdx.y, seq = blockIdx.z;
    const PsfmFinSeq& q = T[seq];
    if (threadIdx.x < PSFM_NSHARD) { const int fc = q.shards[threadIdx.x].fin_cnt; s_cnt[threadIdx.x] = fc > q.shard_cap ? q.shard_cap : fc; }
    __syncthreads();
    int64_t start = q.rec_start;
    for (int k = 0; k < shard; ++k) start += s_cnt[k];
    const int n = s_cnt[shard];
    for (int i = blockIdx.x * PSFM_BLOCK + threadIdx.x; i < n; i += gridDim.x * PSFM_BLOCK) {
        const unsigned long long k = q.fin_keys[(int64_t)shard * q.shard_cap + i];
        if (fmt.use32) ((unsigned*)keys)[start + i] = psfm_key32(k, fmt) | (seq_shift < 32 ? (unsigned)seq << seq_shift : 0u);
        else keys[start + i] = k | ((unsigned long long)seq << seq_shift);
        lanes[start + i] = q.fin_lanes[(int64_t)shard * q.shard_cap + i];
    }
}

// sorted combined record i -> (sequence, local id): birth / length into the sequence's result, length into the combined scan input
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_decode_batch_kernel(const PsfmFinSeq* __restrict__ T, const void* __restrict__ keys, int64_t n,
                                                                       PsfmKeyFmt fmt, int seq_shift, int64_t* __restrict__ len64)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i > n) return;
    if (i == n) { len64[i] = 0; return; }
    int seq, b, last;
    if (fmt.use32) {
        const unsigned k = ((const unsigned*)keys)[i];
        seq = seq_shift < 32 ? (int)(k >> seq_shift) : 0;       // (a batch of one sequence whose key fills the word)
        const unsigned tri = (seq_shift < 32 ? k & ((1u << seq_shift) - 1u) : k) >> fmt.shift_b;
        int l = (int)((sqrtf(8.0f * (float)tri + 1.0f) - 1.0f) * 0.5f);      // (as psfm_decode32_kernel)
        while (l > 0 && (unsigned)l * (unsigned)(l + 1) / 2u > tri) --l;
        while ((unsigned)(l + 1) * (unsigned)(l + 2) / 2u <= tri) ++l;
        last = l;
        b = (int)(tri - (unsigned)l * (unsigned)(l + 1) / 2u);
    } else {
        const unsigned long long kk = ((const unsigned long long*)keys)[i];
        seq = (int)(kk >> seq_shift);
        const unsigned long long k = kk & ((1ull << seq_shift) - 1ull);
        last = (int)(k >> fmt.shift_d);
        b = (int)((k >> fmt.shift_b) & ((1ull << (fmt.shift_d - fmt.shift_b)) - 1ull));
    }
    const PsfmFinSeq& q = T[seq];
    const int64_t local = i - q.rec_start;
    q.res_birth[local] = b;
    q.res_len[local] = last - b + 1;
    len64[i] = (int64_t)(last - b + 1);
}

// the sequence's offsets = the combined scan minus its value at the sequence's first record
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_offsets_batch_kernel(const PsfmFinSeq* __restrict__ T, const int64_t* __restrict__ scan)
{
    const PsfmFinSeq& q = T[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i > q.n) return;
    q.res_off[i] = scan[q.rec_start + i] - scan[q.rec_start];
}

__global__ __launch_bounds__(PSFM_BLOCK) void psfm_gather_batch_kernel(const PsfmFinSeq* __restrict__ T, const int* __restrict__ lanes)
{
    const PsfmFinSeq& q = T[blockIdx.y];
    if ((int64_t)blockIdx.x * TILE_J >= q.n) return;
    PsfmPlan none; none.gstart = nullptr;
    psfm_gather_body(q.log, q.cap, lanes + q.rec_start, q.res_birth, q.res_len, q.res_off, q.n, q.res_xy, none);
}

psfm_status psfm_finalize_batch(psfm_ctx* own, psfm_ctx* const* ctxs, const PsfmTrackDims* dims, int n_seq, hipStream_t s)
{
    if (n_seq < 1 || n_seq > PSFM_BATCH_MAX) { psfm_set_error("psfm_finalize_batch: bad batch size %d", n_seq); return PSFM_ERR_ARG; }
    psfm_status st;
    // the table: device copy in the owner's batch workspace behind the frame loop's table, host copy in its pinned staging
    const size_t tbytes = sizeof(PsfmFinSeq) * (size_t)n_seq, cbytes = sizeof(long long) * 4 * (size_t)n_seq;
    const size_t need_host = tbytes + cbytes;
    if (own->host_seg_bytes < need_host) {
        if (own->host_seg) (void)hipHostFree(own->host_seg);
        own->host_seg = nullptr; own->host_seg_bytes = 0;
        PSFM_HIP(hipHostMalloc(&own->host_seg, need_host + 4096, hipHostMallocDefault));
        own->host_seg_bytes = need_host + 4096;
    }
    if ((st = own->seg_table.ensure(tbytes + cbytes)) != PSFM_OK) return st;
    PsfmFinSeq* hT = (PsfmFinSeq*)own->host_seg;
    long long* hcnt = (long long*)((char*)own->host_seg + tbytes);
    PsfmFinSeq* dT = own->seg_table.as<PsfmFinSeq>();
    long long* dcnt = (long long*)((char*)own->seg_table.p + tbytes);
    int64_t cap_max = 0;
    PsfmTrackDims dk = dims[0];       // the batch's key format: common shifts, the longest sequence's time range
    for (int i = 0; i < n_seq; ++i) {
        const PsfmTrackDims& d = dims[i];
        psfm_ctx* c = ctxs[i];
        if (d.shift_b != dk.shift_b || d.shift_d != dk.shift_d) { psfm_set_error("psfm_finalize_batch: sequences with different key formats"); return PSFM_ERR_ARG; }
        if (d.n_flows > dk.n_flows) dk.n_flows = d.n_flows;
        memset(&hT[i], 0, sizeof(PsfmFinSeq));
        hT[i].birth_frame = c->birth_frame.as<int>(); hT[i].birth_idx = c->birth_idx.as<int>();
        hT[i].ctr = c->counters.as<PsfmCounters>(); hT[i].shards = c->shards.as<PsfmShard>();
        hT[i].fin_keys = c->fin_keys.as<unsigned long long>(); hT[i].fin_lanes = c->fin_lanes.as<int>();
        hT[i].log = c->log.as<double2>(); hT[i].cap = d.cap;
        hT[i].shard_cap = d.shard_cap; hT[i].last_time = d.n_flows;
        if (d.cap > cap_max) cap_max = d.cap;
    }
    PSFM_HIP(hipMemcpyAsync(dT, hT, tbytes, hipMemcpyHostToDevice, s));
    // 1. survivors -> records; 2. the counts of every sequence in one copy, ONE synchronisation
    hipLaunchKernelGGL(psfm_collect_alive_batch_kernel, dim3((unsigned)((cap_max + PSFM_BLOCK - 1) / PSFM_BLOCK), (unsigned)n_seq), dim3(PSFM_BLOCK),
                       0, s, dT, dk.shift_b, dk.shift_d);
    hipLaunchKernelGGL(psfm_batch_counts_kernel, dim3((unsigned)n_seq), dim3(PSFM_WAVE), 0, s, dT, dcnt);
    PSFM_HIP(hipGetLastError());
    PSFM_HIP(hipMemcpyAsync(hcnt, dcnt, cbytes, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    int64_t N = 0, n_max = 0;
    for (int i = 0; i < n_seq; ++i) {
        psfm_ctx* c = ctxs[i];
        const int64_t n = hcnt[4 * i + 0], npts = hcnt[4 * i + 1];
        PsfmCounters* hc = (PsfmCounters*)c->host_pinned;      // (psfm_track_info reads the lane count from here)
        hc->n_lanes = (int)hcnt[4 * i + 3]; hc->overflow = (int)hcnt[4 * i + 2];
        if (hcnt[4 * i + 2] != 0 || hcnt[4 * i + 3] > dims[i].cap) {
            psfm_set_error("capacity exceeded (sequence %d of the batch): lanes used %lld of %lld, trajectory records %lld of %lld, overflow bits %lld; "
                           "raise psfm_ctx_set_capacity", i, hcnt[4 * i + 3], (long long)dims[i].cap, (long long)n, (long long)dims[i].traj_cap,
                           hcnt[4 * i + 2]);
            return PSFM_ERR_CAPACITY;
        }
        c->res_n_traj = n;
        c->res_n_points = npts;
        hT[i].rec_start = N; hT[i].n = n;
        N += n;
        if (n > n_max) n_max = n;
        if ((st = c->res_birth.ensure(sizeof(int) * (size_t)(n > 0 ? n : 1))) != PSFM_OK) return st;
        if ((st = c->res_len.ensure(sizeof(int) * (size_t)(n > 0 ? n : 1))) != PSFM_OK) return st;
        if ((st = c->res_off.ensure(sizeof(int64_t) * (size_t)(n + 1))) != PSFM_OK) return st;
        if ((st = c->res_xy.ensure(sizeof(double2) * (size_t)(npts > 0 ? npts : 1))) != PSFM_OK) return st;
        hT[i].res_birth = c->res_birth.as<int>(); hT[i].res_len = c->res_len.as<int>();
        hT[i].res_off = c->res_off.as<int64_t>(); hT[i].res_xy = c->res_xy.as<double2>();
    }
    PSFM_HIP(hipMemcpyAsync(dT, hT, tbytes, hipMemcpyHostToDevice, s));
    if (N == 0) {
        for (int i = 0; i < n_seq; ++i) PSFM_HIP(hipMemsetAsync(ctxs[i]->res_off.p, 0, sizeof(int64_t), s));
        return PSFM_OK;
    }
    // 3. the combined (sequence : key, lane) array, sorted once
    unsigned end_bit = 0;
    PsfmKeyFmt fmt = psfm_key_fmt(dk, &end_bit);
    int seq_bits = 0;
    while ((1 << seq_bits) < n_seq) ++seq_bits;
    if (fmt.use32 && (int)end_bit + seq_bits > 32) {        // the sequence number does not fit beside a 32-bit key: 64-bit keys
        fmt.use32 = 0;
        int tbits = 1;
        while ((1ll << tbits) < (long long)dk.n_flows + 2) ++tbits;
        end_bit = (unsigned)(dk.shift_d + tbits);
    }
    if (!fmt.use32 && (int)end_bit + seq_bits > 64) { psfm_set_error("psfm_finalize_batch: key does not fit 64 bits"); return PSFM_ERR_ARG; }
    const int seq_shift = (int)end_bit;
    const unsigned sort_end = end_bit + (unsigned)seq_bits;
    if ((st = own->sort_keys.ensure(sizeof(unsigned long long) * (size_t)N * 2)) != PSFM_OK) return st;
    if ((st = own->sort_lanes.ensure(sizeof(int) * (size_t)N * 2)) != PSFM_OK) return st;
    if ((st = own->scan_tmp.ensure(sizeof(int64_t) * (size_t)(N + 1) * 2)) != PSFM_OK) return st;
    // (this file's sort ends in the first half whatever its pass count: with an even count the records start there)
    const bool own_sort = psfm_own_sort(fmt, N);
    const int64_t half = (own_sort && psfm_sort_pairs32_passes(sort_end) % 2 == 0) ? 0 : N;
    unsigned long long* kdst = fmt.use32 ? (unsigned long long*)(own->sort_keys.as<unsigned>() + half) : own->sort_keys.as<unsigned long long>() + half;
    hipLaunchKernelGGL(psfm_compact_shards_batch_kernel, dim3(8, PSFM_NSHARD, (unsigned)n_seq), dim3(PSFM_BLOCK), 0, s, dT, kdst,
                       own->sort_lanes.as<int>() + half, fmt, seq_shift);
    PSFM_HIP(hipGetLastError());
    int* l_in = own->sort_lanes.as<int>() + N;
    int* l_out = own->sort_lanes.as<int>();
    size_t tmp_bytes = 0, scan_bytes = 0;
    PSFM_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(N + 1),
                                     rocprim::plus<int64_t>(), s));
    if (own_sort) {
        if ((st = own->sort_tmp.ensure(scan_bytes)) != PSFM_OK) return st;      // (before the sort's launches: growing the buffer frees it)
        if ((st = psfm_sort_pairs32(own, own->sort_keys.as<unsigned>(), l_out, own->sort_keys.as<unsigned>() + N, l_in, N, sort_end, s)) != PSFM_OK) return st;
    } else if (fmt.use32) {
        unsigned* k_in = own->sort_keys.as<unsigned>() + N;
        unsigned* k_out = own->sort_keys.as<unsigned>();
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)N, 0u, sort_end, s));
        if ((st = own->sort_tmp.ensure(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(own->sort_tmp.p, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)N, 0u, sort_end, s));
    } else {
        unsigned long long* k_in = own->sort_keys.as<unsigned long long>() + N;
        unsigned long long* k_out = own->sort_keys.as<unsigned long long>();
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)N, 0u, sort_end, s));
        if ((st = own->sort_tmp.ensure(tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(own->sort_tmp.p, tmp_bytes, k_in, k_out, l_in, l_out, (size_t)N, 0u, sort_end, s));
    }
    // 4. decode -> per-sequence birth / length, combined lengths -> ONE scan -> per-sequence offsets; 5. the transpose gather
    int64_t* len64 = own->scan_tmp.as<int64_t>();
    int64_t* scan = len64 + (N + 1);
    hipLaunchKernelGGL(psfm_decode_batch_kernel, dim3((unsigned)((N + 1 + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK), 0, s, dT,
                       (const void*)own->sort_keys.p, N, fmt, seq_shift, len64);
    PSFM_HIP(hipGetLastError());
    PSFM_HIP(rocprim::exclusive_scan(own->sort_tmp.p, scan_bytes, len64, scan, (int64_t)0, (size_t)(N + 1), rocprim::plus<int64_t>(), s));
    hipLaunchKernelGGL(psfm_offsets_batch_kernel, dim3((unsigned)((n_max + 1 + PSFM_BLOCK - 1) / PSFM_BLOCK), (unsigned)n_seq), dim3(PSFM_BLOCK), 0, s,
                       dT, (const int64_t*)scan);
    hipLaunchKernelGGL(psfm_gather_batch_kernel, dim3((unsigned)((n_max + TILE_J - 1) / TILE_J), (unsigned)n_seq), dim3(PSFM_BLOCK), 0, s, dT,
                       (const int*)own->sort_lanes.as<int>());
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}
