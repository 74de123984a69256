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
#include "psfm_internal.h"

#define PSFM_BLOCK 256

// ------------------------------------------------------------------------------------------------
// K1  flow_check: one thread per pixel of one frame pair (blockIdx.y = pair).
// Algorithmic bytes per pair: 8P (F, streamed) + 8P (B, gathered near p+F) + P (occ) = 17P.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_flow_check_kernel(
    const float2* __restrict__ flows_f, const float2* __restrict__ flows_b, int H, int W, float cw, float ch,
    float thres, uint8_t* __restrict__ occ_out, float* __restrict__ err_out)
{
    const int64_t P = (int64_t)H * W;
    const int64_t p = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (p >= P) return;
    const int64_t base = (int64_t)blockIdx.y * P;
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    const float2 f = flows_f[base + p];
    // utils.py:73-78: pixel coordinate + flow in fp32
    const float X = __fadd_rn((float)x, f.x), Y = __fadd_rn((float)y, f.y);
    const PsfmTaps t = psfm_taps(X, Y, cw, ch, H, W);           // utils.py:79-82
    const float2 b = psfm_sample_flow(flows_b + base, H, W, t);
    // utils.py:87: torch.norm(warp + flow, dim=1) == sqrtf(fma(ev,ev, eu*eu))
    const float eu = __fadd_rn(b.x, f.x), ev = __fadd_rn(b.y, f.y);
    const float e = sqrtf(__fmaf_rn(ev, ev, __fmul_rn(eu, eu)));   // correctly rounded (hipcc default)
    // utils.py:58-68 (oob) and :88-91 (union)
    const bool oob = (X < 0.0f) | (X > (float)(W - 1)) | (Y < 0.0f) | (Y > (float)(H - 1));
    occ_out[base + p] = (uint8_t)((e > thres) | oob);
    if (err_out) err_out[base + p] = e;
}

psfm_status psfm_launch_flow_check(const float* ff, const float* fb, int n_pairs, int h, int w, float thres,
                                   uint8_t* occ, float* err, hipStream_t s)
{
    if (n_pairs <= 0) return PSFM_OK;
    const int64_t P = (int64_t)h * w;
    const float cw = (float)((double)(w - 1) / 2.0), ch = (float)((double)(h - 1) / 2.0);
    dim3 grid((unsigned)((P + PSFM_BLOCK - 1) / PSFM_BLOCK), (unsigned)n_pairs);
    hipLaunchKernelGGL(psfm_flow_check_kernel, grid, dim3(PSFM_BLOCK), 0, s, (const float2*)ff, (const float2*)fb,
                       h, w, cw, ch, thres, occ, err);
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
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_track_init_kernel(int64_t cap, int64_t G, int GW, int ratio,
                                                                     int* __restrict__ birth_frame,
                                                                     int* __restrict__ birth_idx,
                                                                     double2* __restrict__ log0,
                                                                     PsfmCounters* __restrict__ ctr,
                                                                     PsfmShard* __restrict__ shards)
{
    const int64_t i = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    if (i == 0) { ctr->n_lanes = (int)G; ctr->overflow = 0; }
    if (i < PSFM_NSHARD) { shards[i].fin_cnt = 0; shards[i].free_top = 0; }
    if (i >= cap) return;
    if (i < G) {
        birth_frame[i] = 0;
        birth_idx[i] = (int)i;
        log0[i] = make_double2((double)((int)(i % GW) * ratio), (double)((int)(i / GW) * ratio));
    } else {
        birth_frame[i] = -1;
    }
}

psfm_status psfm_launch_track_init(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s)
{
    PSFM_HIP(hipMemsetAsync(c->occupied.p, 0, (size_t)d.H * d.W, s));
    PSFM_HIP(hipMemsetAsync(c->survivors.p, 0, sizeof(int) * (size_t)(d.n_flows + 1), s));
    hipLaunchKernelGGL(psfm_track_init_kernel, dim3((unsigned)((d.cap + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                       0, s, d.cap, d.G, d.GW, d.ratio, c->birth_frame.as<int>(), c->birth_idx.as<int>(),
                       c->log.as<double2>(), c->counters.as<PsfmCounters>(), c->shards.as<PsfmShard>());
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// K2  chain_step: one thread per lane.
//   p = log[t][L];  flow = S(F_t, p), occ = S(occ_t, p) > 0.1     (trajectory.py:25-37, :50)
//   next = p + flow (f64);  valid = strictly inside               (trajectory.py:55-57)
//   alive -> log[t+1][L] = next, occupied[int(ny), int(nx)] = stamp   (trajectory.py:144-146)
//   dead  -> record (key, lane), push the lane on a free stack        (trajectory.py:140-142)
// Deaths are counted with wavefront ballots, summed per block in LDS and published with ONE atomic per
// block per table, on the shard blockIdx % PSFM_NSHARD (different words -> no serialisation).
// Algorithmic bytes per alive lane: 16 (p) + 32 (4 flow taps) + 4 (occ taps) + 16 (next) + 1 (occupied).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_chain_step_kernel(
    const float2* __restrict__ flow, const uint8_t* __restrict__ occ, int H, int W, float cw, float ch,
    const double2* __restrict__ log_cur, double2* __restrict__ log_next, int* __restrict__ birth_frame,
    const int* __restrict__ birth_idx, uint8_t* __restrict__ occupied, uint8_t stamp,
    const PsfmCounters* __restrict__ ctr, int* __restrict__ overflow, int* __restrict__ survivors_f,
    PsfmShard* __restrict__ shards, int* __restrict__ free_stack, unsigned long long* __restrict__ fin_keys,
    int* __restrict__ fin_lanes, int cap, int shard_cap, int free_cap, int frame, int shift_b, int shift_d)
{
    __shared__ int s_dead[PSFM_BLOCK / PSFM_WAVE];
    __shared__ int s_alive_any;
    __shared__ int s_base_fin, s_base_free;
    const int n_lanes = min(ctr->n_lanes, cap);
    if ((int)(blockIdx.x * PSFM_BLOCK) >= n_lanes) return;   // block-uniform
    const int i = blockIdx.x * PSFM_BLOCK + threadIdx.x;
    int bf = -1;
    if (i < n_lanes) bf = birth_frame[i];
    bool alive = false, dead = false;
    if (bf >= 0) {
        const double2 p = log_cur[i];
        const PsfmTaps t = psfm_taps((float)p.x, (float)p.y, cw, ch, H, W);
        const float2 fl = psfm_sample_flow(flow, H, W, t);
        const float oc = psfm_sample_mask(occ, H, W, t);
        const double nx = p.x + (double)fl.x, ny = p.y + (double)fl.y;
        const bool valid = (nx > 0.0) & (nx < (double)(W - 1)) & (ny > 0.0) & (ny < (double)(H - 1));
        alive = valid & !(oc > 0.1f);
        dead = !alive;
        if (alive) {
            log_next[i] = make_double2(nx, ny);
            occupied[(int64_t)((int)ny) * W + (int)nx] = stamp;
        }
    }
    const unsigned long long am = __ballot(alive);
    const unsigned long long dm = __ballot(dead);
    const int lane = psfm_lane_id(), wave = threadIdx.x / PSFM_WAVE;
    if (threadIdx.x == 0) s_alive_any = 0;
    if (lane == 0) s_dead[wave] = __popcll(dm);
    __syncthreads();
    if (lane == 0 && am != 0ull) s_alive_any = 1;   // benign race: every writer stores 1
    const int shard = blockIdx.x % PSFM_NSHARD;
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < PSFM_BLOCK / PSFM_WAVE; ++w) tot += s_dead[w];
        if (tot > 0) {
            s_base_fin = atomicAdd(&shards[shard].fin_cnt, tot);
            s_base_free = atomicAdd(&shards[shard].free_top, tot);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_alive_any) *survivors_f = 1;   // "some track survived" (respawn's degenerate rule)
    if (dead) {
        int r = psfm_rank_in(dm);
        for (int w = 0; w < wave; ++w) r += s_dead[w];
        birth_frame[i] = -1;
        const int fpos = s_base_free + r;
        if (fpos < free_cap) free_stack[(int64_t)shard * free_cap + fpos] = i;   // cannot overflow by construction
        const int rpos = s_base_fin + r;
        if (rpos < shard_cap) {
            const int64_t o = (int64_t)shard * shard_cap + rpos;
            fin_keys[o] = ((unsigned long long)frame << shift_d) | ((unsigned long long)bf << shift_b) |
                          (unsigned long long)birth_idx[i];
            fin_lanes[o] = i;
        } else {
            atomicOr(overflow, 2);
        }
    }
}

psfm_status psfm_launch_chain_step(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ,
                                   int frame, hipStream_t s)
{
    const uint8_t stamp = (uint8_t)((frame % 255) + 1);
    if (frame > 0 && (frame % 255) == 0) PSFM_HIP(hipMemsetAsync(c->occupied.p, 0, (size_t)d.H * d.W, s));
    double2* lg = c->log.as<double2>();
    PsfmCounters* ctr = c->counters.as<PsfmCounters>();
    hipLaunchKernelGGL(psfm_chain_step_kernel, dim3((unsigned)((d.cap + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK),
                       0, s, (const float2*)flow, occ, d.H, d.W, d.cw, d.ch, lg + (int64_t)frame * d.cap,
                       lg + (int64_t)(frame + 1) * d.cap, c->birth_frame.as<int>(), c->birth_idx.as<int>(),
                       c->occupied.as<uint8_t>(), stamp, ctr, &ctr->overflow, c->survivors.as<int>() + frame,
                       c->shards.as<PsfmShard>(), c->free_stack.as<int>(), c->fin_keys.as<unsigned long long>(),
                       c->fin_lanes.as<int>(), (int)d.cap, d.shard_cap, d.free_cap, frame, d.shift_b, d.shift_d);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// K3  respawn: one thread per stride-r grid point.
//   distance_transform_edt(1 - occupied) > r   (trajectory.py:150-151)
//     == no occupied pixel inside the integer disc dx^2+dy^2 <= r^2 (SURVEY A-5, pinned by fixtures);
//   with NO occupied pixel at all SciPy measures to a phantom feature at (y=-1,x=0): every grid point
//   but (0,0) respawns.  Births are counted per block (ballot + LDS); thread 0 pops that many lanes from
//   up to PSFM_PROBE free-stack shards (one atomic each) and takes fresh lanes for the remainder.
// ------------------------------------------------------------------------------------------------
#define PSFM_PROBE 8
__global__ __launch_bounds__(PSFM_BLOCK) void psfm_respawn_kernel(
    const uint8_t* __restrict__ occupied, uint8_t stamp, int H, int W, int ratio, int GW, int64_t G,
    const int* __restrict__ survivors_f, int* __restrict__ birth_frame, int* __restrict__ birth_idx,
    double2* __restrict__ log_next, PsfmCounters* __restrict__ ctr, PsfmShard* __restrict__ shards,
    const int* __restrict__ free_stack, int cap, int free_cap, int next_frame)
{
    __shared__ int s_births[PSFM_BLOCK / PSFM_WAVE];
    __shared__ int s_seg_start[PSFM_PROBE + 1];   // first free-stack index (absolute) or first fresh lane
    __shared__ int s_seg_end[PSFM_PROBE + 1];     // cumulative birth rank at which the segment ends
    __shared__ int s_nseg;
    const int64_t g = (int64_t)blockIdx.x * PSFM_BLOCK + threadIdx.x;
    bool birth = false;
    int cx = 0, cy = 0;
    if (g < G) {
        cx = (int)(g % GW) * ratio;
        cy = (int)(g / GW) * ratio;
        const int r2 = ratio * ratio;
        if (*survivors_f == 0) {
            birth = ((cy + 1) * (cy + 1) + cx * cx) > r2;
        } else {
            int hit = 0;
            for (int dy = -ratio; dy <= ratio; ++dy) {
                const int yy = cy + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -ratio; dx <= ratio; ++dx) {
                    const int xx = cx + dx;
                    if (dx * dx + dy * dy > r2 || xx < 0 || xx >= W) continue;
                    hit |= (occupied[(int64_t)yy * W + xx] == stamp);
                }
            }
            birth = !hit;
        }
    }
    const unsigned long long bm = __ballot(birth);
    const int lane = psfm_lane_id(), wave = threadIdx.x / PSFM_WAVE;
    if (lane == 0) s_births[wave] = __popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int need = 0;
        for (int w = 0; w < PSFM_BLOCK / PSFM_WAVE; ++w) need += s_births[w];
        int nseg = 0, done = 0;
        for (int k = 0; k < PSFM_PROBE && need > 0; ++k) {
            const int sh = (blockIdx.x + k * 7) % PSFM_NSHARD;
            const int old_top = atomicSub(&shards[sh].free_top, need);
            const int take = old_top < 0 ? 0 : (old_top > need ? need : old_top);
            if (take < need) atomicAdd(&shards[sh].free_top, need - take);   // give back what the stack did not have
            if (take > 0) {
                // entries [old_top - take, old_top) of shard sh; rank q in the segment -> index old_top-1-q
                s_seg_start[nseg] = sh * free_cap + old_top - 1;
                done += take;
                s_seg_end[nseg] = done;
                ++nseg;
                need -= take;
            }
        }
        if (need > 0) {
            const int base_new = atomicAdd(&ctr->n_lanes, need);
            s_seg_start[nseg] = -(base_new + 1);   // negative: fresh lanes base_new, base_new+1, ...
            done += need;
            s_seg_end[nseg] = done;
            ++nseg;
        }
        s_nseg = nseg;
    }
    __syncthreads();
    if (birth) {
        int r = psfm_rank_in(bm);
        for (int w = 0; w < wave; ++w) r += s_births[w];
        int k = 0, prev = 0;
        while (k < s_nseg - 1 && r >= s_seg_end[k]) { prev = s_seg_end[k]; ++k; }
        const int q = r - prev;
        const int st = s_seg_start[k];
        const int L = st >= 0 ? free_stack[st - q] : (-(st + 1) + q);
        if (L < cap) {
            birth_frame[L] = next_frame;
            birth_idx[L] = (int)g;
            log_next[L] = make_double2((double)cx, (double)cy);
        } else {
            atomicOr(&ctr->overflow, 1);
        }
    }
}

psfm_status psfm_launch_respawn(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s)
{
    const uint8_t stamp = (uint8_t)((frame % 255) + 1);
    double2* lg = c->log.as<double2>();
    hipLaunchKernelGGL(psfm_respawn_kernel, dim3((unsigned)((d.G + PSFM_BLOCK - 1) / PSFM_BLOCK)), dim3(PSFM_BLOCK), 0, s,
                       c->occupied.as<uint8_t>(), stamp, d.H, d.W, d.ratio, d.GW, d.G, c->survivors.as<int>() + frame,
                       c->birth_frame.as<int>(), c->birth_idx.as<int>(), lg + (int64_t)(frame + 1) * d.cap,
                       c->counters.as<PsfmCounters>(), c->shards.as<PsfmShard>(), c->free_stack.as<int>(), (int)d.cap,
                       d.free_cap, frame + 1);
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}
