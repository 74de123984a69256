// psfm_api.hip -- the extern "C" boundary declared in include/psfm.h: context, workspace, frame loop.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <shared_mutex>
#include <atomic>
#include <chrono>
#include <stdlib.h>

#include "psfm_internal.h"

static thread_local char g_err[512] = "";
// The per-device gate of psfm_internal.h (the one piece of process-wide state in the library, documented in psfm.h).
static std::shared_mutex g_dev_gate[16];
std::shared_mutex& psfm_device_gate(int device) { return g_dev_gate[device & 15]; }
static std::atomic<int> g_dev_waiters[16];
int psfm_gate_waiters(int device) { return g_dev_waiters[device & 15].load(std::memory_order_relaxed); }
void psfm_gate_waiters_add(int device, int d) { g_dev_waiters[device & 15].fetch_add(d, std::memory_order_relaxed); }
// Would this call run the persistent loop?  0 no, 1 yes if the device is free, 2 yes, wait for the device.
// Mode 0 decides by shape, from measurements on MI355X (scripts/probe_shapes.py; 100 frames, flow_check + recurrence +
// finalize, persistent / per-frame; round-4 sources, profiles/r04/r04_y_probe_shapes.txt): a frame is one dependent chain in
// either form -- the loop ~8 us of barrier + own step and ~5.5 us of births behind it whatever the frame size, a per-frame
// launch ~10 us + 22 us per million lanes -- so the loop wins where the lanes are many and births the exception:
//   psfm_track on ready maps: 1080p r=2 0.90, 4K r=4 0.93, 720p r=2 0.96, 1080p r=4 0.99 -- 436x1024 r=2 1.02, 480x854 r=4
//     (25 k grid points) 1.06, and r=1 (every pixel a grid point: a death is a birth one frame later, births are the bulk)
//     720p 1.00, 540x960 1.07, 480x640 1.11  ->  sample_ratio >= 2 and >= 100 k grid points;
//   psfm_connect with flow_check fused in: 1080p r=2 0.94 -- 720p r=2 1.02, 1080p r=4 1.62, 4K r=4 1.20 (inside the loop
//     flow_check costs its full bandwidth time, profiles/EXPERIMENTS.md 6.8; beside a short or flow_check-heavy step the
//     side stream of the per-frame path is better)  ->  additionally >= 400 k grid points and at most 6 pixels per grid point.
// What this policy cannot fix is the SIZE of a small sequence: below ~100 k grid points a frame is 8-13 us of dependent round trips
// in either form on 5-40 % of the device's block slots (configs[0]: 96 blocks, 0.025 of the HBM peak).  The answer for those shapes is
// not another way of running one sequence but several sequences per launch: psfm_connect_batch (psfm_batch.hip) -- blockIdx.y =
// sequence in every frame launch, one checkpoint and one finalize for the batch: 4.3x the sequences per second at 16 DAVIS-sized
// sequences (chain step 0.025 -> 0.17 of the HBM peak), 2.4x with path consistency at Sintel size (frame kernel 0.21 -> 0.38 of the
// VALU-issue peak), 1.3x at ScanNet size (a dense 307 k grid already fills the block slots once) -- profiles/r05.
static int psfm_wants_persist(psfm_ctx* c, bool optimize, int h, int w, int ratio, bool fused)
{
    if (c->chain_mode == 1 || ratio < 1 || !psfm_frame_ok(h, w)) return 0;
    // track_optimize: no persistent frame loop, but solves that reject steps run their trust-region loop as one persistent
    // launch (psfm_pc_resident_kernel) when the call has the device to itself -- take the gate if it is free.  A context with a
    // resident budget runs those launches on its share of the device's block slots beside other contexts': shared gate
    if (optimize) return c->resident_budget > 0 ? 0 : 1;
    const int64_t G = (int64_t)((w + ratio - 1) / ratio) * ((h + ratio - 1) / ratio);
    const int64_t P = (int64_t)h * w;
    const int maxb = psfm_persist_max_blocks(c);
    if (maxb <= 0 || (G + 255) / 256 > maxb) return 0;
    if (c->chain_mode == 2) return 2;
    if (ratio < 2 || G < 100000) return 0;
    if (fused && (G < 400000 || P > 6 * G)) return 0;
    return 1;
}

void psfm_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* psfm_last_error(void) { return g_err; }
extern "C" int psfm_version(void) { return PSFM_VERSION; }

extern "C" int psfm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

psfm_status PsfmBuf::ensure(size_t need)
{
    if (need <= bytes) return PSFM_OK;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    // round up so slightly different sequence lengths do not reallocate
    size_t want = (need + (size_t)(1 << 20) - 1) & ~((size_t)(1 << 20) - 1);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        psfm_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return PSFM_ERR_HIP;
    }
    bytes = want;
    return PSFM_OK;
}

void PsfmBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

// ---- profiler: HIP events on the launch stream around each kernel family -----------------------
hipEvent_t PsfmProfiler::get()
{
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void PsfmProfiler::begin(int kind, hipStream_t s)
{
    if (!enabled) return;
    Span sp; sp.kind = kind; sp.a = get(); sp.b = get();
    (void)hipEventRecord(sp.a, s);
    spans.push_back(sp);
}
void PsfmProfiler::kernel_span(int kind, hipEvent_t* a, hipEvent_t* b, bool always)
{
    if (!enabled || (!always && (calls++ % stride) != 0)) { *a = nullptr; *b = nullptr; return; }
    Span sp; sp.kind = kind; sp.a = get(); sp.b = get();
    spans.push_back(sp);
    *a = sp.a; *b = sp.b;
}
void PsfmProfiler::end(hipStream_t s)
{
    if (!enabled) return;
    (void)hipEventRecord(spans.back().b, s);
}
void PsfmProfiler::collect()
{
    for (auto& sp : spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { total_ms[sp.kind] += ms; launches[sp.kind]++; }
        pool.push_back(sp.a); pool.push_back(sp.b);
    }
    spans.clear();
}
void PsfmProfiler::reset()
{
    for (int k = 0; k < PSFM_PROF_KINDS; ++k) { total_ms[k] = 0; launches[k] = 0; }
}
void PsfmProfiler::destroy()
{
    collect();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
}

#define PSFM_CHECK_CTX(c)                                                     \
    do {                                                                      \
        if (!(c)) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }     \
        PSFM_HIP(hipSetDevice((c)->device));                                  \
    } while (0)

extern "C" psfm_status psfm_ctx_create(int device, psfm_ctx** out)
{
    if (!out) { psfm_set_error("psfm_ctx_create: out is NULL"); return PSFM_ERR_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        psfm_set_error("no HIP device available (%s); libpsfm_hip has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return PSFM_ERR_HIP;
    }
    if (device < 0 || device >= n) { psfm_set_error("device %d out of range [0,%d)", device, n); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(device));
    psfm_ctx* c = new psfm_ctx();
    c->device = device;
    c->host_pinned_bytes = 512 + sizeof(PsfmShard) * PSFM_NSHARD + 4096;
    e = hipHostMalloc(&c->host_pinned, c->host_pinned_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { delete c; psfm_set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return PSFM_ERR_HIP; }
    *out = c;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_destroy(psfm_ctx* c)
{
    if (!c) return PSFM_OK;
    (void)hipSetDevice(c->device);
    PsfmBuf* bufs[] = {&c->log, &c->birth_frame, &c->birth_idx, &c->free_stack, &c->fin_keys, &c->fin_lanes,
                       &c->occupied, &c->counters, &c->shards, &c->survivors, &c->sort_keys, &c->sort_lanes, &c->sort_tmp,
                       &c->scan_tmp, &c->fin_marks, &c->res_birth, &c->res_len, &c->res_off, &c->res_xy, &c->sol_x, &c->sol_state,
                       &c->sol_partials, &c->sol_ctrl, &c->sol_misc, &c->sol_stats, &c->sol_fused, &c->sol_bar, &c->sol_list, &c->occ_own, &c->occ2_own,
                       &c->handoff, &c->seg_info, &c->seg_table, &c->persist_bar, &c->batch_tab, &c->batch_ws, &c->batch_fc, &c->win_ws, &c->flt_ids, &c->flt_birth, &c->flt_len, &c->flt_off, &c->flt_xy,
                       &c->mt_kp_off, &c->mt_q, &c->mt_pts, &c->mt_kp_ind, &c->mt_kp_xy, &c->mt_moff, &c->mt_keys, &c->mt_rows, &c->mt_gid, &c->mt_pairs};
    for (auto b : bufs) b->release();
    for (void*& q : c->peer_opened) { if (q) (void)hipIpcCloseMemHandle(q); q = nullptr; }
    c->peer_area.release();
    psfm_shard_abandon(c);
    c->prof.destroy();
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->redo_stream) (void)hipStreamDestroy(c->redo_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->copy_stream2) (void)hipStreamDestroy(c->copy_stream2);
    for (void* p : c->ingest_slots) (void)hipHostFree(p);
    for (hipEvent_t e : c->ingest_events) (void)hipEventDestroy(e);
    if (c->host_pinned) (void)hipHostFree(c->host_pinned);
    if (c->host_seg) (void)hipHostFree(c->host_seg);
    if (c->host_batch) (void)hipHostFree(c->host_batch);
    if (c->host_batch2) (void)hipHostFree(c->host_batch2);
    delete c;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_set_capacity(psfm_ctx* c, double lane_factor, double traj_factor)
{
    if (!c || !(lane_factor >= 1.0) || !(traj_factor >= 1.0)) { psfm_set_error("psfm_ctx_set_capacity: factors must be >= 1"); return PSFM_ERR_ARG; }
    c->lane_factor = lane_factor;
    c->traj_factor = traj_factor;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_set_chain_mode(psfm_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 2) { psfm_set_error("psfm_ctx_set_chain_mode: mode must be 0 (auto), 1 (per-frame launches) or 2 (persistent loop)"); return PSFM_ERR_ARG; }
    c->chain_mode = mode;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_set_solver(psfm_ctx* c, int mode, int k)
{
    if (!c || mode < 0 || mode > 2 || k < 0 || k > psfm_solve_kmax()) {
        psfm_set_error("psfm_ctx_set_solver: mode must be 0 (adaptive), 1 (launch chain) or 2 (fused solve), k in [0, %d]", psfm_solve_kmax());
        return PSFM_ERR_ARG;
    }
    c->solver_mode = mode;
    c->solver_K = k;
    return PSFM_OK;
}

extern "C" psfm_status psfm_solver_counters(psfm_ctx* c, int64_t* fused, int64_t* fused_redone, int64_t* chain, int32_t* k_now)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    if (fused) *fused = c->n_fused_ok;
    if (fused_redone) *fused_redone = c->n_fused_redone;
    if (chain) *chain = c->n_chain;
    if (k_now) *k_now = c->solve_K;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_set_resident_budget(psfm_ctx* c, int blocks)
{
    if (!c || blocks < 0) { psfm_set_error("psfm_ctx_set_resident_budget: blocks must be >= 0"); return PSFM_ERR_ARG; }
    c->resident_budget = blocks;
    return PSFM_OK;
}

extern "C" psfm_status psfm_resident_capacity(psfm_ctx* c, int32_t* blocks)
{
    PSFM_CHECK_CTX(c);
    if (blocks) *blocks = psfm_resident_blocks(c);
    return PSFM_OK;
}

extern "C" psfm_status psfm_solver_launches(psfm_ctx* c, int64_t* resident, int64_t* giveups, int64_t* iterations)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    if (resident) *resident = c->n_resident;
    if (giveups) *giveups = c->pc_giveups;
    if (iterations) *iterations = c->n_iter_launches;
    return PSFM_OK;
}

extern "C" psfm_status psfm_ctx_set_profiling(psfm_ctx* c, int enable)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    c->prof.enabled = enable != 0;
    c->prof.stride = enable > 1 ? enable : 1;   // enable = N > 1: time every N-th chain_step launch only
    c->prof.calls = 0;
    c->prof.collect();
    c->prof.reset();
    return PSFM_OK;
}

extern "C" psfm_status psfm_profile_get(psfm_ctx* c, int kind, double* total_ms, int64_t* launches)
{
    if (!c || kind < 0 || kind >= PSFM_PROF_KINDS) { psfm_set_error("psfm_profile_get: bad argument"); return PSFM_ERR_ARG; }
    if (total_ms) *total_ms = c->prof.total_ms[kind];
    if (launches) *launches = c->prof.launches[kind];
    return PSFM_OK;
}

extern "C" psfm_status psfm_flow_check(psfm_ctx* c, const float* flows_f, const float* flows_b, int n_pairs, int h, int w,
                                       float thres, uint8_t* occ_out, float* err_out, void* stream)
{
    PSFM_CHECK_CTX(c);
    PsfmGate gate(c->device, 0);
    if (n_pairs < 0 || !psfm_frame_ok(h, w) || (n_pairs > 0 && (!flows_f || !flows_b || !occ_out))) {
        psfm_set_error("psfm_flow_check: bad argument (n_pairs=%d h=%d w=%d)", n_pairs, h, w);
        return PSFM_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    c->prof.begin(PSFM_PROF_FLOW_CHECK, s);
    psfm_status st = psfm_launch_flow_check(flows_f, flows_b, n_pairs, h, w, thres, occ_out, err_out, s);
    c->prof.end(s);
    return st;
}

extern "C" psfm_status psfm_grid_sample(psfm_ctx* c, const float* map_hwc, int ch, int h, int w, const double* xy,
                                        int64_t n, float* out, void* stream)
{
    PSFM_CHECK_CTX(c);
    PsfmGate gate(c->device, 0);
    if ((ch != 1 && ch != 2) || !psfm_frame_ok(h, w) || n < 0 || (n > 0 && (!map_hwc || !xy || !out))) {
        psfm_set_error("psfm_grid_sample: bad argument (c=%d h=%d w=%d n=%lld)", ch, h, w, (long long)n);
        return PSFM_ERR_ARG;
    }
    return psfm_launch_grid_sample(map_hwc, ch, h, w, xy, n, out, (hipStream_t)stream);
}

extern "C" psfm_status psfm_optimize_location(psfm_ctx* c, const double* uv12, const double* ref1, const double* ref2,
                                              const double* scale, const float* flow12, int64_t n, int w, int h,
                                              double* out, psfm_solve_stats* stats_host, void* stream)
{
    PSFM_CHECK_CTX(c);
    // exclusive if no other psfm call is in flight: the solve may then run as ONE resident launch; psfm_ctx_set_chain_mode(ctx, 1)
    // -- what callers that overlap several solves from several host threads set -- keeps the gate shared and the solve on launches
    PsfmGate gate(c->device, (c->chain_mode == 1 || c->resident_budget > 0) ? 0 : 1);
    c->pc_persist_ok = gate.exclusive || c->resident_budget > 0;
    c->pc_giveups = 0;
    c->n_resident = c->n_iter_launches = 0;
    if (n < 0 || !psfm_frame_ok(h, w) || (n > 0 && (!uv12 || !ref1 || !ref2 || !scale || !flow12 || !out))) {
        psfm_set_error("psfm_optimize_location: bad argument (n=%lld h=%d w=%d)", (long long)n, h, w);
        return PSFM_ERR_ARG;
    }
    psfm_solve_stats st;
    memset(&st, 0, sizeof(st));
    hipStream_t s = (hipStream_t)stream;
    c->prof.begin(PSFM_PROF_SOLVER, s);
    psfm_status rc = psfm_solve_batch(c, uv12, ref1, ref2, scale, flow12, n, w, h, out, &st, s);
    c->prof.end(s);
    if (stats_host) *stats_host = st;
    return rc;
}

extern "C" psfm_status psfm_path_consistency_eval(psfm_ctx* c, const double* uv12, const double* ref1, const double* ref2,
                                                  const double* scale, const float* flow12, int64_t n, int w, int h,
                                                  double* residuals, double* jacobians, void* stream)
{
    PSFM_CHECK_CTX(c);
    PsfmGate gate(c->device, 0);
    if (n < 0 || n > 0x3fffffff || !psfm_frame_ok(h, w) || (n > 0 && (!uv12 || !ref1 || !ref2 || !scale || !flow12 || (!residuals && !jacobians)))) {
        psfm_set_error("psfm_path_consistency_eval: bad argument (n=%lld h=%d w=%d)", (long long)n, h, w);
        return PSFM_ERR_ARG;
    }
    return psfm_launch_pc_eval(uv12, ref1, ref2, scale, flow12, n, w, h, residuals, jacobians, (hipStream_t)stream);
}

extern "C" psfm_status psfm_sort_records(psfm_ctx* c, uint32_t* keys, int32_t* values, int64_t n, int end_bit, void* stream)
{
    PSFM_CHECK_CTX(c);
    PsfmGate gate(c->device, 0);
    if (n < 0 || n >= (1ll << 31) || end_bit < 1 || end_bit > 32 || (n > 0 && (!keys || !values))) {
        psfm_set_error("psfm_sort_records: bad argument (n=%lld end_bit=%d)", (long long)n, end_bit);
        return PSFM_ERR_ARG;
    }
    if (n == 0) return PSFM_OK;
    hipStream_t s = (hipStream_t)stream;
    psfm_status st;
    // the sort ping-pongs between two halves and ends in the first: stage through the context's sort buffers
    if ((st = c->sort_keys.ensure(sizeof(unsigned long long) * (size_t)n * 2)) != PSFM_OK) return st;
    if ((st = c->sort_lanes.ensure(sizeof(int) * (size_t)n * 2)) != PSFM_OK) return st;
    const int64_t half = psfm_sort_pairs32_passes((unsigned)end_bit) % 2 == 0 ? 0 : n;
    unsigned* k0 = c->sort_keys.as<unsigned>();
    int* v0 = c->sort_lanes.as<int>();
    PSFM_HIP(hipMemcpyAsync(k0 + half, keys, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, s));
    PSFM_HIP(hipMemcpyAsync(v0 + half, values, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
    if ((st = psfm_sort_pairs32(c, k0, v0, k0 + n, v0 + n, n, (unsigned)end_bit, s)) != PSFM_OK) return st;
    PSFM_HIP(hipMemcpyAsync(keys, k0, sizeof(unsigned) * (size_t)n, hipMemcpyDeviceToDevice, s));
    PSFM_HIP(hipMemcpyAsync(values, v0, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
    return PSFM_OK;
}

// Sizes of one run of the frame recurrence.  g_own < 0: the whole stride-r grid; otherwise the number of grid points whose
// births this process owns (track-sharded runs): lane / record tables are sized for those, keys for the whole grid.
psfm_status psfm_track_dims(psfm_ctx* c, int n_flows, int h, int w, int ratio, int64_t g_own, PsfmTrackDims& d)
{
    d.H = h; d.W = w; d.ratio = ratio; d.n_flows = n_flows;
    d.GW = (w + ratio - 1) / ratio; d.GH = (h + ratio - 1) / ratio;   // trajectory.py:110-115
    d.G = (int64_t)d.GW * d.GH;
    const int64_t Gown = g_own < 0 ? d.G : (g_own > 0 ? g_own : 1);
    d.cap = (int64_t)(c->lane_factor * (double)Gown);
    d.cap = ((d.cap + 255) / 256) * 256;
    const int64_t n_blocks = d.cap / 256;
    {   // free-lane stacks: one per block of the grid up to PSFM_NSHARD (a small grid must not probe stacks nobody fills)
        const int64_t gb = (Gown + 255) / 256;
        d.nsh = (int)(gb < PSFM_NSHARD ? (gb > 0 ? gb : 1) : PSFM_NSHARD);
    }
    d.free_cap = (int)(((n_blocks + d.nsh - 1) / d.nsh) * 256) + 1024;   // per-stack entries: every lane of the blocks that push there
    // records are spread over PSFM_NSHARD slices by block index: give every slice head-room
    // trajectory records: traj_factor x G, but at least G x n_flows / 8 (one death in eight per frame and grid point)
    const double tf = c->traj_factor > (double)n_flows / 8.0 ? c->traj_factor : (double)n_flows / 8.0;
    // (a slice is filled by the blocks whose index it is modulo PSFM_NSHARD: a grid of fewer than PSFM_NSHARD blocks uses that many
    // slices only -- round 5: a 736-point grid put 9 865 records into three slices sized for a 64th of them each)
    d.shard_cap = (int)(((int64_t)(tf * (double)Gown) + d.cap) / d.nsh) + 1024;
    d.traj_cap = (int64_t)d.shard_cap * PSFM_NSHARD;
    if (d.cap > 0x7fffffff / 2 || d.traj_cap > 0x7fffffff / 2) { psfm_set_error("psfm_track: grid too large"); return PSFM_ERR_ARG; }
    d.shift_b = 1; while ((1ll << d.shift_b) < d.G) ++d.shift_b;
    int tbits = 1; while ((1ll << tbits) < (long long)n_flows + 2) ++tbits;
    d.shift_d = d.shift_b + tbits;
    if (d.shift_d + tbits > 63) { psfm_set_error("psfm_track: key does not fit 64 bits"); return PSFM_ERR_ARG; }
    d.cw = (float)((double)(w - 1) / 2.0); d.ch = (float)((double)(h - 1) / 2.0);
    return PSFM_OK;
}

psfm_status psfm_track_alloc(psfm_ctx* c, const PsfmTrackDims& d)
{
    psfm_status st;
    const int n_flows = d.n_flows;
    if ((st = c->log.ensure(sizeof(double2) * (size_t)(n_flows + 1) * d.cap)) != PSFM_OK) return st;
    if ((st = c->birth_frame.ensure(sizeof(int) * d.cap)) != PSFM_OK) return st;
    if ((st = c->birth_idx.ensure(sizeof(int) * d.cap)) != PSFM_OK) return st;
    if ((st = c->free_stack.ensure(sizeof(int) * (size_t)d.free_cap * PSFM_NSHARD * 2)) != PSFM_OK) return st;
    if ((st = c->shards.ensure(sizeof(PsfmShard) * PSFM_NSHARD * 2)) != PSFM_OK) return st;
    if ((st = c->fin_keys.ensure(sizeof(unsigned long long) * d.traj_cap)) != PSFM_OK) return st;
    if ((st = c->fin_lanes.ensure(sizeof(int) * d.traj_cap)) != PSFM_OK) return st;
    if ((st = c->occupied.ensure((size_t)d.G * 3 + 8)) != PSFM_OK) return st;   // (the persistent loop rotates three maps)
    if ((st = c->counters.ensure(sizeof(PsfmCounters))) != PSFM_OK) return st;
    if ((st = c->survivors.ensure(sizeof(int) * (size_t)(n_flows + 1))) != PSFM_OK) return st;
    return PSFM_OK;
}

// Occlusion maps produced on a side stream while the frame loop consumes them (psfm_connect): one event per chunk
// of frame pairs; the loop waits for the chunk that contains the pair it is about to read.
struct PsfmOccPipeline {
    int chunk = 0;                       // frame pairs per chunk (0 = maps already complete)
    std::vector<hipEvent_t> ready;       // stride-1 maps: chunk c covers pairs [c*chunk, (c+1)*chunk)
    std::vector<hipEvent_t> ready_s2;    // stride-2 maps
    int waited = -1, waited_s2 = -1;
    psfm_status need(int pair, bool s2, hipStream_t s)
    {
        if (chunk <= 0) return PSFM_OK;
        std::vector<hipEvent_t>& ev = s2 ? ready_s2 : ready;
        int& w = s2 ? waited_s2 : waited;
        const int cidx = pair / chunk;
        while (w < cidx && w + 1 < (int)ev.size()) {
            ++w;
            PSFM_HIP(hipStreamWaitEvent(s, ev[w], 0));
        }
        return PSFM_OK;
    }
};

static psfm_status psfm_track_impl(psfm_ctx* c, const float* flows, const uint8_t* occ, const float* flows_f2,
                                   const uint8_t* occ_s2, int n_flows, int h, int w, int ratio, psfm_track_info* info,
                                   void* stream, PsfmOccPipeline* pipe, PsfmGate& gate,
                                   const float* fuse_flows_b = nullptr, float fuse_thres = 0.f, int64_t occ_pitch = 0)
{
    const bool device_is_ours = gate.exclusive;   // (no other psfm call of this process is in flight on the device)
    // fuse_flows_b != NULL (psfm_connect): `occ` is still EMPTY -- the persistent loop computes the maps itself (fused
    // flow_check); any other way of running the recurrence first fills them with the stand-alone kernel.
    if (occ_pitch == 0) occ_pitch = (int64_t)h * w;
    PSFM_CHECK_CTX(c);
    // a psfm_shard_begin run that was never finished: this call overwrites its lane tables, so it ends here (a later
    // psfm_shard_* call on the context reports "no sharded run in progress" instead of stepping foreign state)
    psfm_shard_abandon(c);
    const bool optimize = flows_f2 != nullptr;
    if (n_flows < 1 || !psfm_frame_ok(h, w) || ratio < 1 || !flows || !occ || (optimize && !occ_s2 && n_flows > 1)) {
        psfm_set_error("psfm_track: bad argument (n_flows=%d h=%d w=%d ratio=%d)", n_flows, h, w, ratio);
        return PSFM_ERR_ARG;
    }
    if (ratio > 64) { psfm_set_error("psfm_track: sample_ratio %d > 64 unsupported", ratio); return PSFM_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    PsfmTrackDims d;
    psfm_status st;
    if ((st = psfm_track_dims(c, n_flows, h, w, ratio, -1, d)) != PSFM_OK) return st;
    const int64_t P = (int64_t)h * w;
    if ((st = psfm_track_alloc(c, d)) != PSFM_OK) return st;

    c->solve_stats.clear();
    c->res_n_traj = c->res_n_points = 0;
    c->res_n_flows = n_flows;
    c->pc_persist_ok = device_is_ours || c->resident_budget > 0;
    c->pc_giveups = 0;

    // ---- track mode: the whole recurrence as ONE persistent launch when every lane can be resident at once ----
    if (!optimize && c->chain_mode != 1) {
        const int maxb = psfm_persist_max_blocks(c);
        const int64_t need_blocks = (d.G + 255) / 256;
        if (maxb > 0 && need_blocks <= maxb && device_is_ours) {   // (not ours: another psfm call is in flight on this device)
            PsfmTrackDims dp = d;
            // spare lanes for tracks born faster than lanes come back (alive tracks can exceed the grid where flows
            // converge): twice the grid while that is cheap, a quarter more otherwise, never more than fits the device
            int64_t nb = need_blocks <= 256 ? 2 * need_blocks + 4 : need_blocks + need_blocks / 4;
            dp.nblk = (int)(nb < maxb ? nb : maxb);
            // fused flow_check: the occlusion maps are computed by whatever blocks exist -- enough of them to cover the
            // image in one round of 1024-pixel chunks (blocks without grid points only check flows and lend lanes)
            if (fuse_flows_b) {
                const int64_t fc_blocks = (P + 1023) / 1024;
                const int64_t want = fc_blocks < maxb ? fc_blocks : maxb;
                if (want > dp.nblk) dp.nblk = (int)want;
            }
            dp.cap = (int64_t)dp.nblk * (256 + psfm_persist_guests());   // log columns: thread lanes, then guest lanes
            dp.nsh = dp.nblk < PSFM_NSHARD ? dp.nblk : PSFM_NSHARD;
            dp.free_cap = (int)(((dp.nblk + dp.nsh - 1) / dp.nsh) * 256) + 1024;
            dp.seg_cap = (int)(d.traj_cap / dp.nblk) + 64;
            dp.spill_cap = (int)(d.traj_cap / 8) + 4096;
            const int64_t fin_total = (int64_t)dp.nblk * dp.seg_cap + dp.spill_cap;
            if ((st = c->log.ensure(sizeof(double2) * (size_t)(n_flows + 1) * dp.cap)) != PSFM_OK) return st;
            if ((st = c->free_stack.ensure(sizeof(int) * (size_t)dp.free_cap * PSFM_NSHARD * 2)) != PSFM_OK) return st;
            if ((st = c->fin_keys.ensure(sizeof(unsigned long long) * fin_total)) != PSFM_OK) return st;
            if ((st = c->fin_lanes.ensure(sizeof(int) * fin_total)) != PSFM_OK) return st;
            if ((st = c->handoff.ensure((size_t)dp.nblk * 256 * 24)) != PSFM_OK) return st;
            if ((st = c->seg_info.ensure(sizeof(int) * 2 * (size_t)dp.nblk)) != PSFM_OK) return st;
            if ((st = c->persist_bar.ensure((size_t)(4 * PSFM_NSHARD + 1) * 128)) != PSFM_OK) return st;
            if (pipe && (st = pipe->need(n_flows - 1, false, s)) != PSFM_OK) return st;   // every occlusion map
            bool fallback = false;
            {
                const bool trace = getenv("PSFM_TRACE") != nullptr;
                const auto t1 = std::chrono::steady_clock::now();
                st = psfm_launch_chain_persist(c, dp, flows, occ, occ_pitch, fuse_flows_b, fuse_thres, s);
                if (st == PSFM_ERR_CAPACITY) {      // the cooperative launch was refused: per-frame launches below
                    fallback = true;
                    st = PSFM_OK;
                } else {
                    if (st != PSFM_OK) return st;
                    c->prof.begin(PSFM_PROF_FINALIZE, s);
                    st = psfm_finalize_persist(c, dp, &fallback, s);
                    c->prof.end(s);
                }
                if (trace) {
                    const auto t2 = std::chrono::steady_clock::now();
                    fprintf(stderr, "[psfm %p] persistent loop: launch+finalize %.2f ms, fallback %d, overflow %d\n", (void*)c,
                            std::chrono::duration<double, std::milli>(t2 - t1).count(), (int)fallback,
                            ((PsfmCounters*)c->host_pinned)->overflow);
                }
            }
            if (st != PSFM_OK) return st;
            if (!fallback) {
                PSFM_HIP(hipStreamSynchronize(s));
                c->prof.collect();
                if (info) {
                    memset(info, 0, sizeof(*info));
                    info->n_traj = c->res_n_traj;
                    info->n_points = c->res_n_points;
                    info->n_lanes_peak = ((PsfmCounters*)c->host_pinned)->n_lanes;
                    info->lane_capacity = dp.cap;
                    info->chain_mode = 2;
                }
                return PSFM_OK;
            }
            c->prof.collect();
        }
    }
    if (fuse_flows_b) {   // the persistent loop did not run (or gave up): the maps it would have produced, stand-alone
        c->prof.begin(PSFM_PROF_FLOW_CHECK, s);
        for (int f = 0; f < n_flows; ++f)
            if ((st = psfm_launch_flow_check(flows + (size_t)f * P * 2, fuse_flows_b + (size_t)f * P * 2, 1, h, w, fuse_thres,
                                             const_cast<uint8_t*>(occ) + (size_t)f * occ_pitch, nullptr, s)) != PSFM_OK) return st;
        c->prof.end(s);
    }
    if ((st = psfm_launch_track_init(c, d, s)) != PSFM_OK) return st;
    int64_t total_iters = 0;
    // Frame loop.  In track_optimize mode nothing returns to the host inside a window of PSFM_CHECK frames: each
    // solve is enqueued with `solve_unroll` iterations; a solve that needs more raises a device-side stall flag that
    // turns every later launch into a no-op, and the checkpoint below resumes it and re-enqueues from there.
    // frames between two host checkpoints (every checkpoint drains the queue).  Round 1 (150 us per frame): 8 -> 16 gained
    // 1.5 %, 32 nothing more; PSFM_CHECK_FRAMES overrides for measurements
    static const int check_env = getenv("PSFM_CHECK_FRAMES") ? atoi(getenv("PSFM_CHECK_FRAMES")) : 0;
    const int PSFM_CHECK = check_env >= 2 ? check_env : 16;
    std::vector<psfm_solve_stats> hstats((size_t)n_flows + 1);
    int first_unchecked = 1;
    if (optimize) {
        if ((st = c->sol_stats.ensure(sizeof(psfm_solve_stats) * (size_t)(n_flows + 1))) != PSFM_OK) return st;
    }
    // (tests: PSFM_SOLVE_UNROLL=1 makes every solve that needs more than two iterations stall and resume)
    const char* unroll_env = getenv("PSFM_SOLVE_UNROLL");
    const int unroll_fixed = unroll_env ? (atoi(unroll_env) < 1 ? 1 : atoi(unroll_env)) : 0;
    // How a frame's solve is enqueued (psfm_ctx_set_solver): the fused solve -- ONE launch that speculates solve_K
    // Gauss-Newton iterations -- or the launch chain (init + solve_unroll iterations + write-back).  Adaptive: a
    // window (the frames between two checkpoints) whose solves rejected steps / left the Gauss-Newton path sends the next
    // window to the chain, a clean window brings the fused solve back; solve_K follows the accepted steps seen.
    c->n_fused_ok = c->n_fused_redone = c->n_chain = 0;
    c->n_resident = c->n_iter_launches = 0;
    if (optimize && (st = psfm_solve_prepare(c, d, s)) != PSFM_OK) return st;
    // PSFM_MERGE_FRAME=0: chain step and fused solve as two launches (what the merged frame kernel is measured against);
    // PSFM_SEQ=0: host-paced frame kernels (one per frame, stall + redo when a solve needs more iterations than speculated)
    static const bool merge = !(getenv("PSFM_MERGE_FRAME") && atoi(getenv("PSFM_MERGE_FRAME")) == 0);
    const bool seq_env = !(getenv("PSFM_SEQ") && atoi(getenv("PSFM_SEQ")) == 0);
    bool seq_ok = optimize && merge && seq_env && unroll_fixed == 0;
    int launch_id = 0;            // device-paced windows: id of the next psfm_seq_kernel launch (== PsfmCounters::pc_owner)
    int idle_windows = 0;         // ... consecutive windows in which no frame completed (the device inside one long solve)
    bool pc_in_step = true;       // the device's program counter is where the host thinks it is (track_init: frame 1, launch 0)
    int* hpc = (int*)((char*)c->host_pinned + 320);     // pinned staging for {pc_frame, pc_phase, pc_owner, solve_K}
    PsfmCounters* dctr = c->counters.as<PsfmCounters>();

    // One checkpoint: counters + the window's statistics to the host, the stalled solve (if any) redone, statistics folded
    // into the result and into the adaptation of mode / K / unroll.  f_hi: last frame whose launch has been enqueued.
    // Returns the first frame that is NOT complete in *f_next.
    auto checkpoint = [&](int f_hi, bool fused_now, bool seq, int* f_next) -> psfm_status {
        PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
        PSFM_HIP(hipMemcpyAsync(hc, c->counters.p, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
        PSFM_HIP(hipMemcpyAsync(hstats.data() + first_unchecked, c->sol_stats.as<psfm_solve_stats>() + first_unchecked,
                                sizeof(psfm_solve_stats) * (size_t)(f_hi - first_unchecked + 1), hipMemcpyDeviceToHost, s));
        PSFM_HIP(hipStreamSynchronize(s));
        // Nothing of this call is in flight now.  If other psfm calls of the process wait for the device (this sequence holds it
        // exclusively for its resident solves), let them in: the rest of the sequence runs its solves as launches, which overlap
        // with other sequences -- a late-comer waits for one window of frames at most, not for the whole sequence.
        if (gate.yield_exclusive()) c->pc_persist_ok = c->resident_budget > 0;
        // A full table ends the sequence here: a launch behind that point counts on blocks the grid does not have (its solve never
        // sees its last arrival: no stall flag, no progress), and finalize would refuse the result anyway.  The caller raises the
        // capacity and runs the sequence again (the Python mirror does: trajectory.run_connect).
        if ((hc->overflow & 7) != 0 || hc->n_lanes > d.cap) {
            psfm_set_error("capacity exceeded: lanes used %d of %lld, overflow bits %d (frame %d of %d); raise psfm_ctx_set_capacity",
                           hc->n_lanes, (long long)d.cap, hc->overflow, hc->pc_frame, n_flows);
            return PSFM_ERR_CAPACITY;
        }
        const int stalled = hc->stall ? hc->stall - 1 : -1;   // (the redo below reuses the pinned block `hc` points at)
        int last_ok = f_hi;
        if (seq) {      // frames below the device's program counter are complete (it may be in the middle of the next solve)
            const int pcf = hc->pc_frame < n_flows ? hc->pc_frame : n_flows;
            last_ok = pcf - 1;
        }
        if (stalled >= 0) {
            const int fs = stalled;
            psfm_solve_stats ss;
            memset(&ss, 0, sizeof(ss));
            // (a fused solve that ran out of iterations is first retried with two iterations more; the chain
            // takes what is left -- and every stalled solve of a chain window)
            const int k_used = c->solver_K > 0 ? c->solver_K : c->solve_K;
            psfm_status st2 = psfm_solve_frame_resume(c, d, flows + (size_t)(fs - 1) * P * 2, flows + (size_t)fs * P * 2,
                                                      flows_f2 + (size_t)(fs - 1) * P * 2, occ_s2 + (size_t)(fs - 1) * P, fs, &ss,
                                                      (fused_now && !seq && c->solver_K == 0 && k_used < psfm_solve_kmax())
                                                          ? (k_used + 2 < psfm_solve_kmax() ? k_used + 2 : psfm_solve_kmax()) : 0,
                                                      !fused_now, s);
            if (st2 != PSFM_OK) return st2;
            hstats[fs] = ss;
            last_ok = fs;
            pc_in_step = false;     // the device's counter still points at the redone frame
        }
        int max_it = 0, n_solved = 0, n_unclean = 0, k_need = 2;
        for (int k = first_unchecked; k <= last_ok; ++k) {
            // (termination 5 = Ceres' FAILURE: the reference ignores it, trajectory_optimize.cpp:81-82, and carries on
            // with the positions it had -- so does the device; the caller sees it in the solve statistics)
            if (hstats[k].termination >= 0) {   // -1: no track had a full buffer, nothing was solved
                c->solve_stats.push_back(hstats[k]);
                total_iters += hstats[k].iterations;
                // "clean": every iteration but the terminating one took the Gauss-Newton step and was accepted --
                // what the fused solve speculates; it then needs successful_steps + 1 iterations in its launch
                const psfm_solve_stats& q = hstats[k];
                const bool clean = q.dogleg_nonGN == 0 && q.termination != PSFM_TERM_FAILURE &&
                                   (q.iterations == q.successful_steps + 1 ||
                                    (q.termination == PSFM_TERM_GRADIENT_TOL && q.iterations == q.successful_steps)) &&
                                   q.successful_steps + 1 <= psfm_solve_kmax();
                ++n_solved;
                if (!clean) ++n_unclean;
                else if (q.successful_steps + 1 > k_need) k_need = q.successful_steps + 1;
                if (!fused_now) ++c->n_chain;
                else if (k == stalled) ++c->n_fused_redone;
                else ++c->n_fused_ok;
            }
            if (hstats[k].iterations > max_it) max_it = hstats[k].iterations;
        }
        if (n_solved > 0) {
            c->solve_mode = (n_unclean * 8 > n_solved) ? 1 : 0;
            if (seq) {
                // device-paced: an iteration more than speculated costs one more launch, not a redo -- follow the MOST COMMON
                // need of the window instead of its maximum
                int hist[16] = {0};
                for (int k = first_unchecked; k <= last_ok; ++k)
                    if (hstats[k].termination >= 0) ++hist[hstats[k].successful_steps + 1 < 15 ? hstats[k].successful_steps + 1 : 15];
                int best = 2;
                for (int q = 2; q <= psfm_solve_kmax(); ++q) if (hist[q] > hist[best]) best = q;
                c->solve_K = best;
            } else {
                c->solve_K = k_need > c->solve_K ? k_need : c->solve_K - (c->solve_K - k_need + 1) / 2;
            }
        }
        // adapt the unroll to what this sequence needs, within [4, 64]: a solve of k iterations needs k-1 pc_iter
        // launches (pc_init does the first), so max+1 leaves two spare launches for the slowest solve seen in the
        // window (a spare launch is a no-op that costs < 1 us behind another one; a solve that still runs out raises the stall flag)
        int want = max_it + 1;
        want = want < 4 ? 4 : (want > 64 ? 64 : want);
        c->solve_unroll = want > c->solve_unroll ? want : c->solve_unroll - (c->solve_unroll - want + 1) / 2;   // decays all the way
        first_unchecked = last_ok + 1;
        *f_next = last_ok + 1;   // after a stall: re-enqueue the (poisoned) frames behind the redone solve
        return PSFM_OK;
    };

    int f = 0;
    while (f < n_flows) {
        const bool fused_now = c->solver_mode == 2 || (c->solver_mode == 0 && c->solve_mode == 0);
        if (seq_ok && fused_now && f >= 1) {
            // ---- a window of the device-paced sequence: launches that each do "the next thing" (psfm_seq_kernel) ----
            const int k_now = c->solver_K > 0 ? c->solver_K : c->solve_K;
            if (!pc_in_step) {       // behind a host-paced window or a redo: put the device's counter where the host is
                hpc[0] = f; hpc[1] = 0; hpc[2] = launch_id; hpc[3] = k_now;
                PSFM_HIP(hipMemcpyAsync(&dctr->pc_frame, hpc, 4 * sizeof(int), hipMemcpyHostToDevice, s));
                pc_in_step = true;
            } else {
                hpc[3] = k_now;
                PSFM_HIP(hipMemcpyAsync(&dctr->solve_K, hpc + 3, sizeof(int), hipMemcpyHostToDevice, s));
            }
            const int left = n_flows - f;
            const int n_launch = (left < PSFM_CHECK ? left : PSFM_CHECK) + 2;   // two spare: continuation launches of the window
            const int f_hi = f + n_launch - 1 < n_flows - 1 ? f + n_launch - 1 : n_flows - 1;   // the furthest the device can get
            if (pipe && (st = pipe->need(f_hi, false, s)) != PSFM_OK) return st;
            if (pipe && (st = pipe->need(f_hi - 1, true, s)) != PSFM_OK) return st;
            if ((st = psfm_launch_seq(c, d, flows, occ, occ_pitch, flows_f2, occ_s2, n_launch, launch_id, s)) != PSFM_OK) return st;
            launch_id += n_launch;
            const int f_before = f;
            if ((st = checkpoint(f_hi, true, true, &f)) != PSFM_OK) return st;
            if (f == f_before) {
                // No frame completed and no solve stalled: the device is INSIDE the solve of frame f -- every iteration so far accepted,
                // none terminating, the window's launches used up (a sequence's last frame has three: K + 2 + 2 iterations).  Its chain
                // step has run: the next window's launches go on with the solve (at PC_KMAX accepted iterations it stalls and is redone
                // above).  Running the frame from the top here -- what rounds 2-4 did, "never expected" -- gave birth to the frame's
                // newborns twice (found by scripts/stress_batch.py in round 5: a two-flow sequence whose one solve takes 8 iterations).
                if (getenv("PSFM_TRACE")) {
                    const PsfmCounters* hc = (const PsfmCounters*)c->host_pinned;
                    unsigned tk[8] = {0};
                    if (c->sol_fused.p) (void)hipMemcpy(tk, c->sol_fused.p, sizeof(tk), hipMemcpyDeviceToHost);
                    fprintf(stderr, "[psfm %p] device-paced window without a completed frame: host frame %d, next launch %d; device pc {frame %d, phase %d, "
                            "owner %d, K %d}, lanes %d (snapshots %d %d), stall %d, overflow %d, tickets %u %u %u %u\n", (void*)c, f, launch_id,
                            hc->pc_frame, hc->pc_phase, hc->pc_owner, hc->solve_K, hc->n_lanes, hc->n_lanes_snap[0], hc->n_lanes_snap[1], hc->stall,
                            hc->overflow, tk[0], tk[1], tk[2], tk[3]);
                }
                if (++idle_windows > 8) { psfm_set_error("psfm_track: the device-paced sequence made no progress in 8 windows (frame %d)", f); return PSFM_ERR_SOLVER; }
            } else {
                idle_windows = 0;
            }
            continue;
        }
        pc_in_step = false;         // a host-paced frame: the device-side counter is not maintained
        // track.py:31-47 / track_optimize.py:31-50, one loop iteration:
        // one launch = births of frame f (new_traj_all) + chain step f (step_forward, extend_all)
        if (pipe && (st = pipe->need(f, false, s)) != PSFM_OK) return st;
        const bool solve_now = optimize && f + 1 >= 2;   // track_optimize.py:49-50
        if (solve_now && fused_now && merge) {
            // ONE launch: chain step of the frame + the fused solve of its tracks
            if (pipe && (st = pipe->need(f - 1, true, s)) != PSFM_OK) return st;
            st = psfm_launch_frame(c, d, flows + (size_t)(f - 1) * P * 2, flows + (size_t)f * P * 2, flows_f2 + (size_t)(f - 1) * P * 2,
                                   occ + (size_t)f * occ_pitch, occ_s2 + (size_t)(f - 1) * P, f, c->solver_K > 0 ? c->solver_K : c->solve_K, s);
            if (st != PSFM_OK) return st;
        } else {
            st = psfm_launch_chain_step(c, d, flows + (size_t)f * P * 2, occ + (size_t)f * occ_pitch, f, optimize, s);
            if (st != PSFM_OK) return st;
            if (solve_now) {
                if (pipe && (st = pipe->need(f - 1, true, s)) != PSFM_OK) return st;
                if (fused_now) {   // (timed inside: kernel begin / end events)
                    st = psfm_solve_frame_fused(c, d, flows + (size_t)(f - 1) * P * 2, flows + (size_t)f * P * 2,
                                                flows_f2 + (size_t)(f - 1) * P * 2, occ_s2 + (size_t)(f - 1) * P, f,
                                                c->solver_K > 0 ? c->solver_K : c->solve_K, s);
                } else {
                    c->prof.begin(PSFM_PROF_SOLVER, s);
                    st = psfm_solve_frame_enqueue(c, d, flows + (size_t)(f - 1) * P * 2, flows + (size_t)f * P * 2,
                                                  flows_f2 + (size_t)(f - 1) * P * 2, occ_s2 + (size_t)(f - 1) * P, f,
                                                  unroll_fixed > 0 ? unroll_fixed : c->solve_unroll, s);
                    c->prof.end(s);
                }
                if (st != PSFM_OK) return st;
            }
        }
        if (optimize && f >= 1 && ((f % PSFM_CHECK) == PSFM_CHECK - 1 || f == n_flows - 1)) {
            if ((st = checkpoint(f, fused_now, false, &f)) != PSFM_OK) return st;
            continue;
        }
        ++f;
    }
    // the positions of the last fused solve are still in their iterate buffer (no chain step followed to move them)
    if (optimize && n_flows >= 2 && (st = psfm_solve_flush(c, d, n_flows - 1, s)) != PSFM_OK) return st;
    c->prof.begin(PSFM_PROF_FINALIZE, s);
    st = psfm_finalize(c, d, s);
    c->prof.end(s);
    if (st != PSFM_OK) return st;
    PSFM_HIP(hipStreamSynchronize(s));
    c->prof.collect();
    if (info) {
        memset(info, 0, sizeof(*info));
        info->n_traj = c->res_n_traj;
        info->n_points = c->res_n_points;
        info->n_lanes_peak = ((PsfmCounters*)c->host_pinned)->n_lanes;
        info->lane_capacity = d.cap;
        info->solver_iterations = total_iters;
        info->n_solves = (int32_t)c->solve_stats.size();
        info->chain_mode = 1;
    }
    return PSFM_OK;
}

extern "C" psfm_status psfm_track(psfm_ctx* c, const float* flows, const uint8_t* occ, const float* flows_f2,
                                  const uint8_t* occ_s2, int n_flows, int h, int w, int ratio, psfm_track_info* info,
                                  void* stream)
{
    PSFM_CHECK_CTX(c);   // (selects the context's device: the residency query below is per device)
    PsfmGate gate(c->device, psfm_wants_persist(c, flows_f2 != nullptr, h, w, ratio, false));
    return psfm_track_impl(c, flows, occ, flows_f2, occ_s2, n_flows, h, w, ratio, info, stream, nullptr, gate);
}

// The compute part of the stage entry (main_connect_point_trajectories.py:36-53): flow_check of the stride-1 (and
// stride-2) stacks + track / track_optimize.  flow_check is bandwidth-bound, the frame loop is latency-bound: the
// maps are produced in chunks on a side stream and the loop only waits for the chunk it is about to read, so most
// of flow_check's time disappears behind the recurrence.
extern "C" psfm_status psfm_connect(psfm_ctx* c, const float* flows_f, const float* flows_b, const float* flows_f2,
                                    const float* flows_b2, int n_flows, int h, int w, float thres, int ratio,
                                    uint8_t* occ, uint8_t* occ_s2, psfm_track_info* info, void* stream)
{
    PSFM_CHECK_CTX(c);
    const bool optimize = flows_f2 != nullptr;
    PsfmGate gate(c->device, psfm_wants_persist(c, optimize, h, w, ratio, true));
    if (n_flows < 1 || !psfm_frame_ok(h, w) || !flows_f || !flows_b || (optimize && n_flows > 1 && !flows_b2)) {
        psfm_set_error("psfm_connect: bad argument (n_flows=%d h=%d w=%d)", n_flows, h, w);
        return PSFM_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t P = (size_t)h * w;
    psfm_status st;
    // Track mode with the device to ourselves: ONE persistent launch computes the occlusion maps AND runs the recurrence
    // (the blocks check flow consistency in the time they would otherwise wait at the frame barriers).  The maps are
    // written through to HBM by their producers and first read by other XCDs a few barriers later; a cache line must
    // not straddle two maps, hence the 128-byte pitch (caller-provided buffers qualify when H*W is a multiple of 128
    // and the buffer is 128-byte aligned).
    if (gate.exclusive && !optimize && (!occ || (P % 128 == 0 && ((uintptr_t)occ & 127) == 0))) {
        int64_t pitch = (int64_t)P;
        if (!occ) {
            pitch = (int64_t)((P + 127) / 128 * 128);
            if ((st = c->occ_own.ensure((size_t)pitch * (size_t)n_flows)) != PSFM_OK) return st;
            occ = c->occ_own.as<uint8_t>();
        }
        return psfm_track_impl(c, flows_f, occ, nullptr, nullptr, n_flows, h, w, ratio, info, stream, nullptr, gate, flows_b,
                               thres, pitch);
    }
    if (!occ) {
        if ((st = c->occ_own.ensure(P * (size_t)n_flows)) != PSFM_OK) return st;
        occ = c->occ_own.as<uint8_t>();
    }
    const int n2 = optimize ? (n_flows > 1 ? n_flows - 1 : 0) : 0;
    if (optimize && !occ_s2) {
        if ((st = c->occ2_own.ensure(P * (size_t)(n2 > 0 ? n2 : 1))) != PSFM_OK) return st;
        occ_s2 = c->occ2_own.as<uint8_t>();
    }
    // (a lowest-priority side stream was measured: no gain -- 11.05 vs 10.95 ms per 1080p track_optimize sequence)
    if (!c->side_stream) PSFM_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    hipStream_t side = c->side_stream;
    // the side stream starts after whatever the caller enqueued on `stream` (the inputs)
    hipEvent_t e_in = c->prof.get();
    PSFM_HIP(hipEventRecord(e_in, s));
    PSFM_HIP(hipStreamWaitEvent(side, e_in, 0));
    PsfmOccPipeline pipe;
    // pairs per side-stream launch: 10 beside the chain step of track mode; 5 beside track_optimize's frame kernel (round 3:
    // 8.25 ms per 1080p sequence against 8.28-8.40 with 10 and 8.31-8.44 with 20)
    pipe.chunk = getenv("PSFM_FC_CHUNK") && atoi(getenv("PSFM_FC_CHUNK")) > 0 ? atoi(getenv("PSFM_FC_CHUNK")) : (optimize ? 5 : 10);
    c->prof.begin(PSFM_PROF_FLOW_CHECK, side);
    // track_optimize: every chunk as a full-occupancy launch of the stand-alone kernel.  (Round 2 ran all but the first chunk as a
    // BACKGROUND kernel shaped to fit beside the frame kernel's three 160-VGPR waves per SIMD -- psfm_flow_check_bg_kernel, still
    // there behind PSFM_FC_BG=1.  Since round 3 the frame kernel runs four 128-VGPR waves per SIMD, which leaves that kernel no
    // registers to live in, and the stand-alone kernel got 15 % faster: 8.58-8.64 ms per 1080p sequence with the background form,
    // 8.28-8.40 without.)
    const bool bg = optimize && getenv("PSFM_FC_BG") && atoi(getenv("PSFM_FC_BG")) == 1;
    int cus = 256;
    if (bg) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
        const int mult = getenv("PSFM_FC_BG_BLOCKS") ? atoi(getenv("PSFM_FC_BG_BLOCKS")) : 0;   // resident blocks per CU; 0: short-lived blocks
        cus *= mult > 0 ? mult : 0;
    }
    for (int p0 = 0; p0 < n_flows; p0 += pipe.chunk) {
        const int np = n_flows - p0 < pipe.chunk ? n_flows - p0 : pipe.chunk;
        if (bg && p0 > 0) st = psfm_launch_flow_check_bg(flows_f + (size_t)p0 * P * 2, flows_b + (size_t)p0 * P * 2, np, h, w, thres,
                                                        occ + (size_t)p0 * P, cus, side);
        else st = psfm_launch_flow_check(flows_f + (size_t)p0 * P * 2, flows_b + (size_t)p0 * P * 2, np, h, w, thres,
                                         occ + (size_t)p0 * P, nullptr, side);
        if (st != PSFM_OK) return st;
        hipEvent_t e = c->prof.get();
        PSFM_HIP(hipEventRecord(e, side));
        pipe.ready.push_back(e);
        // the stride-2 maps of the same time range follow their stride-1 chunk (needed one frame later)
        if (p0 < n2) {
            const int np2 = n2 - p0 < pipe.chunk ? n2 - p0 : pipe.chunk;
            if (bg && p0 > 0) st = psfm_launch_flow_check_bg(flows_f2 + (size_t)p0 * P * 2, flows_b2 + (size_t)p0 * P * 2, np2, h, w, thres,
                                                            occ_s2 + (size_t)p0 * P, cus, side);
            else st = psfm_launch_flow_check(flows_f2 + (size_t)p0 * P * 2, flows_b2 + (size_t)p0 * P * 2, np2, h, w,
                                             thres, occ_s2 + (size_t)p0 * P, nullptr, side);
            if (st != PSFM_OK) return st;
            hipEvent_t e2 = c->prof.get();
            PSFM_HIP(hipEventRecord(e2, side));
            pipe.ready_s2.push_back(e2);
        }
    }
    c->prof.end(side);
    st = psfm_track_impl(c, flows_f, occ, flows_f2, occ_s2, n_flows, h, w, ratio, info, stream, &pipe, gate);
    // psfm_track_impl synchronised `stream`, which waited on every chunk: the side stream is idle too
    c->prof.pool.push_back(e_in);
    for (auto e : pipe.ready) c->prof.pool.push_back(e);
    for (auto e : pipe.ready_s2) c->prof.pool.push_back(e);
    return st;
}

extern "C" psfm_status psfm_result_device(psfm_ctx* c, const int32_t** birth, const int32_t** len, const int64_t** off,
                                          const double** xy)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    if (birth) *birth = c->res_birth.as<int32_t>();
    if (len) *len = c->res_len.as<int32_t>();
    if (off) *off = c->res_off.as<int64_t>();
    if (xy) *xy = c->res_xy.as<double>();
    return PSFM_OK;
}

extern "C" psfm_status psfm_result_copy(psfm_ctx* c, int32_t* birth_host, int32_t* len_host, int64_t* off_host,
                                        double* xy_host, void* stream)
{
    PSFM_CHECK_CTX(c);
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = c->res_n_traj, np = c->res_n_points;
    if (n > 0) {
        if (birth_host) PSFM_HIP(hipMemcpyAsync(birth_host, c->res_birth.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
        if (len_host) PSFM_HIP(hipMemcpyAsync(len_host, c->res_len.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
        if (off_host) PSFM_HIP(hipMemcpyAsync(off_host, c->res_off.p, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, s));
        if (xy_host && np > 0) PSFM_HIP(hipMemcpyAsync(xy_host, c->res_xy.p, sizeof(double) * 2 * np, hipMemcpyDeviceToHost, s));
    } else if (off_host) {
        off_host[0] = 0;
    }
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}

extern "C" psfm_status psfm_result_solve_stats(psfm_ctx* c, psfm_solve_stats* stats_host, int32_t max_n, int32_t* n_out)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    const int32_t n = (int32_t)c->solve_stats.size();
    if (n_out) *n_out = n;
    if (stats_host) for (int32_t i = 0; i < n && i < max_n; ++i) stats_host[i] = c->solve_stats[i];
    return PSFM_OK;
}
