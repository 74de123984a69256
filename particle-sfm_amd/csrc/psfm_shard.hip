// psfm_shard.hip -- ONE sequence over several processes / GPUs, exactly (SURVEY.md 8e Stage B; the driver is
// particle-sfm_amd/psfm_dist.connect_sharded, the Python engine point_trajectory/shard.py).
//
// The frame recurrence (track.py:31-47 / track_optimize.py:31-50) cannot be cut into frame ranges -- births at t+1 need
// every survivor of t -- but its TRACKS split exactly: this process owns the tracks born on grid points [g0, g1) (whole
// rows of the stride-r grid) and runs every frame for them with the same kernels as a single-process run:
//   psfm_shard_step            chain step + births of the own band (psfm_chain_step_kernel with the band offset).  The
//                              stamped grid-resolution `blocked` map of the frame (+ the survivor byte at offset G) sits in
//                              a caller-owned buffer: the ranks all-reduce(max) its G + 1 bytes before the next step reads it
//   psfm_shard_solve_export    the solve of the frame for the own tracks, sums only (fused: K x 13; chain: 13)
//   psfm_shard_frame           the two above (fused export) as one launch
//   psfm_shard_solve_control   Ceres' control step on the totals over all ranks (replicated: same numbers on every rank)
//   psfm_shard_finish          write-backs pending, finalize: the own trajectories as the usual CSR result, sorted by the
//                              key (last valid time, birth frame, birth grid index) -- global ids follow from the keys
// No collective is issued from here: the caller owns the process group (RCCL over xGMI, or gloo in tests).
#include <stdlib.h>
#include <string.h>

#include "psfm_internal.h"

#define PSFM_SHARD_CHECK(c)                                                                                  \
    do {                                                                                                     \
        if (!(c) || !(c)->shard_dims) { psfm_set_error("psfm_shard_*: no sharded run in progress"); return PSFM_ERR_ARG; } \
        PSFM_HIP(hipSetDevice((c)->device));                                                                 \
    } while (0)

#define PSFM_KEY_TIME_BITS 16
#define PSFM_KEY_GRID_BITS 31

void psfm_shard_abandon(psfm_ctx* c)
{
    delete c->shard_dims;
    c->shard_dims = nullptr;
}

extern "C" psfm_status psfm_shard_begin(psfm_ctx* c, int n_flows, int h, int w, int ratio, int64_t g0, int64_t g1, int optimize,
                                        uint8_t* maps, int64_t map_pitch, void* stream)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    if (n_flows < 1 || !psfm_frame_ok(h, w) || ratio < 1 || ratio > 64 || !maps) {
        psfm_set_error("psfm_shard_begin: bad argument (n_flows=%d h=%d w=%d ratio=%d)", n_flows, h, w, ratio);
        return PSFM_ERR_ARG;
    }
    // the ids over ranks come from the key last << 47 | birth << 31 | grid index (psfm_result_keys): 16 bits per time field, 31 for
    // the grid index (a grid has fewer than 2^30 points: psfm_track_dims)
    if (n_flows + 2 >= (1 << PSFM_KEY_TIME_BITS)) {
        psfm_set_error("psfm_shard_begin: %d flows: the (last, birth, grid) key of a sharded run holds times below %d", n_flows,
                       (1 << PSFM_KEY_TIME_BITS) - 2);
        return PSFM_ERR_ARG;
    }
    PsfmTrackDims d;
    psfm_status st;
    const int64_t G = (int64_t)((w + ratio - 1) / ratio) * ((h + ratio - 1) / ratio);
    if (G >= ((int64_t)1 << PSFM_KEY_GRID_BITS)) { psfm_set_error("psfm_shard_begin: %lld grid points", (long long)G); return PSFM_ERR_ARG; }
    if (g0 < 0 || g1 < g0 || g1 > G || map_pitch < G + 1) {
        psfm_set_error("psfm_shard_begin: band [%lld, %lld) of %lld grid points, map pitch %lld", (long long)g0, (long long)g1,
                       (long long)G, (long long)map_pitch);
        return PSFM_ERR_ARG;
    }
    if ((st = psfm_track_dims(c, n_flows, h, w, ratio, g1 - g0, d)) != PSFM_OK) return st;
    d.g0 = g0; d.Gband = g1 - g0; d.shard_maps = maps; d.shard_pitch = map_pitch;
    if ((st = psfm_track_alloc(c, d)) != PSFM_OK) return st;
    if (optimize && (st = psfm_solve_prepare(c, d, (hipStream_t)stream)) != PSFM_OK) return st;
    c->solve_stats.clear();
    c->res_n_traj = c->res_n_points = 0;
    c->res_n_flows = n_flows;
    *(int32_t*)((char*)c->host_pinned + c->host_pinned_bytes - 64) = 0;
    delete c->shard_dims;
    c->shard_dims = new PsfmTrackDims(d);
    c->shard_optimize = optimize != 0;
    // (psfm_shard_solve_local / _redo_local: a resident launch only inside a budget -- the calls of a sharded run come and go under
    // the shared gate, nothing holds the device for the sequence)
    c->pc_persist_ok = c->resident_budget > 0;
    c->pc_giveups = 0;
    c->n_resident = c->n_iter_launches = 0;
    return psfm_launch_track_init(c, d, (hipStream_t)stream);
}

extern "C" psfm_status psfm_shard_step(psfm_ctx* c, const float* flow, const uint8_t* occ, int frame, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (!flow || !occ || frame < 0 || frame >= d.n_flows) { psfm_set_error("psfm_shard_step: bad argument (frame %d)", frame); return PSFM_ERR_ARG; }
    return psfm_launch_chain_step(c, d, flow, occ, frame, c->shard_optimize, (hipStream_t)stream);
}

extern "C" psfm_status psfm_shard_solve_export(psfm_ctx* c, const float* flow01, const float* flow12, const float* flow02,
                                               const uint8_t* occ02, int frame, int kind, int k, double* sums_out, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    if (kind < 0 || kind > 2 || !sums_out || frame < 1) { psfm_set_error("psfm_shard_solve_export: bad argument"); return PSFM_ERR_ARG; }
    return psfm_solve_export(c, *c->shard_dims, flow01, flow12, flow02, occ02, frame, kind, k, sums_out, (hipStream_t)stream);
}

// Chain step of `frame` AND the fused export of its solve as ONE launch (psfm_frame_kernel with the sums exported): what
// psfm_shard_step + psfm_shard_solve_export(kind 0) do in two, without the round trip of the tracks' tails through the log
// in between.  The solve of frame t only needs this process's own tracks, so it does not wait for the exchange of the
// frame's marks; the caller all-reduces the map and combines the sums behind the launch, then runs the control step.
extern "C" psfm_status psfm_shard_frame(psfm_ctx* c, const float* flow01, const float* flow12, const float* flow02, const uint8_t* occ,
                                        const uint8_t* occ02, int frame, int k, double* sums_out, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (!flow01 || !flow12 || !flow02 || !occ || !occ02 || frame < 1 || frame >= d.n_flows || !c->shard_optimize) {
        psfm_set_error("psfm_shard_frame: bad argument (frame %d)", frame);
        return PSFM_ERR_ARG;
    }
    if (!sums_out) {
        // a shard that is the whole sequence: nothing to exchange, the launch runs the control step on its own totals (the frame
        // kernel as psfm_connect's host-paced windows launch it) -- no export, no control launch behind every frame.  A solve that
        // does not go as speculated raises the device-side stall flag; every 8th frame carries it to where psfm_shard_peek_stall looks
        if (d.g0 != 0 || d.Gband != d.G) {
            psfm_set_error("psfm_shard_frame: sums_out is NULL for a band [%lld, %lld) of %lld grid points (the whole grid only)",
                           (long long)d.g0, (long long)(d.g0 + d.Gband), (long long)d.G);
            return PSFM_ERR_ARG;
        }
        psfm_status st = psfm_launch_frame(c, d, flow01, flow12, flow02, occ, occ02, frame, k, (hipStream_t)stream, nullptr);
        if (st != PSFM_OK) return st;
        if ((frame & 7) == 7 || frame == d.n_flows - 1)
            PSFM_HIP(hipMemcpyAsync((char*)c->host_pinned + c->host_pinned_bytes - 64, &c->counters.as<PsfmCounters>()->stall, sizeof(int32_t),
                                    hipMemcpyDeviceToHost, (hipStream_t)stream));
        return PSFM_OK;
    }
    return psfm_launch_frame(c, d, flow01, flow12, flow02, occ, occ02, frame, k, (hipStream_t)stream, sums_out);
}

extern "C" psfm_status psfm_shard_solve_control(psfm_ctx* c, int frame, int kind, int k, const double* totals, int32_t* done_host,
                                                int32_t* redo_host, psfm_solve_stats* stats_host, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    psfm_status st = psfm_solve_control(c, *c->shard_dims, frame, kind, k, totals, s);
    if (st != PSFM_OK) return st;
    int done = 0, stall = 0;
    psfm_solve_stats ss;
    memset(&ss, 0, sizeof(ss));
    if ((st = psfm_solve_state(c, &done, &stall, &ss, s)) != PSFM_OK) return st;
    if (done_host) *done_host = done;
    if (redo_host) *redo_host = stall != 0;     // the fused solve met something it had not speculated: redo with the chain
    if (stats_host) *stats_host = ss;
    return PSFM_OK;
}

// The control step of a fused export WITHOUT reading anything back: the caller goes on enqueuing the next frames and looks at
// a whole window of solves at once (psfm_shard_window_state).  A solve that did not go as speculated raises the device-side
// stall flag, which turns every later launch of this context into a no-op until psfm_shard_solve_restore.
extern "C" psfm_status psfm_shard_solve_control_async(psfm_ctx* c, int frame, int k, const double* totals, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    // the stall flag follows the control step into pinned memory (the control kernel stores it there itself):
    // psfm_shard_peek_stall reads it without a synchronisation
    int32_t* peek = (int32_t*)((char*)c->host_pinned + c->host_pinned_bytes - 64);
#ifdef PSFM_SHARD_STALL_MEMCPY      // (A/B builds: round 3's form)
    psfm_status st = psfm_solve_control(c, *c->shard_dims, frame, 0, k, totals, (hipStream_t)stream);
    if (st != PSFM_OK) return st;
    PSFM_HIP(hipMemcpyAsync(peek, &c->counters.as<PsfmCounters>()->stall, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PSFM_OK;
#else
    return psfm_solve_control(c, *c->shard_dims, frame, 0, k, totals, (hipStream_t)stream, (int*)peek);
#endif
}

// The redo of a solve that did not go as speculated, WITHOUT a host round trip per trust-region iteration: the control step behind
// an export of kind 1 (iteration 0) / 2 (one iteration) is only enqueued; the caller enqueues a batch of rounds (export -> its
// exchange -> this) ahead and asks once per batch (psfm_shard_solve_poll).  Rounds behind the one that terminated the solve find
// the control block done: their export and their control step return at once (the exchange in between carries stale sums nobody
// reads).  Every rank enqueues the same batches -- the decision is made from the same totals everywhere.
extern "C" psfm_status psfm_shard_solve_control_chain_async(psfm_ctx* c, int frame, int kind, const double* totals, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    if (kind != 1 && kind != 2) { psfm_set_error("psfm_shard_solve_control_chain_async: kind must be 1 or 2"); return PSFM_ERR_ARG; }
    return psfm_solve_control(c, *c->shard_dims, frame, kind, 1, totals, (hipStream_t)stream);
}

// Synchronises: is the solve the chain is working on done, and its statistics so far.
extern "C" psfm_status psfm_shard_solve_poll(psfm_ctx* c, int32_t* done_host, psfm_solve_stats* stats_host, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    int done = 0, stall = 0;
    psfm_solve_stats ss;
    memset(&ss, 0, sizeof(ss));
    psfm_status st = psfm_solve_state(c, &done, &stall, &ss, (hipStream_t)stream);
    if (st != PSFM_OK) return st;
    if (done_host) *done_host = done;
    if (stats_host) *stats_host = ss;
    return PSFM_OK;
}

// ---- a shard that is the WHOLE sequence (g0 = 0, g1 = G: one rank -- the windowed engine used for one long sequence on one GPU).
// Nothing of its solves has to be exchanged, so a solve whose steps get rejected does not have to go round export -> exchange ->
// control once per trust-region iteration: it runs like the one-GPU call's -- the resident solve (one launch per solve, the context's
// resident budget permitting: psfm_ctx_set_resident_budget) or the launch chain with its own reduction, and the same write-back.  The
// sums are added in the same order either way (psfm_pc_resident.h: pc_tree_totals), so the positions are the ones the exchange
// form gives.  Refused for a band of the grid (PSFM_ERR_ARG): a band's totals are not the sequence's.
static bool shard_is_whole(const PsfmTrackDims& d) { return d.g0 == 0 && d.Gband == d.G; }

// ENQUEUES the solve of `frame` (behind psfm_shard_step(frame)) -- what psfm_connect does for a window whose solves reject steps; a
// solve that is not done behind its launches raises the device-side stall flag like a fused one (psfm_shard_window_state reports it;
// it does not reach psfm_shard_peek_stall, which follows the control steps of fused solves only)
extern "C" psfm_status psfm_shard_solve_local(psfm_ctx* c, const float* flow01, const float* flow12, const float* flow02,
                                              const uint8_t* occ02, int frame, int unroll, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (!shard_is_whole(d) || !c->shard_optimize || !flow01 || !flow12 || !flow02 || !occ02 || frame < 1 || frame >= d.n_flows) {
        psfm_set_error("psfm_shard_solve_local: frame %d of a shard [%lld, %lld) of %lld grid points (the whole grid only)", frame,
                       (long long)d.g0, (long long)(d.g0 + d.Gband), (long long)d.G);
        return PSFM_ERR_ARG;
    }
    return psfm_solve_frame_enqueue(c, d, flow01, flow12, flow02, occ02, frame, unroll < 1 ? 1 : (unroll > 64 ? 64 : unroll),
                                    (hipStream_t)stream);
}

// The stalled solve of `frame` redone to termination and written back (synchronises).  chain_stalled: it had been enqueued by
// psfm_shard_solve_local (its resident launch gave up / its launches did not suffice), not by a fused export
extern "C" psfm_status psfm_shard_solve_redo_local(psfm_ctx* c, const float* flow01, const float* flow12, const float* flow02,
                                                   const uint8_t* occ02, int frame, int chain_stalled, psfm_solve_stats* stats_host,
                                                   void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (!shard_is_whole(d) || !c->shard_optimize || !flow01 || !flow12 || !flow02 || !occ02 || frame < 1 || frame >= d.n_flows) {
        psfm_set_error("psfm_shard_solve_redo_local: frame %d of a shard [%lld, %lld) of %lld grid points (the whole grid only)", frame,
                       (long long)d.g0, (long long)(d.g0 + d.Gband), (long long)d.G);
        return PSFM_ERR_ARG;
    }
    *(int32_t*)((char*)c->host_pinned + c->host_pinned_bytes - 64) = 0;     // (the caller has synchronised: nothing is in flight)
    psfm_solve_stats ss;
    memset(&ss, 0, sizeof(ss));
    psfm_status st = psfm_solve_frame_resume(c, d, flow01, flow12, flow02, occ02, frame, &ss, 0, chain_stalled != 0, (hipStream_t)stream);
    if (st != PSFM_OK) return st;
    if (ss.termination >= 0) c->solve_stats.push_back(ss);
    if (stats_host) *stats_host = ss;
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// ONE solve over several ranks WITHOUT the host or a collective library in its loop (round 6).  A window whose solves reject steps
// costs the exchange form one export launch + one all-gather + one control launch per trust-region iteration and a host poll every
// few of them (3x the one-GPU call on a hard 1080p sequence).  Here every rank runs the RESIDENT solve on its own tracks and the
// second hop of its all-reduce crosses the ranks: leaders write their sums into every rank's granule area through peer-mapped
// pointers (csrc/psfm_solver.hip: PcPeers).  Set-up, once per engine:
//     psfm_shard_peer_area      this rank's area (allocated once per context, zeroed) + its IPC handle
//     psfm_shard_peer_open      another PROCESS's area from its handle (hipIpcOpenMemHandle; threads of one process pass pointers)
//     psfm_shard_peer_connect   world, rank, every rank's area as this process addresses it, every rank's launch size
// and per frame psfm_shard_solve_peer(frame, epoch) where psfm_shard_solve_local would run on one rank.  All ranks pass the same
// epoch (a counter they advance together): it tags every granule of the solve.
// ------------------------------------------------------------------------------------------------
extern "C" psfm_status psfm_shard_peer_area(psfm_ctx* c, void** area_dev, void* ipc_handle_64, void* stream)
{
    if (!c || !area_dev) { psfm_set_error("psfm_shard_peer_area: bad argument"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    const bool fresh = c->peer_area.p == nullptr;
    if (fresh) {
        // FINE-GRAINED device memory: other GPUs write into this area over xGMI while this GPU's launch polls it.  Coarse-grained memory
        // is only guaranteed coherent across agents at kernel boundaries; fine-grained allocations are not held in the L2s.  (Within one
        // device the granules work on either kind -- the one-GPU launch keeps its rows in a plain allocation.)  PSFM_PEER_COARSE=1: plain
        // hipMalloc (A/B runs); also the fall-back where the extension is refused.
        void* q = nullptr;
        const bool coarse = getenv("PSFM_PEER_COARSE") && atoi(getenv("PSFM_PEER_COARSE")) != 0;
        if (!coarse && hipExtMallocWithFlags(&q, psfm_peer_area_bytes(), hipDeviceMallocFinegrained) == hipSuccess && q) {
            c->peer_area.p = q;
            c->peer_area.bytes = psfm_peer_area_bytes();
        } else {
            (void)hipGetLastError();
        }
    }
    psfm_status st = c->peer_area.ensure(psfm_peer_area_bytes());
    if (st != PSFM_OK) return st;
    if (fresh) {
        PSFM_HIP(hipMemsetAsync(c->peer_area.p, 0, psfm_peer_area_bytes(), (hipStream_t)stream));
        PSFM_HIP(hipStreamSynchronize((hipStream_t)stream));
    }
    *area_dev = c->peer_area.p;
    if (ipc_handle_64) {
        hipIpcMemHandle_t h;
        PSFM_HIP(hipIpcGetMemHandle(&h, c->peer_area.p));
        memcpy(ipc_handle_64, &h, sizeof(h));
    }
    return PSFM_OK;
}

// The last epoch this context's area has seen (every launch of psfm_shard_solve_peer records its epoch).  Engines take the maximum
// over the ranks and go on from there; reset != 0 (every rank, between two collectives, nothing in flight: the epochs are about to
// wrap around their 20 bits): the area is zeroed and the count starts over.
extern "C" psfm_status psfm_shard_peer_epoch(psfm_ctx* c, uint32_t* last_epoch, int reset, void* stream)
{
    if (!c || !last_epoch) { psfm_set_error("psfm_shard_peer_epoch: bad argument"); return PSFM_ERR_ARG; }
    if (reset && c->peer_area.p) {
        PSFM_HIP(hipSetDevice(c->device));
        PSFM_HIP(hipStreamSynchronize((hipStream_t)stream));
        PSFM_HIP(hipMemsetAsync(c->peer_area.p, 0, psfm_peer_area_bytes(), (hipStream_t)stream));
        PSFM_HIP(hipStreamSynchronize((hipStream_t)stream));
        c->peer_epoch = 0;
    }
    *last_epoch = c->peer_epoch;
    return PSFM_OK;
}

extern "C" psfm_status psfm_shard_peer_open(psfm_ctx* c, const void* ipc_handle_64, int peer_rank, void** mapped)
{
    if (!c || !ipc_handle_64 || !mapped || peer_rank < 0 || peer_rank >= PSFM_MAX_PEERS) { psfm_set_error("psfm_shard_peer_open: bad argument"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    if (c->peer_opened[peer_rank]) { (void)hipIpcCloseMemHandle(c->peer_opened[peer_rank]); c->peer_opened[peer_rank] = nullptr; }
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle_64, sizeof(h));
    void* q = nullptr;
    PSFM_HIP(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
    c->peer_opened[peer_rank] = q;
    *mapped = q;
    return PSFM_OK;
}

// n_blocks[r]: blocks of rank r's solver launches (psfm_shard_solve_blocks on that rank, after ITS psfm_shard_begin); 0 anywhere = that
// rank cannot run the resident form -> PSFM_ERR_ARG here on every rank (the caller keeps the exchange form)
extern "C" psfm_status psfm_shard_peer_connect(psfm_ctx* c, int world, int rank, void* const* areas, const int32_t* n_blocks)
{
    if (!c || world < 1 || world > PSFM_MAX_PEERS || rank < 0 || rank >= world || !areas || !n_blocks || !c->peer_area.p || areas[rank] != c->peer_area.p) {
        psfm_set_error("psfm_shard_peer_connect: bad argument (world %d of at most %d, rank %d)", world, PSFM_MAX_PEERS, rank);
        return PSFM_ERR_ARG;
    }
    for (int r = 0; r < world; ++r)
        if (!areas[r] || n_blocks[r] < 1) { psfm_set_error("psfm_shard_peer_connect: rank %d has no area / no resident launch", r); return PSFM_ERR_ARG; }
    c->peer_world = world; c->peer_rank = rank;
    for (int r = 0; r < world; ++r) {
        c->peer_lead[r] = (char*)areas[r] + psfm_peer_lead_offset();
        c->peer_L[r] = psfm_peer_leaders(n_blocks[r]);
    }
    return PSFM_OK;
}

// blocks of this rank's solver launches for the run psfm_shard_begin set up, or 0 when they cannot all be resident on the device
// (the context's resident budget, else the device's capacity)
extern "C" psfm_status psfm_shard_solve_blocks(psfm_ctx* c, int32_t* n_blocks)
{
    PSFM_SHARD_CHECK(c);
    if (!n_blocks) { psfm_set_error("psfm_shard_solve_blocks: bad argument"); return PSFM_ERR_ARG; }
    const int nb = psfm_solve_blocks(c, *c->shard_dims);
    const int room = c->resident_budget > 0 ? c->resident_budget : psfm_resident_blocks(c);
    *n_blocks = (c->shard_optimize && nb <= room) ? nb : 0;
    return PSFM_OK;
}

extern "C" psfm_status psfm_shard_solve_peer(psfm_ctx* c, const float* flow01, const float* flow12, const float* flow02,
                                             const uint8_t* occ02, int frame, uint32_t epoch, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (c->peer_world < 1 || !c->shard_optimize || !flow01 || !flow12 || !flow02 || !occ02 || frame < 1 || frame >= d.n_flows) {
        psfm_set_error("psfm_shard_solve_peer: frame %d, %d connected rank(s)", frame, c->peer_world);
        return PSFM_ERR_ARG;
    }
    if (epoch == 0 || epoch > 0xfffffu) { psfm_set_error("psfm_shard_solve_peer: epoch %u outside 1 .. 2^20 - 1", epoch); return PSFM_ERR_ARG; }
    c->peer_epoch = epoch;
    return psfm_solve_frame_enqueue_peer(c, d, flow01, flow12, flow02, occ02, frame, epoch, (hipStream_t)stream);
}

// The stall flag as of the last control step the device has COMPLETED (no synchronisation: the value lags the queue by the
// frames in flight): frame whose solve stalled, or -1.
extern "C" psfm_status psfm_shard_peek_stall(psfm_ctx* c, int32_t* stalled_frame)
{
    if (!c || !stalled_frame) { psfm_set_error("psfm_shard_peek_stall: NULL argument"); return PSFM_ERR_ARG; }
    const volatile int32_t* peek = (const volatile int32_t*)((char*)c->host_pinned + c->host_pinned_bytes - 64);
    const int32_t v = *peek;
    *stalled_frame = v ? v - 1 : -1;
    return PSFM_OK;
}

// Synchronises: the stalled frame (-1: none) and the statistics of the solves of frames [f_lo, f_hi] (those below the stalled
// frame are final; a frame whose solve had no track reports termination -1).
extern "C" psfm_status psfm_shard_window_state(psfm_ctx* c, int f_lo, int f_hi, psfm_solve_stats* stats_host, int32_t* stalled_frame,
                                               void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    const PsfmTrackDims& d = *c->shard_dims;
    if (f_lo < 1 || f_hi < f_lo || f_hi >= d.n_flows || !stats_host || !stalled_frame) {
        psfm_set_error("psfm_shard_window_state: bad argument (frames %d..%d)", f_lo, f_hi);
        return PSFM_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    PsfmCounters* hc = (PsfmCounters*)c->host_pinned;
    PSFM_HIP(hipMemcpyAsync(hc, c->counters.p, sizeof(PsfmCounters), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipMemcpyAsync(stats_host, c->sol_stats.as<psfm_solve_stats>() + f_lo, sizeof(psfm_solve_stats) * (size_t)(f_hi - f_lo + 1),
                            hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    *stalled_frame = hc->stall ? hc->stall - 1 : -1;
    return PSFM_OK;
}

extern "C" psfm_status psfm_shard_solve_restore(psfm_ctx* c, int frame, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    *(int32_t*)((char*)c->host_pinned + c->host_pinned_bytes - 64) = 0;     // (the caller has synchronised: nothing is in flight)
    return psfm_solve_restore(c, *c->shard_dims, frame, (hipStream_t)stream);
}

extern "C" psfm_status psfm_shard_solve_writeback(psfm_ctx* c, int frame, const psfm_solve_stats* stats, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    if (stats && stats->termination >= 0) c->solve_stats.push_back(*stats);
    return psfm_solve_writeback(c, *c->shard_dims, frame, (hipStream_t)stream);
}

// a fused solve that finished in its launch: nothing to copy (the next chain step / the final flush picks the iterate up)
extern "C" psfm_status psfm_shard_solve_record(psfm_ctx* c, const psfm_solve_stats* stats)
{
    if (!c || !stats) { psfm_set_error("psfm_shard_solve_record: NULL argument"); return PSFM_ERR_ARG; }
    if (stats->termination >= 0) c->solve_stats.push_back(*stats);
    return PSFM_OK;
}

extern "C" psfm_status psfm_shard_finish(psfm_ctx* c, psfm_track_info* info, void* stream)
{
    PSFM_SHARD_CHECK(c);
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    const PsfmTrackDims d = *c->shard_dims;
    delete c->shard_dims;
    c->shard_dims = nullptr;
    psfm_status st;
    if (c->shard_optimize && d.n_flows >= 2 && (st = psfm_solve_flush(c, d, d.n_flows - 1, s)) != PSFM_OK) return st;
    if ((st = psfm_finalize(c, d, s)) != PSFM_OK) return st;
    PSFM_HIP(hipStreamSynchronize(s));
    if (info) {
        memset(info, 0, sizeof(*info));
        info->n_traj = c->res_n_traj;
        info->n_points = c->res_n_points;
        info->n_lanes_peak = ((PsfmCounters*)c->host_pinned)->n_lanes;
        info->lane_capacity = d.cap;
        info->n_solves = (int32_t)c->solve_stats.size();
        for (auto& q : c->solve_stats) info->solver_iterations += q.iterations;
        info->chain_mode = 1;
    }
    return PSFM_OK;
}

// key (last valid time, birth frame, birth grid index) of every trajectory of the result, in result order (ascending):
// what psfm_dist.global_ids ranks over all processes.  Layout: last << 47 | birth << 31 | grid index (16 + 16 + 31 bits: sequences of
// up to 65533 flows -- round 3 gave the times 11 bits each and the grid 40, and refused sequences beyond 2045 flows for no reason).
__global__ __launch_bounds__(256) void psfm_shard_keys_kernel(const int* __restrict__ birth, const int* __restrict__ len,
                                                             const int64_t* __restrict__ off, const double2* __restrict__ xy, int64_t n,
                                                             int ratio, int GW, int64_t* __restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 p = xy[off[i]];
    const int64_t g = (int64_t)((int)p.y / ratio) * GW + (int)p.x / ratio;
    keys[i] = ((int64_t)(birth[i] + len[i] - 1) << (PSFM_KEY_GRID_BITS + PSFM_KEY_TIME_BITS)) | ((int64_t)birth[i] << PSFM_KEY_GRID_BITS) | g;
}

extern "C" psfm_status psfm_result_keys(psfm_ctx* c, int ratio, int w, int64_t* keys_dev, void* stream)
{
    if (!c || !keys_dev || ratio < 1 || w < 1) { psfm_set_error("psfm_result_keys: bad argument"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    const int64_t n = c->res_n_traj;
    if (c->res_n_flows + 2 >= (1 << PSFM_KEY_TIME_BITS)) {     // (the result of a plain psfm_track of a longer sequence)
        psfm_set_error("psfm_result_keys: the result spans %d flows: the packed key holds times below %d", c->res_n_flows, (1 << PSFM_KEY_TIME_BITS) - 2);
        return PSFM_ERR_ARG;
    }
    if (n > 0) {
        hipLaunchKernelGGL(psfm_shard_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           c->res_birth.as<int>(), c->res_len.as<int>(), c->res_off.as<int64_t>(), c->res_xy.as<double2>(), n, ratio,
                           (w + ratio - 1) / ratio, keys_dev);
        PSFM_HIP(hipGetLastError());
    }
    return PSFM_OK;
}
