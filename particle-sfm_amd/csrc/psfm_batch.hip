// psfm_batch.hip -- psfm_connect_batch: B same-shape sequences through ONE set of launches.
//
// The reference's driver walks a directory of sequences (run_particlesfm.py:168-176: connect_point_trajectory per sequence), and the
// sequences real data has are small: DAVIS 480x854 at sample_ratio 4 is 25 k grid points, Sintel 436x1024 at 2 is 112 k, ScanNet
// 640x480 dense 307 k (BASELINE configs[0] / [2] / [4]).  A frame of such a sequence is ONE dependent chain of memory round trips
// that occupies 100-1200 of the device's >= 2048 block slots for 8-45 us whatever its size: one sequence at a
// time leaves 75-97 % of the machine idle, and host threads + streams (point_trajectory.batch) buy back at most 2x
// (profiles/r05/r05_a_concurrent_small.txt) because every sequence still pays its own launches, checkpoints and finalize.
// Here blockIdx.y of every frame launch is the sequence:
//   * every sequence keeps its OWN context -- lane tables, log, counters, shard tables, solver buffers, result -- so a block only
//     needs its sequence's row of a small table in device memory (chain-step arguments of frame 1 + strides: the device-side
//     rebase of the device-paced sequence, psfm_chain_args_rebase / pc_params_rebase, turns them into any frame's);
//   * track mode: one psfm_chain_step_batch_kernel launch per frame index; track_optimize: psfm_seq_batch_kernel launches, every
//     sequence with its own device-side program counter and K -- sequences advance independently (one may continue a solve while
//     its neighbours take their next frame; a finished or stalled one turns its blocks into no-ops);
//   * ONE checkpoint (one packed copy, one synchronisation) per window of frames for all sequences, and ONE segmented finalize
//     (psfm_finalize_batch: one sort over the records of all sequences);
//   * a sequence whose solves reject steps (what the launch chain / the resident solve are for) is redone at the checkpoint like in
//     psfm_track; when a whole window of it is like that it leaves the batch and runs alone through psfm_connect afterwards.
// Results are those of psfm_connect, sequence by sequence (tests/test_gpu_batch.py: bit-identical to the oracle's).
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "psfm_internal.h"
#include "psfm_chain_step.h"

namespace {
struct SeqState {
    int f = 0;                   // first frame whose launch has not completed
    int first_unchecked = 1;     // first solve whose statistics have not been folded in
    bool resync = false;         // the device-side program counter must be set before the next window
    bool dropped = false;        // runs alone through psfm_connect behind the batch
    int k_dev = 3;               // solve_K as the device has it
    int64_t total_iters = 0;
    std::vector<psfm_solve_stats> hstats;
};
}  // namespace

static psfm_status batch_host_staging2(psfm_ctx* own, size_t need)
{
    if (own->host_batch2_bytes >= need) return PSFM_OK;
    if (own->host_batch2) (void)hipHostFree(own->host_batch2);
    own->host_batch2 = nullptr; own->host_batch2_bytes = 0;
    PSFM_HIP(hipHostMalloc(&own->host_batch2, need + 1024, hipHostMallocDefault));
    own->host_batch2_bytes = need + 1024;
    return PSFM_OK;
}

static psfm_status batch_host_staging(psfm_ctx* own, size_t need)
{
    if (own->host_batch_bytes >= need) return PSFM_OK;
    if (own->host_batch) (void)hipHostFree(own->host_batch);
    own->host_batch = nullptr; own->host_batch_bytes = 0;
    PSFM_HIP(hipHostMalloc(&own->host_batch, need + 4096, hipHostMallocDefault));
    own->host_batch_bytes = need + 4096;
    return PSFM_OK;
}

static psfm_status batch_run(psfm_ctx* const* ctxs, int n_seq, const float* const* flows_f, const float* const* flows_b,
                             const float* const* flows_f2, const float* const* flows_b2, const int* n_flows, int h, int w,
                             float thres, int ratio, psfm_track_info* infos, void* stream, bool trim, bool* grid_too_small);

extern "C" psfm_status psfm_connect_batch(psfm_ctx* const* ctxs, int n_seq, const float* const* flows_f, const float* const* flows_b,
                                          const float* const* flows_f2, const float* const* flows_b2, const int* n_flows, int h, int w,
                                          float thres, int ratio, psfm_track_info* infos, void* stream)
{
    if (!ctxs || n_seq < 1 || n_seq > PSFM_BATCH_MAX || !flows_f || !flows_b || !n_flows || !psfm_frame_ok(h, w) || ratio < 1 || ratio > 64 ||
        ((flows_f2 == nullptr) != (flows_b2 == nullptr))) {
        psfm_set_error("psfm_connect_batch: bad argument (n_seq=%d of at most %d, h=%d w=%d sample_ratio=%d)", n_seq, PSFM_BATCH_MAX, h, w, ratio);
        return PSFM_ERR_ARG;
    }
    const bool optimize = flows_f2 != nullptr;
    for (int i = 0; i < n_seq; ++i) {
        if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device) { psfm_set_error("psfm_connect_batch: context %d is NULL or on another device", i); return PSFM_ERR_ARG; }
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) { psfm_set_error("psfm_connect_batch: context %d given twice (one per sequence)", i); return PSFM_ERR_ARG; }
        if (n_flows[i] < 1 || !flows_f[i] || !flows_b[i] || (optimize && n_flows[i] > 1 && (!flows_f2[i] || !flows_b2[i]))) {
            psfm_set_error("psfm_connect_batch: bad sequence %d (n_flows=%d)", i, n_flows[i]);
            return PSFM_ERR_ARG;
        }
    }
    // The frame launches cover the lanes a sequence can be expected to use (its grid + 1/8), not its whole lane table (2 x the grid by
    // default): a block beyond the lanes in use leaves at once, but it still has to be dispatched -- half of the blocks of every launch
    // (PSFM_BATCH_GRID_TRIM=0: the whole table).  A launch that finds more lanes in use than it covers says so (overflow bit 16) and the
    // batch is run again with launches that cover the tables.
    static const bool trim_env = !(getenv("PSFM_BATCH_GRID_TRIM") && atoi(getenv("PSFM_BATCH_GRID_TRIM")) == 0);
    bool too_small = false;
    psfm_ctx* own0 = ctxs[0];
    const int64_t shape_key = ((int64_t)h << 40) ^ ((int64_t)w << 16) ^ (int64_t)ratio ^ (flows_f2 ? (1ll << 62) : 0);
    const bool trim = trim_env && own0->batch_notrim_shape != shape_key;      // (a shape whose sequences outgrew the trimmed launches once: not again)
    // (the counters a failed finalize reads the "grid too small" bit from are the ones THIS call refreshes: none left over from an earlier one)
    for (int i = 0; i < n_seq; ++i) if (ctxs[i]->host_pinned) ((PsfmCounters*)ctxs[i]->host_pinned)->overflow = 0;
    // A failed run leaves nothing in flight that still reads the caller's flow stacks or writes the maps: the side stream's flow_check
    // launches are drained before the error goes back (the caller may free or reuse the tensors), and before the untrimmed rerun.
    auto quiesce = [&]() {
        if (hipSetDevice(own0->device) != hipSuccess) return;
        if (own0->side_stream) (void)hipStreamSynchronize(own0->side_stream);
        (void)hipStreamSynchronize((hipStream_t)stream);
    };
    psfm_status rc = PSFM_ERR_HIP;
    try {
        rc = batch_run(ctxs, n_seq, flows_f, flows_b, flows_f2, flows_b2, n_flows, h, w, thres, ratio, infos, stream, trim, &too_small);
        if (rc != PSFM_OK) quiesce();
        if (rc == PSFM_ERR_CAPACITY && too_small) {
            own0->batch_notrim_shape = shape_key;
            rc = batch_run(ctxs, n_seq, flows_f, flows_b, flows_f2, flows_b2, n_flows, h, w, thres, ratio, infos, stream, false, &too_small);
            if (rc != PSFM_OK) quiesce();
        }
    } catch (const std::exception& e) {      // (std::thread / std::vector: nothing may unwind through the extern "C" boundary)
        quiesce();
        psfm_set_error("psfm_connect_batch: %s", e.what());
        rc = PSFM_ERR_HIP;
    }
    return rc;
}

static psfm_status batch_run(psfm_ctx* const* ctxs, int n_seq, const float* const* flows_f, const float* const* flows_b,
                             const float* const* flows_f2, const float* const* flows_b2, const int* n_flows, int h, int w,
                             float thres, int ratio, psfm_track_info* infos, void* stream, bool trim, bool* grid_too_small)
{
    const bool optimize = flows_f2 != nullptr;
    *grid_too_small = false;
    psfm_ctx* own = ctxs[0];
    PSFM_HIP(hipSetDevice(own->device));
    hipStream_t s = (hipStream_t)stream;
    const int B = n_seq;
    const int64_t P = (int64_t)h * w;
    std::vector<int> redo;
    std::vector<SeqState> S((size_t)B);
    psfm_status st;
    // events of the flow_check pipeline: back into the owner's pool when the call ends, however it ends (the stream is synchronised
    // or the call failed; a recycled event is re-recorded before anything waits on it)
    struct EventReturn {
        psfm_ctx* c; std::vector<hipEvent_t> ev;
        void push_back(hipEvent_t e) { ev.push_back(e); }
        ~EventReturn() { for (auto e : ev) c->prof.pool.push_back(e); }
    } fc_events{own, {}};
    {
        PsfmGate gate(own->device, 0);        // launches only: nothing here needs the device to itself
        // ---- dimensions (one key format for the whole batch: the time bits of its longest sequence), workspaces ----
        std::vector<PsfmTrackDims> D((size_t)B);
        int n_max = 0;
        for (int i = 0; i < B; ++i) n_max = n_flows[i] > n_max ? n_flows[i] : n_max;
        int tbits = 1;
        while ((1ll << tbits) < (long long)n_max + 2) ++tbits;
        int64_t cap_max = 0;
        int64_t grid_lanes = 0;      // lanes the frame launches cover (<= cap_max); see below
        for (int i = 0; i < B; ++i) {
            psfm_ctx* c = ctxs[i];
            psfm_shard_abandon(c);
            if ((st = psfm_track_dims(c, n_flows[i], h, w, ratio, -1, D[i])) != PSFM_OK) return st;
            D[i].shift_d = D[i].shift_b + tbits;
            if (D[i].shift_d + tbits > 63) { psfm_set_error("psfm_connect_batch: key does not fit 64 bits"); return PSFM_ERR_ARG; }
            if ((st = psfm_track_alloc(c, D[i])) != PSFM_OK) return st;
            if ((st = c->occ_own.ensure((size_t)P * (size_t)n_flows[i])) != PSFM_OK) return st;
            if (optimize) {
                if ((st = c->occ2_own.ensure((size_t)P * (size_t)(n_flows[i] > 1 ? n_flows[i] - 1 : 1))) != PSFM_OK) return st;
                if ((st = psfm_solve_prepare(c, D[i], s)) != PSFM_OK) return st;
            }
            c->solve_stats.clear();
            c->res_n_traj = c->res_n_points = 0;
            c->res_n_flows = n_flows[i];
            c->pc_persist_ok = false;
            c->pc_giveups = 0;
            c->n_fused_ok = c->n_fused_redone = c->n_chain = 0;
            c->n_resident = c->n_iter_launches = 0;
            if (D[i].cap > cap_max) cap_max = D[i].cap;
            if (D[i].G + D[i].G / 8 + 2048 > grid_lanes) grid_lanes = D[i].G + D[i].G / 8 + 2048;
            S[i].hstats.assign((size_t)n_flows[i] + 1, psfm_solve_stats());
            // a context told to use the launch chain for every solve has nothing to gain from the batch: it runs alone
            if (optimize && c->solver_mode == 1 && n_flows[i] >= 2) S[i].dropped = true;
        }
        if (trim && getenv("PSFM_BATCH_GRID_LANES")) grid_lanes = atoll(getenv("PSFM_BATCH_GRID_LANES"));      // (tests: launches that cover too little)
        if (!trim || grid_lanes > cap_max || grid_lanes < 256) grid_lanes = cap_max;
        // ---- flow_check of every stack (utils.py:94-105) ----
        // Bandwidth-bound work beside a latency-bound frame loop: the maps are produced on the owner's side stream in chunks of
        // frame pairs -- ONE launch per chunk for all sequences (psfm_flow_check_x2v_batch_kernel) -- and the frame loop only waits
        // for the chunk it is about to read (what psfm_connect does for one sequence).  PSFM_BATCH_FC_CHUNK=0, or stacks the
        // 16-byte-load kernel cannot take: every stack up front on the launch stream.
        const int fc_env = getenv("PSFM_BATCH_FC_CHUNK") ? atoi(getenv("PSFM_BATCH_FC_CHUNK")) : -1;
        int fc_chunk = fc_env >= 0 ? fc_env : (optimize ? 6 : 8);
        for (int i = 0; i < B && fc_chunk > 0; ++i) {
            if (S[i].dropped) continue;
            if (!psfm_flow_check_batch_ok(h, w, flows_f[i], flows_b[i], ctxs[i]->occ_own.p)) fc_chunk = 0;
            if (optimize && n_flows[i] > 1 && !psfm_flow_check_batch_ok(h, w, flows_f2[i], flows_b2[i], ctxs[i]->occ2_own.p)) fc_chunk = 0;
        }
        std::vector<hipEvent_t> fc_ready, fc_ready2;      // chunk c of the stride-1 / stride-2 maps: pairs [c * fc_chunk, (c + 1) * fc_chunk)
        int fc_waited = -1, fc_waited2 = -1;
        auto fc_need = [&](int pair, bool s2) -> psfm_status {
            if (fc_chunk <= 0) return PSFM_OK;
            std::vector<hipEvent_t>& ev = s2 ? fc_ready2 : fc_ready;
            int& wv = s2 ? fc_waited2 : fc_waited;
            const int cidx = pair / fc_chunk;
            while (wv < cidx && wv + 1 < (int)ev.size()) {
                ++wv;
                PSFM_HIP(hipStreamWaitEvent(s, ev[(size_t)wv], 0));
            }
            return PSFM_OK;
        };
        own->prof.begin(PSFM_PROF_FLOW_CHECK, s);
        if (fc_chunk <= 0) {
            for (int i = 0; i < B; ++i) {
                if (S[i].dropped) continue;
                psfm_ctx* c = ctxs[i];
                if ((st = psfm_launch_flow_check(flows_f[i], flows_b[i], n_flows[i], h, w, thres, c->occ_own.as<uint8_t>(), nullptr, s)) != PSFM_OK) return st;
                if (optimize && n_flows[i] > 1 &&
                    (st = psfm_launch_flow_check(flows_f2[i], flows_b2[i], n_flows[i] - 1, h, w, thres, c->occ2_own.as<uint8_t>(), nullptr, s)) != PSFM_OK) return st;
            }
        } else {
            const size_t fbytes = sizeof(PsfmFcSeq) * (size_t)B * 2;
            if ((st = batch_host_staging2(own, fbytes)) != PSFM_OK) return st;
            if ((st = own->batch_fc.ensure(fbytes)) != PSFM_OK) return st;
            PsfmFcSeq* hfc = (PsfmFcSeq*)own->host_batch2;
            for (int i = 0; i < B; ++i) {
                psfm_ctx* c = ctxs[i];
                hfc[i].ff = flows_f[i]; hfc[i].fb = flows_b[i]; hfc[i].occ = c->occ_own.as<uint8_t>(); hfc[i].n_pairs = S[i].dropped ? 0 : n_flows[i]; hfc[i].pad = 0;
                hfc[B + i].ff = optimize ? flows_f2[i] : nullptr; hfc[B + i].fb = optimize ? flows_b2[i] : nullptr;
                hfc[B + i].occ = optimize ? c->occ2_own.as<uint8_t>() : nullptr;
                hfc[B + i].n_pairs = (optimize && !S[i].dropped && n_flows[i] > 1) ? n_flows[i] - 1 : 0; hfc[B + i].pad = 0;
            }
            PSFM_HIP(hipMemcpyAsync(own->batch_fc.p, hfc, fbytes, hipMemcpyHostToDevice, s));
            if (!own->side_stream) PSFM_HIP(hipStreamCreateWithFlags(&own->side_stream, hipStreamNonBlocking));
            hipStream_t side = own->side_stream;
            hipEvent_t e_in = own->prof.get();      // the side stream starts behind whatever the caller enqueued on `stream` (the inputs, the table)
            PSFM_HIP(hipEventRecord(e_in, s));
            PSFM_HIP(hipStreamWaitEvent(side, e_in, 0));
            fc_events.push_back(e_in);
            const PsfmFcSeq* dfc = own->batch_fc.as<PsfmFcSeq>();
            for (int p0 = 0; p0 < n_max; p0 += fc_chunk) {
                const int np = n_max - p0 < fc_chunk ? n_max - p0 : fc_chunk;
                if ((st = psfm_launch_flow_check_batch(dfc, B, p0, np, h, w, thres, side)) != PSFM_OK) return st;
                hipEvent_t e = own->prof.get();
                PSFM_HIP(hipEventRecord(e, side));
                fc_ready.push_back(e); fc_events.push_back(e);
                if (optimize && p0 < n_max - 1) {     // the stride-2 maps of the same time range follow their stride-1 chunk
                    const int np2 = n_max - 1 - p0 < fc_chunk ? n_max - 1 - p0 : fc_chunk;
                    if ((st = psfm_launch_flow_check_batch(dfc + B, B, p0, np2, h, w, thres, side)) != PSFM_OK) return st;
                    hipEvent_t e2 = own->prof.get();
                    PSFM_HIP(hipEventRecord(e2, side));
                    fc_ready2.push_back(e2); fc_events.push_back(e2);
                }
            }
        }
        own->prof.end(s);
        // ---- the table of sequences ----
        const int CHECK = 16;
        const int win_max = CHECK + 2;
        const size_t row_opt = psfm_batch_seq_opt_bytes();
        const size_t tab_bytes = sizeof(PsfmBatchSeq) * (size_t)B, tabo_bytes = optimize ? row_opt * (size_t)B : 0;
        const size_t pack_row = psfm_batch_pack_row_bytes(win_max), pack_bytes = pack_row * (size_t)B;
        if ((st = batch_host_staging(own, tab_bytes + tabo_bytes + pack_bytes)) != PSFM_OK) return st;
        if ((st = own->batch_tab.ensure(tab_bytes + tabo_bytes)) != PSFM_OK) return st;
        if ((st = own->batch_ws.ensure(pack_bytes)) != PSFM_OK) return st;
        PsfmBatchSeq* htab = (PsfmBatchSeq*)own->host_batch;
        char* htabo = (char*)own->host_batch + tab_bytes;
        char* hpack = htabo + tabo_bytes;
        PsfmBatchSeq* dtab = own->batch_tab.as<PsfmBatchSeq>();
        char* dtabo = (char*)own->batch_tab.p + tab_bytes;
        for (int i = 0; i < B; ++i) {
            psfm_ctx* c = ctxs[i];
            // (a sequence that has left the batch keeps a row -- blockIdx.y indexes the table -- with no frames to run)
            psfm_batch_fill_seq(c, D[i], flows_f[i], c->occ_own.as<uint8_t>(), P, &htab[i]);
            if (S[i].dropped) htab[i].n_flows = 0;
            if (optimize) {
                const float* f2 = n_flows[i] > 1 ? flows_f2[i] : flows_f[i];      // (a one-pair sequence has no stride-2 stack: never read)
                if ((st = psfm_batch_fill_seq_opt(c, D[i], flows_f[i], c->occ_own.as<uint8_t>(), P, f2, c->occ2_own.as<uint8_t>(),
                                                  htabo + row_opt * (size_t)i, s)) != PSFM_OK) return st;
                // a run whose launches covered too few lanes has left the fused solve's tickets mid-count (blocks they were waiting
                // for never existed): the run that repeats it starts them from zero
                if (!trim && c->sol_fused.p) PSFM_HIP(hipMemsetAsync(c->sol_fused.p, 0, 4096 * sizeof(unsigned), s));
            }
        }
        PSFM_HIP(hipMemcpyAsync(dtab, htab, tab_bytes + tabo_bytes, hipMemcpyHostToDevice, s));
        // (dropped sequences: n_flows = 0 in the chain table keeps the init kernel's survivor loop short; their lane tables are
        // re-initialised by the psfm_connect that redoes them)
        if ((st = psfm_launch_track_init_batch(dtab, B, cap_max, s)) != PSFM_OK) return st;

        if (!optimize) {
            // ---- track.py:31-47: one launch per frame index for the whole batch ----
            for (int f = 0; f < n_max; ++f) {
                if ((st = fc_need(f, false)) != PSFM_OK) return st;
                if ((st = psfm_launch_chain_step_batch(own, dtab, B, ratio, grid_lanes, f, false, s)) != PSFM_OK) return st;
            }
        } else {
            // ---- track_optimize.py:31-50: frame 0 is a plain chain step; from frame 1 on device-paced launches ----
            if ((st = fc_need(0, false)) != PSFM_OK) return st;
            if ((st = psfm_launch_chain_step_batch(own, dtab, B, ratio, grid_lanes, 0, true, s)) != PSFM_OK) return st;
            int launch_id = 0, idle_windows = 0;
            int v[PSFM_BATCH_MAX][4];
            for (int i = 0; i < B; ++i) {
                S[i].f = 1;
                S[i].resync = true;       // (track_init left pc = {frame 1, phase 0, launch 0, K 3}: set this context's K)
            }
            for (;;) {
                int left_max = 0, f_top = 0;
                bool any_set = false;
                for (int i = 0; i < B; ++i) {
                    psfm_ctx* c = ctxs[i];
                    v[i][0] = -1; v[i][1] = 0; v[i][2] = launch_id; v[i][3] = 0;
                    const int n_i = S[i].dropped ? 0 : n_flows[i];
                    const int k_now = c->solver_K > 0 ? c->solver_K : c->solve_K;
                    if (S[i].f < n_i) {
                        if (n_i - S[i].f > left_max) left_max = n_i - S[i].f;
                        if (S[i].f > f_top) f_top = S[i].f;
                        if (S[i].resync) { v[i][0] = S[i].f; v[i][3] = k_now; any_set = true; }
                        else if (k_now != S[i].k_dev) { v[i][0] = -2; v[i][3] = k_now; any_set = true; }      // (K only: the device may be inside a solve)
                        S[i].k_dev = k_now;
                    } else if (S[i].resync) {   // finished or left the batch behind a redone solve: park its program counter at the end
                        v[i][0] = n_flows[i]; v[i][3] = k_now; any_set = true;
                    }
                    S[i].resync = false;
                }
                if (left_max == 0) break;
                if (any_set && (st = psfm_launch_batch_set_pc(dtabo, v, B, s)) != PSFM_OK) return st;
                const int n_launch = (left_max < CHECK ? left_max : CHECK) + 2;      // two spare: continuation launches of the window
                {   // the furthest frame a launch of this window can reach, and the maps it reads
                    const int f_hi = f_top + n_launch - 1 < n_max - 1 ? f_top + n_launch - 1 : n_max - 1;
                    if ((st = fc_need(f_hi, false)) != PSFM_OK) return st;
                    if ((st = fc_need(f_hi - 1, true)) != PSFM_OK) return st;
                }
                if ((st = psfm_launch_seq_batch(own, dtabo, B, ratio, grid_lanes, n_launch, launch_id, s)) != PSFM_OK) return st;
                launch_id += n_launch;
                // ---- ONE checkpoint for all sequences ----
                int lo[PSFM_BATCH_MAX];
                for (int i = 0; i < B; ++i) lo[i] = S[i].first_unchecked;
                if ((st = psfm_launch_batch_pack(dtabo, lo, B, win_max, own->batch_ws.as<char>(), s)) != PSFM_OK) return st;
                PSFM_HIP(hipMemcpyAsync(hpack, own->batch_ws.p, pack_bytes, hipMemcpyDeviceToHost, s));
                PSFM_HIP(hipStreamSynchronize(s));
                bool progress = false;
                for (int i = 0; i < B; ++i) {
                    if (S[i].dropped || S[i].f >= n_flows[i]) continue;
                    psfm_ctx* c = ctxs[i];
                    const PsfmCounters hc = *(const PsfmCounters*)(hpack + pack_row * (size_t)i);
                    const psfm_solve_stats* hw = (const psfm_solve_stats*)(hpack + pack_row * (size_t)i + sizeof(PsfmCounters));
                    const int n_i = n_flows[i];
                    if ((hc.overflow & 7) != 0 || hc.n_lanes > D[i].cap) {      // (as psfm_track_impl's checkpoint: a full table ends the run here)
                        psfm_set_error("capacity exceeded (sequence %d of the batch): lanes used %d of %lld, overflow bits %d; raise psfm_ctx_set_capacity",
                                       i, hc.n_lanes, (long long)D[i].cap, hc.overflow);
                        return PSFM_ERR_CAPACITY;
                    }
                    if (hc.overflow & 16) {      // a launch covered fewer lanes than the sequence had in use: once more, with launches that cover the tables
                        *grid_too_small = true;
                        psfm_set_error("psfm_connect_batch: sequence %d uses more lanes (%d) than the trimmed launches cover (%lld)", i, hc.n_lanes, (long long)grid_lanes);
                        return PSFM_ERR_CAPACITY;
                    }
                    for (int k = 0; k < win_max && lo[i] + k <= n_i; ++k) S[i].hstats[(size_t)(lo[i] + k)] = hw[k];
                    const int stalled = hc.stall ? hc.stall - 1 : -1;
                    int last_ok = (hc.pc_frame < n_i ? hc.pc_frame : n_i) - 1;     // frames below the device's program counter are complete
                    if (stalled >= 0 && c->solver_mode != 2 && stalled - S[i].first_unchecked < 8) {
                        // ... with fewer than eight solves of the window in front of it: whatever they were, the window counts as one whose
                        // solves reject steps (one in eight is the bar below) and the sequence is about to leave the batch -- and to be run
                        // again from its first frame by psfm_connect.  No point in redoing this solve with launches first.
                        S[i].dropped = true; S[i].resync = true;
                        c->solve_mode = 1;        // (what psfm_connect starts its first window with: the launch chain / the resident solve)
                        progress = true;
                        continue;
                    }
                    if (stalled >= 0) {
                        // a solve that did not go as speculated (a rejected step, a dogleg interpolation, more iterations than the
                        // continuation launches cover): redone by the launch chain from the values it started with, as in psfm_track
                        const int fs = stalled;
                        psfm_solve_stats ss;
                        memset(&ss, 0, sizeof(ss));
                        psfm_status st2 = psfm_solve_frame_resume(c, D[i], flows_f[i] + (size_t)(fs - 1) * P * 2, flows_f[i] + (size_t)fs * P * 2,
                                                                  flows_f2[i] + (size_t)(fs - 1) * P * 2, c->occ2_own.as<uint8_t>() + (size_t)(fs - 1) * P,
                                                                  fs, &ss, 0, false, s);
                        if (st2 != PSFM_OK) return st2;
                        S[i].hstats[(size_t)fs] = ss;
                        last_ok = fs;
                        S[i].resync = true;
                    }
                    int n_solved = 0, n_unclean = 0;
                    int hist[16] = {0};
                    for (int k = S[i].first_unchecked; k <= last_ok; ++k) {
                        const psfm_solve_stats& q = S[i].hstats[(size_t)k];
                        if (q.termination < 0) continue;      // -1: no track had a full buffer, nothing was solved
                        c->solve_stats.push_back(q);
                        S[i].total_iters += q.iterations;
                        const bool clean = q.dogleg_nonGN == 0 && q.termination != PSFM_TERM_FAILURE &&
                                           (q.iterations == q.successful_steps + 1 ||
                                            (q.termination == PSFM_TERM_GRADIENT_TOL && q.iterations == q.successful_steps)) &&
                                           q.successful_steps + 1 <= psfm_solve_kmax();
                        ++n_solved;
                        if (!clean) ++n_unclean;
                        ++hist[q.successful_steps + 1 < 15 ? q.successful_steps + 1 : 15];
                        if (k == stalled) ++c->n_fused_redone; else ++c->n_fused_ok;
                    }
                    if (n_solved > 0) {
                        c->solve_mode = (n_unclean * 8 > n_solved) ? 1 : 0;
                        int best = 2;
                        for (int q = 2; q <= psfm_solve_kmax(); ++q) if (hist[q] > hist[best]) best = q;
                        c->solve_K = best;       // the most common need of the window (an iteration more costs one more launch)
                    }
                    if (last_ok + 1 > S[i].f) progress = true;
                    S[i].first_unchecked = last_ok + 1;
                    S[i].f = last_ok + 1;
                    // a window of solves that reject steps: this sequence belongs to the launch chain / the resident solve -- it leaves the
                    // batch and runs alone behind it (psfm_ctx_set_solver(ctx, 2, k) keeps it in)
                    if (c->solve_mode == 1 && c->solver_mode != 2 && S[i].f < n_i) { S[i].dropped = true; S[i].resync = true; }
                }
                // (no frame completed anywhere and nothing stalled: every sequence still running is inside one long solve -- all iterations
                // accepted, none terminating -- which the next window's launches continue; psfm_track_impl has the same case)
                idle_windows = progress ? 0 : idle_windows + 1;
                if (idle_windows > 8) { psfm_set_error("psfm_connect_batch: no sequence advanced in 8 windows of launches"); return PSFM_ERR_SOLVER; }
                if (trim) {
                    // the next window's launches cover the lanes the sequences have in use NOW plus head-room for 18 frames of growth
                    // (on a dense grid the high-water mark creeps up all through a long sequence: a death is a birth one frame later,
                    // and not always on a lane of the same block)
                    int64_t in_use = 0;
                    for (int i = 0; i < B; ++i) {
                        const PsfmCounters* hc = (const PsfmCounters*)(hpack + pack_row * (size_t)i);
                        if (!S[i].dropped && hc->n_lanes > in_use) in_use = hc->n_lanes;
                    }
                    const int64_t want = in_use + in_use / 16 + 2048;
                    if (want > grid_lanes) grid_lanes = want < cap_max ? want : cap_max;
                }
            }
            if ((st = psfm_launch_flush_batch(dtabo, B, cap_max, s)) != PSFM_OK) return st;
        }
        // (every chunk of the side stream has been waited for by now unless all sequences left the batch early: the redo below
        // writes the same maps)
        if ((st = fc_need(n_max - 1, false)) != PSFM_OK) return st;
        if (optimize && n_max >= 2 && (st = fc_need(n_max - 2, true)) != PSFM_OK) return st;
        // ---- ONE segmented finalize for the sequences that stayed ----
        std::vector<psfm_ctx*> kc;
        std::vector<PsfmTrackDims> kd;
        std::vector<int> ki;
        for (int i = 0; i < B; ++i) {
            if (S[i].dropped) redo.push_back(i);
            else { kc.push_back(ctxs[i]); kd.push_back(D[i]); ki.push_back(i); }
        }
        if (!kc.empty()) {
            own->prof.begin(PSFM_PROF_FINALIZE, s);
            st = psfm_finalize_batch(own, kc.data(), kd.data(), (int)kc.size(), s);
            own->prof.end(s);
            if (st == PSFM_ERR_CAPACITY)
                for (psfm_ctx* c : kc) if (((PsfmCounters*)c->host_pinned)->overflow & 16) *grid_too_small = true;
            if (st != PSFM_OK) return st;
        }
        PSFM_HIP(hipStreamSynchronize(s));
        own->prof.collect();
        if (infos) {
            for (size_t q = 0; q < ki.size(); ++q) {
                const int i = ki[q];
                psfm_ctx* c = ctxs[i];
                psfm_track_info* info = &infos[i];
                memset(info, 0, sizeof(*info));
                info->n_traj = c->res_n_traj;
                info->n_points = c->res_n_points;
                info->n_lanes_peak = ((PsfmCounters*)c->host_pinned)->n_lanes;
                info->lane_capacity = D[i].cap;
                info->solver_iterations = S[i].total_iters;
                info->n_solves = (int32_t)c->solve_stats.size();
                info->chain_mode = 3;
            }
        }
    }
    // ---- sequences that left the batch (their solves reject steps): through psfm_connect, with everything it has for them -- the
    // resident solve above all.  One sequence: the call as it is (it may take the device to itself).  Several: a few of them at a
    // time on host threads of this call, every one with an equal share of the device's co-resident block slots for its resident
    // solves (psfm_ctx_set_resident_budget: 1.6-1.8x one after the other on Sintel / DAVIS-sized realistic flows,
    // profiles/r05/r05_f_resident_budget.txt); as many at a time as have room on chip for their tracks.  (The gate is released.) ----
    if (redo.empty()) return PSFM_OK;
    int n_thr = 1;
    // what the redo threads may share: the CALLER's budget when it set one (psfm.h: the budgets of all contexts in flight on the device
    // add up to at most the capacity -- connect_sequences gives every worker capacity / n_threads, and two workers' batches may redo at
    // the same time), else the device's co-resident block slots
    const int room = own->resident_budget > 0 ? std::min(own->resident_budget, psfm_resident_blocks(own)) : psfm_resident_blocks(own);
    if (redo.size() >= 2 && optimize) {
        const int capacity = room;
        const int64_t G = (int64_t)((w + ratio - 1) / ratio) * ((h + ratio - 1) / ratio);
        const int64_t need = (G * 6 / 10 + 767) / 768;            // blocks that hold ~60 % of the grid's tracks at three per thread
        int fit = need > 0 ? (int)(capacity / need) : 4;
        if (const char* e = getenv("PSFM_BATCH_REDO_THREADS")) fit = atoi(e);
        n_thr = fit < 1 ? 1 : (fit > 4 ? 4 : fit);
        if (n_thr > (int)redo.size()) n_thr = (int)redo.size();
        if (capacity <= 0) n_thr = 1;
    }
    if (n_thr == 1) {
        for (int i : redo) {
            st = psfm_connect(ctxs[i], flows_f[i], flows_b[i], optimize ? flows_f2[i] : nullptr, optimize ? flows_b2[i] : nullptr, n_flows[i], h, w, thres,
                              ratio, nullptr, nullptr, infos ? &infos[i] : nullptr, stream);
            if (st != PSFM_OK) return st;
        }
        return PSFM_OK;
    }
    const int share = std::max(room / n_thr, 1);
    std::vector<psfm_status> rc((size_t)n_thr, PSFM_OK);
    std::vector<std::string> msg((size_t)n_thr);
    std::vector<std::thread> workers;
    auto body = [&](int t) {
            for (size_t q = (size_t)t; q < redo.size(); q += (size_t)n_thr) {
                const int i = redo[q];
                psfm_ctx* c = ctxs[i];
                if (hipSetDevice(c->device) != hipSuccess) { rc[(size_t)t] = PSFM_ERR_HIP; msg[(size_t)t] = "hipSetDevice failed"; return; }
                if (!c->redo_stream && hipStreamCreateWithFlags(&c->redo_stream, hipStreamNonBlocking) != hipSuccess) {
                    rc[(size_t)t] = PSFM_ERR_HIP; msg[(size_t)t] = "hipStreamCreateWithFlags failed"; return;
                }
                const int budget0 = c->resident_budget;
                c->resident_budget = share;
                const psfm_status r = psfm_connect(c, flows_f[i], flows_b[i], flows_f2[i], flows_b2[i], n_flows[i], h, w, thres, ratio, nullptr, nullptr,
                                                   infos ? &infos[i] : nullptr, (void*)c->redo_stream);
                c->resident_budget = budget0;
                if (r != PSFM_OK) { rc[(size_t)t] = r; msg[(size_t)t] = psfm_last_error(); return; }      // (the error text is thread local)
            }
        };
    std::vector<int> inline_shares;      // shares whose thread could not be created (std::system_error): run here, behind the others
    for (int t = 0; t < n_thr; ++t) {
        try { workers.emplace_back(body, t); } catch (const std::exception&) { inline_shares.push_back(t); }
    }
    for (int t : inline_shares) body(t);
    for (auto& th : workers) th.join();
    for (int t = 0; t < n_thr; ++t)
        if (rc[(size_t)t] != PSFM_OK) { psfm_set_error("psfm_connect_batch: a sequence that left the batch failed: %s", msg[(size_t)t].c_str()); return rc[(size_t)t]; }
    return PSFM_OK;
}
