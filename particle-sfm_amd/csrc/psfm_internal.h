// psfm_internal.h -- host-side context and launcher prototypes (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/psfm.h"

#define PSFM_MAX_PEERS 8      // ranks of one solve (psfm_shard_peer_*): the GPUs of one node

void psfm_set_error(const char* fmt, ...);

#define PSFM_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            psfm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PSFM_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

// A frame size every kernel can address: the samplers reach a map as base + a 32-bit BYTE offset (psfm_device.h: "every map / table
// here is < 4 GB"), so the largest per-frame map -- (H,W,2) f32 = 8 H W bytes -- must stay below 2^32 (H W < 2^29: 23170 x 23170;
// a 1080p frame is 2^21).  Every entry point that takes h, w checks this first (PSFM_ERR_ARG), none wraps silently.
static inline bool psfm_frame_ok(int h, int w) { return h >= 2 && w >= 2 && (int64_t)h * (int64_t)w * 8 < ((int64_t)1 << 32); }

// Per device and process: every entry point that launches kernels or copies holds this gate SHARED; the persistent frame
// loop needs it EXCLUSIVE.  All blocks of that kernel must be resident at once, and with another queue feeding the device
// they may never be (measured: a second host thread running psfm_connect alongside stalls the loop until its spin limit)
// -- so it only runs when no other psfm call of this process is in flight on the device, and calls that arrive meanwhile
// wait for it (<= a few ms).  A call that finds the device busy uses per-frame launches, which overlap well with other
// sequences.  (Other PROCESSES on the device are not covered: there the loop's bounded spin and its hand-over to
// per-frame launches are the safety net.)
std::shared_mutex& psfm_device_gate(int device);
int psfm_gate_waiters(int device);             // psfm calls of this process currently blocked behind an exclusive holder on the device
void psfm_gate_waiters_add(int device, int d);
struct PsfmGate {
    std::shared_mutex& m;
    int device;
    bool exclusive = false;
    PsfmGate(int dev, int want_exclusive /* 0 no, 1 if free, 2 wait for it */) : m(psfm_device_gate(dev)), device(dev)
    {
        if (want_exclusive == 2) { m.lock(); exclusive = true; }
        else if (want_exclusive == 1 && m.try_lock()) exclusive = true;
        else if (!m.try_lock_shared()) {
            // somebody holds the device exclusively (a persistent launch, or a track_optimize sequence running its solves as
            // resident launches): say so -- a sequence gives the device up at its next checkpoint (yield_exclusive) -- and wait
            psfm_gate_waiters_add(device, 1);
            m.lock_shared();
            psfm_gate_waiters_add(device, -1);
        }
    }
    // An exclusive holder whose work can go on with plain launches lets the waiting calls in (call with nothing of this call in
    // flight on the device: behind a stream synchronisation).  Returns true when it gave the device up.
    bool yield_exclusive()
    {
        if (!exclusive || psfm_gate_waiters(device) <= 0) return false;
        m.unlock();
        exclusive = false;
        m.lock_shared();
        return true;
    }
    ~PsfmGate() { if (exclusive) m.unlock(); else m.unlock_shared(); }
    PsfmGate(const PsfmGate&) = delete;
    PsfmGate& operator=(const PsfmGate&) = delete;
};

// grow-only device buffer
struct PsfmBuf {
    void* p = nullptr;
    size_t bytes = 0;
    psfm_status ensure(size_t need);
    void release();
    template <class T> T* as() const { return (T*)p; }
};

// Device-side counters of the frame recurrence (one cache line, zeroed at init).
struct PsfmCounters {
    int n_lanes;      // lanes ever handed out (high-water mark); chain_step scans [0, n_lanes)
    int overflow;     // bit0: lane table / free stack full, bit1: trajectory table full, bit2: more tracks than resident
                      // lanes (persistent loop), bit3: barrier spin limit hit (persistent loop), bit4: a batched frame launch
                      // covered fewer lanes than the sequence had in use (psfm_batch.hip runs the batch again, untrimmed)
    int stall;        // != 0: solve of frame stall-1 ran out of unrolled iterations; later launches are no-ops
    int abort;        // persistent frame loop: a block gave up (spin limit); every block leaves at its next barrier
    int spill_cnt;    // persistent frame loop: records written to the shared tail behind the private segments
    int sel;          // track_optimize: iterate buffer that holds the accepted positions (times f, f+1) of the last fused
                      // solve; 0 = they are in the log.  The next chain step copies them on its way (psfm_solver.hip)
    int n_lanes_snap[2]; // n_lanes as the launch in front of frame f left it, in entry f & 1 (set by track_init, by the control
                      // thread of the frame kernel of f-1 and by the solver's write-back): the tile bound every block of the frame
                      // kernel of f agrees on -- n_lanes itself grows while that launch runs, and nothing it reads changes under it
    // device-paced sequence (psfm_seq_kernel): which frame the next launch works on (pc_phase 0: its chain step + fused solve,
    // 1: more iterations of its solve), which launch that is (a block of an earlier launch that starts after its control thread
    // has moved on must not pick the new work up), iterations per fused launch
    int pc_frame, pc_phase, pc_owner, solve_K;
    int pad[4];
};

// Death records and free lanes are published through PSFM_NSHARD independent tables so that the
// per-block atomics land on different words (a single word saturates at ~88 atomics/us on MI355X).
#define PSFM_NSHARD 64
struct PsfmShard {
    int fin_cnt;      // trajectory records written into this shard's slice
    unsigned points;  // trajectory points written through this shard's blocks (set 0 only)
    int pad0[30];
    int free_top;     // top of this shard's free-lane stack
    int pad1[31];
};

enum { PSFM_PROF_FLOW_CHECK = 0, PSFM_PROF_CHAIN = 1, PSFM_PROF_RESPAWN = 2, PSFM_PROF_SOLVER = 3,
       PSFM_PROF_FINALIZE = 4, PSFM_PROF_KINDS = 5 };

struct PsfmProfiler {
    bool enabled = false;
    int stride = 1;          // kernel_span() hands out events for every `stride`-th call only
    int64_t calls = 0;
    struct Span { int kind; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    double total_ms[PSFM_PROF_KINDS] = {0};
    int64_t launches[PSFM_PROF_KINDS] = {0};
    hipEvent_t get();
    void begin(int kind, hipStream_t s);
    void kernel_span(int kind, hipEvent_t* a, hipEvent_t* b, bool always = false);  // events for hipExtLaunchKernelGGL (exact kernel begin/end)
    void end(hipStream_t s);
    void collect();  // after a stream sync: fold spans into totals
    void reset();
    void destroy();
};

// Device control block of the path-consistency solver (written by the last block of each kernel).
struct PsfmSolveCtrl;

struct PsfmTrackDims;

struct psfm_ctx {
    int device = 0;
    double lane_factor = 2.0, traj_factor = 8.0;
    // frame recurrence workspace
    PsfmBuf log;          // (n_flows+1) x cap double2
    PsfmBuf birth_frame;  // cap i32, -1 = free lane
    PsfmBuf birth_idx;    // cap i32
    PsfmBuf free_stack;   // 2 sets x PSFM_NSHARD x free_cap i32 (double-buffered by frame parity)
    PsfmBuf fin_keys;     // traj_cap u64
    PsfmBuf fin_lanes;    // traj_cap i32
    PsfmBuf occupied;     // 2 x G u8: frame-stamped grid-resolution `blocked` maps
    PsfmBuf counters;     // PsfmCounters
    PsfmBuf shards;       // 2 x PSFM_NSHARD x PsfmShard
    PsfmBuf survivors;    // (n_flows+1) i32
    // persistent frame loop (psfm_persist.hip)
    PsfmBuf handoff;      // cap x 3 u64: newborn handed to the owner of a popped lane
    PsfmBuf seg_info;     // per block: records in its private segment, trajectory points it wrote
    PsfmBuf seg_table;    // finalize: (src start, count, dst start) per record segment
    PsfmBuf persist_bar;  // barrier: 64 arrival counters, 1 top counter, 64 release flags (128 B apart)
    int persist_max_blocks = -1;   // resident 256-thread blocks on this device (-1: not queried yet)
    int chain_mode = 0;            // 0 auto, 1 per-frame launches only, 2 persistent loop required
    void* host_seg = nullptr;      // pinned staging for seg_info / seg_table
    size_t host_seg_bytes = 0;
    // finalize workspace + result
    PsfmBuf sort_keys, sort_lanes, sort_tmp, scan_tmp;
    PsfmBuf fin_marks;            // finalize: first sorted record of every (last, birth) group; all -1 between two calls (fin_marks_clean)
    bool fin_marks_clean = false;
    PsfmBuf res_birth, res_len, res_off, res_xy;
    int64_t res_n_traj = 0, res_n_points = 0;
    int res_n_flows = 0;           // flows of the sequence the result came from (psfm_result_keys checks its packed key)
    // solver workspace
    PsfmBuf sol_x, sol_state, sol_partials, sol_ctrl, sol_misc, sol_stats, sol_fused, sol_bar;
    PsfmBuf sol_list;              // launch chain / resident solve: per block, the lanes that take part in the solve (pc_build_list)
    int pc_persist_blocks[4] = {-1, -1, -1, -1};   // co-resident blocks of psfm_pc_resident_kernel<NS> on this device (-1: not queried yet)
    bool pc_persist_ok = false;    // this call has the device to itself (or a resident budget): the launch chain may run as one persistent launch
    int resident_budget = 0;       // psfm_ctx_set_resident_budget: > 0 = resident solves of at most that many blocks under the SHARED gate
    unsigned pc_epoch = 0;         // resident solve: launch counter, part of every granule's tag (stale granules never match)
    // ONE solve over several ranks (psfm_shard_peer_*): this rank's granule area (member rows + the leader rows every rank writes into),
    // the other ranks' leader areas as this process addresses them, every rank's leader count
    PsfmBuf peer_area;
    unsigned peer_epoch = 0;         // the last epoch a launch of this context tagged granules of the area with: a new engine on the same
                                     // context (same area) must go on from here, never start over (stale granules would match)
    int peer_world = 0, peer_rank = 0;
    int peer_L[PSFM_MAX_PEERS] = {0};
    void* peer_lead[PSFM_MAX_PEERS] = {nullptr};
    void* peer_opened[PSFM_MAX_PEERS] = {nullptr};   // mappings this context opened with hipIpcOpenMemHandle (closed with it)
    int pc_giveups = 0;            // resident solves of this call whose hand-off timed out (two of them: launches from there on)
    int solve_K = 4;        // fused solve: trust-region iterations speculated per launch (adapted at checkpoints)
    int solve_mode = 0;     // 0 fused solve (one launch per frame), 1 launch chain (sequences whose solves reject steps)
    int solver_mode = 0, solver_K = 0;   // psfm_ctx_set_solver: 0 adaptive / 1 chain / 2 fused; K 0 = adaptive
    int64_t n_fused_ok = 0, n_fused_redone = 0, n_chain = 0;   // solves of the last psfm_track by how they ran
    int64_t n_resident = 0, n_iter_launches = 0;               // launches of the last call: resident solves, single trust-region iterations
    PsfmTrackDims* shard_dims = nullptr;   // psfm_shard_begin .. psfm_shard_finish
    bool shard_optimize = false;
    int solve_unroll = 6;   // iterations enqueued per frame without polling (adapted at checkpoints)
    PsfmBuf occ_own, occ2_own;           // occlusion maps of psfm_connect when the caller passes none
    PsfmBuf batch_tab, batch_ws, batch_fc;   // psfm_connect_batch (this context as the batch's owner): the table of sequences, packed
                                         // checkpoints, the table of stacks for flow_check
    void* host_batch = nullptr;          // ... and their pinned staging
    size_t host_batch_bytes = 0;
    void* host_batch2 = nullptr;
    size_t host_batch2_bytes = 0;
    int64_t batch_notrim_shape = 0;      // (h, w, sample_ratio, mode) whose sequences once outgrew launches trimmed to the lanes in use
    PsfmBuf win_ws;                      // psfm_window_sample / psfm_result_filter workspace
    PsfmBuf flt_ids, flt_birth, flt_len, flt_off, flt_xy;   // psfm_result_filter: the saved set (length >= traj_min_len), CSR
    int64_t flt_n_traj = 0, flt_n_points = 0;
    // psfm_traj_to_matches (psfm_matches.hip): keypoint / match tables of the saved set
    PsfmBuf mt_kp_off, mt_q, mt_pts, mt_kp_ind, mt_kp_xy, mt_moff, mt_keys, mt_rows, mt_gid, mt_pairs;
    int64_t mt_n_kp = 0, mt_n_m = 0, mt_n_pairs = 0;
    int mt_n_img = 0;
    hipStream_t side_stream = nullptr;   // flow_check of psfm_connect runs here, ahead of the frame loop
    hipStream_t copy_stream = nullptr;   // psfm_load_flo_stack: H2D copies out of the pinned ring
    hipStream_t copy_stream2 = nullptr;  // ... every second slot's copies (two SDMA engines; PSFM_FLO_COPY_STREAMS=1: one)
    std::vector<void*> ingest_slots;     // ... the ring (pinned, ingest_slot_bytes each) and the event behind the last copy out of every slot
    std::vector<hipEvent_t> ingest_events;
    size_t ingest_slot_bytes = 0;
    hipStream_t redo_stream = nullptr;   // psfm_connect_batch: a sequence that left the batch runs here, beside the others that did
    std::vector<psfm_solve_stats> solve_stats;
    PsfmProfiler prof;
    void* host_pinned = nullptr;  // small pinned staging block
    size_t host_pinned_bytes = 0;
};

// ---- launchers (psfm_track.hip) -------------------------------------------------------------
psfm_status psfm_launch_flow_check(const float* ff, const float* fb, int n_pairs, int h, int w, float thres,
                                   uint8_t* occ, float* err, hipStream_t s);
psfm_status psfm_launch_flow_check_bg(const float* ff, const float* fb, int n_pairs, int h, int w, float thres, uint8_t* occ,
                                      int n_blocks, hipStream_t s);
psfm_status psfm_launch_grid_sample(const float* map, int c, int h, int w, const double* xy, int64_t n,
                                    float* out, hipStream_t s);

struct PsfmTrackDims {
    int H, W, ratio, GW, GH, n_flows;
    int64_t G, cap, traj_cap;
    int shard_cap;          // trajectory records per shard (traj_cap = PSFM_NSHARD * shard_cap)
    int free_cap;           // free-lane stack entries per shard
    int nsh = PSFM_NSHARD;  // free-lane stacks in use (fewer on small grids)
    int shift_b, shift_d;   // key = death<<shift_d | birth<<shift_b | grid index
    float cw, ch;
    int nblk = 0, seg_cap = 0, spill_cap = 0;   // persistent loop: blocks, records per private segment, shared tail
    // track-sharded run (psfm_shard_*): births on grid points [g0, g0 + Gband) only; the two stamped blocked maps (+ the
    // survivor byte at offset G) live in a caller-owned buffer, `shard_pitch` bytes apart, that the ranks all-reduce
    int64_t g0 = 0, Gband = 0, shard_pitch = 0;
    uint8_t* shard_maps = nullptr;
};

psfm_status psfm_track_dims(psfm_ctx* c, int n_flows, int h, int w, int ratio, int64_t g_own, PsfmTrackDims& d);
psfm_status psfm_track_alloc(psfm_ctx* c, const PsfmTrackDims& d);
psfm_status psfm_launch_track_init(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s);
psfm_status psfm_launch_chain_step(psfm_ctx* c, const PsfmTrackDims& d, const float* flow, const uint8_t* occ,
                                   int frame, bool optimize, hipStream_t s);

// ---- batch: B same-shape sequences per launch (psfm_batch.hip; kernels beside their single-sequence forms) --------------------
#define PSFM_BATCH_MAX 64
struct PsfmBatchSeq;      // psfm_chain_step.h: one row of the batch table (chain-step arguments of frame 1 + strides)
struct PsfmFcSeq { const float* ff; const float* fb; uint8_t* occ; int n_pairs; int pad; };      // one stack of a batch for flow_check
bool psfm_flow_check_batch_ok(int h, int w, const void* ff, const void* fb, const void* occ);
psfm_status psfm_launch_flow_check_batch(const PsfmFcSeq* tab_dev, int n_seq, int pair0, int n_pairs, int h, int w, float thres, hipStream_t s);
void psfm_batch_fill_seq(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch, PsfmBatchSeq* row);
psfm_status psfm_launch_track_init_batch(const PsfmBatchSeq* tab_dev, int n_seq, int64_t cap_max, hipStream_t s);
psfm_status psfm_launch_chain_step_batch(psfm_ctx* owner, const PsfmBatchSeq* tab_dev, int n_seq, int ratio, int64_t cap_max, int frame,
                                         bool optimize, hipStream_t s);
// track_optimize rows (psfm_solver.hip: chain-step arguments + solver parameters of frame 1 + strides) and their launches
size_t psfm_batch_seq_opt_bytes(void);
psfm_status psfm_batch_fill_seq_opt(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch,
                                    const float* flows_f2, const uint8_t* occ_s2, void* row_host, hipStream_t s);
psfm_status psfm_launch_seq_batch(psfm_ctx* owner, const void* tab_dev, int n_seq, int ratio, int64_t cap_max, int n_launches, int launch_id0,
                                  hipStream_t s);
psfm_status psfm_launch_flush_batch(const void* tab_dev, int n_seq, int64_t cap_max, hipStream_t s);
psfm_status psfm_launch_batch_set_pc(const void* tab_dev, const int (*v)[4], int n_seq, hipStream_t s);
size_t psfm_batch_pack_row_bytes(int win);
psfm_status psfm_launch_batch_pack(const void* tab_dev, const int* lo, int n_seq, int win, char* out_dev, hipStream_t s);
// one segmented finalize for all sequences of a batch (psfm_finalize.hip): ONE host synchronisation, ONE sort.  dims[i] are the
// sequences' dimensions (same key format); the shared workspace is `own`'s, results land in every context's own res_* buffers
psfm_status psfm_finalize_batch(psfm_ctx* own, psfm_ctx* const* ctxs, const PsfmTrackDims* dims, int n_seq, hipStream_t s);

// ---- persistent frame loop (psfm_persist.hip) ---------------------------------------------------
int psfm_persist_max_blocks(psfm_ctx* c);
int psfm_persist_guests(void);   // LDS-resident extra lanes per block
// flows_b != NULL: the occlusion maps are computed inside the loop (fused flow_check) and written to `occ`
psfm_status psfm_launch_chain_persist(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ,
                                      int64_t occ_pitch, const float* flows_b, float thres, hipStream_t s);

// ---- finalize (psfm_finalize.hip) -----------------------------------------------------------
psfm_status psfm_finalize(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s);
// psfm_sort.hip: stable LSD radix sort of (32-bit key, int) pairs for the finalize (input in half0 when the pass count is even, else half1; result in half0)
int psfm_sort_pairs32_passes(unsigned end_bit);
psfm_status psfm_sort_pairs32(psfm_ctx* c, unsigned* k_half0, int* v_half0, unsigned* k_half1, int* v_half1, int64_t n, unsigned end_bit, hipStream_t s);
// after psfm_launch_chain_persist; *fallback = true (and PSFM_OK): the loop gave up, rerun with per-frame launches
psfm_status psfm_finalize_persist(psfm_ctx* c, const PsfmTrackDims& d, bool* fallback, hipStream_t s);

// ---- solver (psfm_solver.hip) ---------------------------------------------------------------
// In-place solve on the trajectory log for frame index f (positions at f-1, f, f+1): enqueue `unroll`
// iterations without host synchronisation / resume a solve that raised the stall flag.
psfm_status psfm_solve_frame_enqueue(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                     const float* flow02, const uint8_t* occ02, int frame, int unroll, hipStream_t s);
psfm_status psfm_solve_frame_resume(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                    const float* flow02, const uint8_t* occ02, int frame, psfm_solve_stats* st,
                                    int try_fused_k, bool chain_stalled, hipStream_t s);
psfm_status psfm_solve_frame_fused(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                   const float* flow02, const uint8_t* occ02, int frame, int K, hipStream_t s);
psfm_status psfm_launch_frame(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12, const float* flow02,
                              const uint8_t* occ, const uint8_t* occ02, int frame, int K, hipStream_t s, double* export_sums = nullptr);
psfm_status psfm_launch_seq(psfm_ctx* c, const PsfmTrackDims& d, const float* flows, const uint8_t* occ, int64_t occ_pitch,
                            const float* flows_f2, const uint8_t* occ_s2, int n_launches, int launch_id0, hipStream_t s);
psfm_status psfm_solve_flush(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s);
psfm_status psfm_solve_prepare(psfm_ctx* c, const PsfmTrackDims& d, hipStream_t s);   // every solver buffer of a sequence, up front
int psfm_solve_kmax(void);
int psfm_resident_blocks(psfm_ctx* c);     // co-resident blocks of the resident solve on the context's device
// track-sharded runs (psfm_shard.hip)
void psfm_shard_abandon(psfm_ctx* c);   // a run that was begun and never finished (context teardown)
psfm_status psfm_solve_export(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12, const float* flow02,
                              const uint8_t* occ02, int frame, int kind, int K, double* sums_out, hipStream_t s);
psfm_status psfm_solve_control(psfm_ctx* c, const PsfmTrackDims& d, int frame, int kind, int K, const double* totals, hipStream_t s,
                               int* stall_host = nullptr);
psfm_status psfm_solve_state(psfm_ctx* c, int* done, int* stall, psfm_solve_stats* st, hipStream_t s);
psfm_status psfm_solve_restore(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s);
psfm_status psfm_solve_writeback(psfm_ctx* c, const PsfmTrackDims& d, int frame, hipStream_t s);
// Batch API form (psfm_optimize_location).
size_t psfm_peer_area_bytes(void);
size_t psfm_peer_lead_offset(void);
int psfm_peer_leaders(int n_blocks);
int psfm_solve_blocks(psfm_ctx* c, const PsfmTrackDims& d);
psfm_status psfm_solve_frame_enqueue_peer(psfm_ctx* c, const PsfmTrackDims& d, const float* flow01, const float* flow12,
                                          const float* flow02, const uint8_t* occ02, int frame, unsigned epoch, hipStream_t s);
psfm_status psfm_launch_pc_eval(const double* uv12, const double* ref1, const double* ref2, const double* scale, const float* flow12,
                                int64_t n, int w, int h, double* res, double* jac, hipStream_t s);
psfm_status psfm_solve_batch(psfm_ctx* c, const double* uv12, const double* ref1, const double* ref2,
                             const double* scale, const float* flow12, int64_t n, int w, int h, double* out,
                             psfm_solve_stats* st, hipStream_t s);
