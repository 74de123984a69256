/*
 * psfm.h -- C ABI of libpsfm_hip.so: the MI355X (gfx950) implementation of
 * ParticleSfM's point-trajectory hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one interface of the
 * reference (paths relative to the reference root, bytedance/particle-sfm):
 *
 *   psfm_load_flo_stack     point_trajectory/utils.py:26-56         load_flows() / read_flo()
 *   psfm_flow_check         point_trajectory/utils.py:94-105        flow_check()
 *   psfm_grid_sample        point_trajectory/trajectory.py:25-37    grid_sample()
 *   psfm_optimize_location  point_trajectory/optimize/src/trajectory_optimize.cpp:30-96
 *                           (pybind entry: optimize/src/bindings.cc:31)
 *   psfm_track              point_trajectory/track.py:24-50          track()
 *                           point_trajectory/track_optimize.py:24-53 track_optimize()
 *                           (+ the IncrementalTrajectorySet / Trajectory bookkeeping they
 *                           drive: trajectory.py:98-194, optimize/src/trajectory_base.cpp:21-93)
 *   psfm_connect            point_trajectory/main_connect_point_trajectories.py:36-53 (flow_check + track[_optimize])
 *   psfm_connect_batch      the same for a batch of sequences: the loop of run_particlesfm.py:168-176
 *   psfm_result_*           the list of Trajectory objects those functions return and the
 *                           id / min-length rule of main_connect_point_trajectories.py:56-60
 *
 * Conventions
 *   - plain C, no C++/torch types.  Every data pointer is a DEVICE pointer
 *     (HBM of the context's GPU) unless the name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Kernels are
 *     enqueued on it; calls that must learn a data-dependent size (psfm_track) synchronise
 *     that stream before returning.
 *   - every function returns a psfm_status; on failure psfm_last_error() (thread local)
 *     describes it.  Nothing throws across this boundary and there is no CPU fallback:
 *     without a GPU every compute entry point fails with PSFM_ERR_HIP.
 *   - a psfm_ctx owns the device workspace (trajectory log, lane tables, results); it is
 *     not thread-safe, use one per host thread / stream.  The library keeps ONE piece of
 *     process-wide state: a shared/exclusive gate per device -- every entry point that launches
 *     work holds it shared, the persistent frame loop (psfm_ctx_set_chain_mode) holds it
 *     exclusively for its few milliseconds, because all of its blocks must be resident at once.
 *     Contexts on different devices never interact; last-error text is thread local.
 */
#ifndef PSFM_H_
#define PSFM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSFM_VERSION 141   /* round 5: psfm_connect_batch */

typedef enum psfm_status {
    PSFM_OK = 0,
    PSFM_ERR_ARG = 1,       /* bad argument */
    PSFM_ERR_HIP = 2,       /* HIP runtime error (no device, OOM, launch failure) */
    PSFM_ERR_CAPACITY = 3,  /* lane / trajectory tables too small: raise psfm_ctx_set_capacity and retry */
    PSFM_ERR_SOLVER = 4     /* the trust-region loop did not terminate (a solve Ceres would end in FAILURE is NOT an error: like the reference, the parameters stay as they came in and psfm_solve_stats.termination says 5) */
} psfm_status;

typedef struct psfm_ctx psfm_ctx;

/* Termination codes of the path-consistency solve (Ceres 2.0.0 TerminationType detail). */
enum {
    PSFM_TERM_FUNCTION_TOL = 0,
    PSFM_TERM_PARAMETER_TOL = 1,
    PSFM_TERM_GRADIENT_TOL = 2,
    PSFM_TERM_MAX_ITER = 3,
    PSFM_TERM_MIN_RADIUS = 4,
    PSFM_TERM_FAILURE = 5
};

typedef struct psfm_solve_stats {
    int32_t iterations;        /* trust-region iterations (excluding iteration 0) */
    int32_t successful_steps;
    int32_t termination;       /* PSFM_TERM_* */
    int32_t dogleg_nonGN;      /* iterations whose step was not the pure Gauss-Newton step */
    double initial_cost;
    double final_cost;
} psfm_solve_stats;

typedef struct psfm_track_info {
    int64_t n_traj;            /* all trajectories, index == id (full_trajs order) */
    int64_t n_points;          /* sum of their lengths */
    int64_t n_lanes_peak;      /* peak number of lanes used (<= lane capacity) */
    int64_t lane_capacity;
    int64_t solver_iterations; /* total trust-region iterations over all frames (track_optimize) */
    int32_t n_solves;
    int32_t chain_mode;        /* how the frame recurrence ran: 1 one launch per frame, 2 one persistent launch, 3 one launch per
                                  frame for a whole batch of sequences (psfm_connect_batch) */
} psfm_track_info;

const char* psfm_last_error(void);
int psfm_version(void);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int psfm_device_count(void);

psfm_status psfm_ctx_create(int device, psfm_ctx** out);
psfm_status psfm_ctx_destroy(psfm_ctx* ctx);

/* Lane table = lane_factor * (grid points); finished-trajectory table = max(traj_factor, n_flows/8) * (grid points).
 * Defaults 2.0 / 8.0.  psfm_track returns PSFM_ERR_CAPACITY when either overflows. */
psfm_status psfm_ctx_set_capacity(psfm_ctx* ctx, double lane_factor, double traj_factor);

/* How psfm_track / psfm_connect run the frame recurrence in track mode (flows_f2 == NULL):
 *   0 (default) by shape: ONE persistent launch for the whole sequence where that is the faster way on an MI355X
 *     (sample_ratio >= 2 and >= 100 k grid points; psfm_connect, which then also checks flow consistency inside that
 *     launch, additionally wants >= 400 k grid points and <= 6 pixels per grid point, e.g. 1080p at sample_ratio 2),
 *     one launch per frame elsewhere.  The persistent loop needs every lane of the stride-r grid resident at once
 *     (grid points <= 256 x resident blocks) and the device to itself; it hands over to per-frame launches by itself
 *     when it runs out of lanes or of patience at a barrier;
 *   1 one launch per frame always;  2 the persistent loop wherever it can run (waits for the device), per-frame elsewhere.
 * Results are identical in every mode.  track_optimize always uses one launch per frame (the solves sit in between). */
psfm_status psfm_ctx_set_chain_mode(psfm_ctx* ctx, int mode);

/* How psfm_track / psfm_connect run the path-consistency solve of a frame (track_optimize.py:49-50 ->
 * trajectory_optimize.cpp:74-82) -- the results do not depend on it:
 *   mode 0 (default) adaptive: the FUSED solve -- one launch per frame (the frame's chain step included) that speculates k
 *     trust-region iterations taking the Gauss-Newton step and being accepted, and replays Ceres' control flow over their
 *     sums; a solve that is not over after k accepted iterations gets a continuation launch (decided on the device) --
 *     while the solves of a sequence go that way; the launch CHAIN (one launch per trust-region iteration, any dogleg case,
 *     rejections) for windows of frames whose solves do not; a fused solve that meets a rejection / a dogleg interpolation /
 *     an invalid step is redone by the chain;
 *   mode 1 the chain always;  mode 2 the fused solve always (+ redo).
 *   k: iterations per fused launch, 0 = follow what the sequence needs (accepted steps + 1), at most 8. */
psfm_status psfm_ctx_set_solver(psfm_ctx* ctx, int mode, int k);
/* Solves of the last psfm_track / psfm_connect by how they ran: fused and done in one launch, fused then redone by
 * the chain, chain.  k_now: the adaptive iterations per fused launch after that sequence.  Any pointer may be NULL. */
psfm_status psfm_solver_counters(psfm_ctx* ctx, int64_t* fused, int64_t* fused_redone, int64_t* chain, int32_t* k_now);

/* Solves that reject steps run their whole trust-region loop as ONE resident launch whose blocks must all be on the device at once
 * (two 256-thread blocks per CU: psfm_resident_capacity, 512 on an MI355X).  By default a call only does that when it has the device
 * to itself (the exclusive gate): sequences processed concurrently by several host threads fall back to one launch per trust-region
 * iteration, 3-5x slower on such flows.  psfm_ctx_set_resident_budget(ctx, n) with n > 0 lets this context's resident solves use at
 * most n blocks while OTHER contexts run theirs -- the caller guarantees that the budgets of all contexts in flight on the device add
 * up to at most the capacity (e.g. 4 worker threads x 128).  A launch that does not become co-resident after all (other work holding
 * the block slots) gives up at its spin limit and is redone with launches.  n = 0: the default policy.
 * What a budget may change: a budget below the solve's own block count (lane capacity / 256) makes the solver's launches smaller, and
 * the f64 sums over the tracks (cost, step norms, the dogleg's inner products) are then added in another grouping.  Every DECISION of
 * the trust-region loop has been the same with and without a budget on all tests (iterations, accepted steps, terminations: tested
 * against the unbudgeted run and the oracle, tests/test_gpu_solver.py::test_budget_below_the_solves_block_count); positions agree to
 * rounding (<= 1e-9 px), not bit for bit.  Ids and lengths never depend on it. */
psfm_status psfm_ctx_set_resident_budget(psfm_ctx* ctx, int blocks);
psfm_status psfm_resident_capacity(psfm_ctx* ctx, int32_t* blocks);

/* Launches of the last psfm_track / psfm_connect / psfm_optimize_location on the context for the solves that did NOT go as the fused
 * solve speculates: resident launches (one per solve: the trust-region loop with the tracks' state on chip -- needs the device to
 * itself), how many of them gave up their hand-off, launches of ONE trust-region iteration each (the launch chain).  Which kernel a
 * measured solver time belongs to.  Any pointer may be NULL. */
psfm_status psfm_solver_launches(psfm_ctx* ctx, int64_t* resident, int64_t* giveups, int64_t* iterations);

/* utils.py:94-105.  flows_f, flows_b: (n_pairs,H,W,2) f32 stacks in the .flo-native interleaved
 * layout.  occ_out: (n_pairs,H,W) u8 0/1.  err_out: (n_pairs,H,W) f32 or NULL (the reference
 * pipeline never consumes it).  Bit-exact with the reference's torch-CPU arithmetic. */
psfm_status psfm_flow_check(psfm_ctx* ctx, const float* flows_f, const float* flows_b, int n_pairs, int h,
                            int w, float thres, uint8_t* occ_out, float* err_out, void* stream);

/* utils.py:26-56 (load_flows / read_flo) for a whole stack, straight into HBM: the n Middlebury .flo files `paths_host` (12-byte header
 * {f32 202021.25, i32 w, i32 h} + h*w interleaved (u,v) f32 -- the layout the kernels read), all of frame size w x h, into
 * dst (n,H,W,2) f32 on the device.  n_threads reader threads -> a ring of pinned staging buffers owned by the context -> asynchronous
 * H2D copies; returns when the last copy has completed (`stream` is only used to order the copies behind what the caller enqueued).
 * A missing / foreign / truncated file or a frame of another size is PSFM_ERR_ARG with the file named in psfm_last_error(). */
psfm_status psfm_load_flo_stack(psfm_ctx* ctx, const char* const* paths_host, int n, int h, int w, float* dst, int n_threads, void* stream);

/* trajectory.py:25-37 on an (H,W,C) f32 map, C in {1,2}; xy: (n,2) f64; out: (n,C) f32. */
psfm_status psfm_grid_sample(psfm_ctx* ctx, const float* map_hwc, int c, int h, int w, const double* xy,
                             int64_t n, float* out, void* stream);

/* trajectory_optimize.cpp:30-96.  uv12 (n,4), ref1 (n,2), ref2 (n,2), scale (n), out (n,4): f64;
 * flow12: (H,W,2) f32 (the f64 force-cast of the reference is exact).  stats_host may be NULL.
 * Synchronises `stream` (the iteration count is data dependent). */
psfm_status psfm_optimize_location(psfm_ctx* ctx, const double* uv12, const double* ref1, const double* ref2,
                                   const double* scale, const float* flow12, int64_t n, int w, int h,
                                   double* out, psfm_solve_stats* stats_host, void* stream);

/* optimize/src/path_consistency_cost.h:42-59 + linear_interpolation.h:97-123 (over ceres::Grid2D's clamp-to-edge GetValue): what
 * ceres::AutoDiffCostFunction<PathConsistencyError, 6, 4>::Evaluate returns for each of the n residual blocks of
 * trajectory_optimize.cpp:56-65 at uv12 -- residuals (n,6) and jacobians (n,6,4) row-major, f64; either may be NULL.  The arithmetic is
 * the one the solver kernels run (csrc/psfm_pc_core.h); exported so that it can be checked without a solve around it.  Asynchronous on
 * `stream`. */
psfm_status psfm_path_consistency_eval(psfm_ctx* ctx, const double* uv12, const double* ref1, const double* ref2,
                                       const double* scale, const float* flow12, int64_t n, int w, int h,
                                       double* residuals, double* jacobians, void* stream);

/* The record sort of the id assignment (the reference's trajectory ids are the rank of a track by (death step, birth
 * frame, birth grid index), the order trajectory.py:129-158 appends to full_trajs in): n (key, value) pairs on the device, keys sorted
 * ascending by their bits [0, end_bit), 1 <= end_bit <= 32, STABLE; in place.  The kernels are the ones psfm_track / psfm_connect run
 * on their records (csrc/psfm_sort.hip); exported so that the sort can be checked on its own.  n < 2^31.  Asynchronous on `stream`. */
psfm_status psfm_sort_records(psfm_ctx* ctx, uint32_t* keys, int32_t* values, int64_t n, int end_bit, void* stream);

/* track.py:24-50 when flows_f2 == NULL, track_optimize.py:24-53 otherwise.
 *   flows    (n_flows,H,W,2) f32      occ     (n_flows,H,W) u8
 *   flows_f2 (n_flows-1,H,W,2) f32    occ_s2  (n_flows-1,H,W) u8        (stride-2 stacks)
 * Runs the whole frame recurrence and the id assignment on the device; the result stays in the
 * context (HBM) until the next psfm_track on it.  info_host may be NULL.  Synchronises `stream`. */
psfm_status psfm_track(psfm_ctx* ctx, const float* flows, const uint8_t* occ, const float* flows_f2,
                       const uint8_t* occ_s2, int n_flows, int h, int w, int sample_ratio,
                       psfm_track_info* info_host, void* stream);

/* The compute part of the stage entry main_connect_point_trajectories.py:36-53 in one call: flow_check of the
 * stride-1 stacks (and of the stride-2 stacks when flows_f2 != NULL) followed by track / track_optimize.
 *   - track mode with the device to itself (see psfm_ctx_set_chain_mode): ONE persistent launch that computes the
 *     occlusion maps and runs the recurrence -- the blocks check flow consistency in the time they would otherwise wait
 *     at the frame barriers (a caller-provided `occ` takes part when H*W is a multiple of 128, else the stand-alone
 *     kernel fills it first);
 *   - otherwise the maps are produced on an internal side stream while the frame loop consumes them.
 *   flows_f, flows_b   (n_flows,H,W,2) f32      flows_f2, flows_b2  (n_flows-1,H,W,2) f32 or NULL
 *   occ, occ_s2        optional outputs (n_flows,H,W) / (n_flows-1,H,W) u8; NULL = kept in the context
 * Result access as for psfm_track.  Synchronises `stream`. */
psfm_status psfm_connect(psfm_ctx* ctx, const float* flows_f, const float* flows_b, const float* flows_f2,
                         const float* flows_b2, int n_flows, int h, int w, float thres, int sample_ratio,
                         uint8_t* occ, uint8_t* occ_s2, psfm_track_info* info_host, void* stream);

/* The loop of the reference's driver over a directory of sequences (run_particlesfm.py:168-176 -> connect_point_trajectory ->
 * main_connect_point_trajectories.py:36-53 per sequence) for n_seq sequences of the SAME frame size and sample ratio (any
 * lengths) in ONE call: psfm_connect for every one of them, with every frame launch covering the whole batch (block (x, y) = tile
 * x of sequence y).  A frame of a small sequence -- DAVIS, Sintel, ScanNet sizes -- is one dependent chain of memory round trips on
 * a few hundred of the device's block slots; a batch fills them.  One host synchronisation per window of 16 frames and one
 * segmented finalize (one sort) for all sequences.
 *   ctxs[i]      one context per sequence, all on one device, each given once; afterwards context i holds sequence i's result
 *                exactly as after psfm_connect (psfm_result_*, psfm_result_filter, psfm_traj_to_matches, psfm_window_sample work
 *                on it); ctxs[0] also owns the batch's shared workspace
 *   flows_f[i], flows_b[i]    (n_flows[i],H,W,2) f32      n_flows[i] >= 1
 *   flows_f2, flows_b2        NULL (track) or arrays of n_seq pointers to (n_flows[i]-1,H,W,2) stacks (track_optimize, every sequence)
 *   infos_host   n_seq entries or NULL
 * track_optimize: a sequence whose solves reject steps has them redone by the launch chain at the checkpoint, like psfm_track; when
 * a whole window of it is like that (or its context is set to the launch chain, psfm_ctx_set_solver mode 1) it leaves the batch
 * and is run alone by psfm_connect behind it.  Ids, lengths and every solver decision do not depend on the batching; track-mode positions are bit-identical, path-consistency
 * positions agree to rounding (<= 1e-9 px: a sequence that leaves the batch solves on a share of the block slots, see
 * psfm_ctx_set_resident_budget).  n_seq <= 64.  Synchronises `stream`. */
psfm_status psfm_connect_batch(psfm_ctx* const* ctxs, int n_seq, const float* const* flows_f, const float* const* flows_b,
                               const float* const* flows_f2, const float* const* flows_b2, const int* n_flows, int h, int w,
                               float thres, int sample_ratio, psfm_track_info* infos_host, void* stream);

/* Device-resident result of the last psfm_track: CSR over trajectories in id order.
 *   birth (n_traj) i32 first frame; len (n_traj) i32; off (n_traj+1) i64; xy (n_points,2) f64.
 * Trajectory i has times birth[i] .. birth[i]+len[i]-1 and points xy[off[i] .. off[i+1]). */
psfm_status psfm_result_device(psfm_ctx* ctx, const int32_t** birth, const int32_t** len,
                               const int64_t** off, const double** xy);

/* Copy the result into caller-provided HOST buffers (any may be NULL to skip). */
psfm_status psfm_result_copy(psfm_ctx* ctx, int32_t* birth_host, int32_t* len_host, int64_t* off_host,
                             double* xy_host, void* stream);

/* Per-solve statistics of the last psfm_track (track_optimize mode): up to `max_n` entries. */
psfm_status psfm_result_solve_stats(psfm_ctx* ctx, psfm_solve_stats* stats_host, int32_t max_n,
                                    int32_t* n_out);

/* The saved trajectory set of main_connect_point_trajectories.py:56-61 -- trajectories of length >= traj_min_len, ids =
 * their indices in the full list -- compacted in HBM from the result of the last psfm_track / psfm_connect, so that only
 * what is kept crosses PCIe (the host-side filter is a boolean gather over every point).
 *   psfm_result_filter        builds the filtered CSR in the context, returns its sizes; synchronises `stream`
 *   psfm_result_filtered_copy copies it to HOST buffers: ids (k) i32, birth (k) i32, len (k) i32, off (k+1) i64,
 *                             xy (n_points,2) f64; any may be NULL */
psfm_status psfm_result_filter(psfm_ctx* ctx, int traj_min_len, int64_t* n_traj_host, int64_t* n_points_host, void* stream);
psfm_status psfm_result_filtered_copy(psfm_ctx* ctx, int32_t* ids_host, int32_t* birth_host, int32_t* len_host,
                                      int64_t* off_host, double* xy_host, void* stream);

/* The motion-segmentation window tensors (motion_seg/load_cut_seq.py:60-89) from the device-resident result of the last
 * psfm_track / psfm_connect -- TrajectorySet::sample_inside_window (optimize/src/trajectory_base.cpp:127-185) for the
 * contiguous window [frame0, frame0 + n_frames) plus the resize / normalise of motion_seg/core/dataset/data_utils.py:74-89:
 *   trajectories of length >= traj_min_len (the saved set, main_connect_point_trajectories.py:56-61) with >= min_length
 *   observations inside the window, ascending id; more than max_num_tracks -> a random subset in shuffled order
 *   (seeded; the reference shuffles unseeded).
 * Outputs, all DEVICE pointers, any may be NULL (all NULL = count only); `capacity` = rows the buffers can hold:
 *   ids_out (K) i32   xy_raw (K,n_frames,2) f64 zero-padded   mask_absent (K,n_frames) f64, 1.0 where the trajectory
 *   has no point (load_cut_seq's `1 - masks`)   xy_norm (K,n_frames,2) f64 = clip(xy / (raw/in) / in, 0, 1)
 * *k_host = K.  Synchronises `stream`. */
psfm_status psfm_window_sample(psfm_ctx* ctx, int frame0, int n_frames, int traj_min_len, int min_length,
                               int64_t max_num_tracks, uint64_t seed, int raw_h, int raw_w, int in_h, int in_w,
                               int64_t capacity, int32_t* ids_out, double* xy_raw, double* xy_norm, double* mask_absent,
                               int64_t* k_host, void* stream);

/* sfm/matches_from_flow.py:51-118 (traj_to_matches) from the saved set that psfm_result_filter left in HBM -- the
 * reference's per-trajectory Python loops as index arithmetic on the device (no track.npy round trip):
 *   keypoints  image i lists the kept points observed in frame i in trajectory (id) order (:67-81); labels_dev: optional
 *              (n_points of the saved set) u8 DEVICE array, 1 = dynamic point, dropped (remove_dynamic, :71-74), NULL = keep all
 *   matches    point j of a trajectory with n kept points pairs with every other point when n <= sample_k (20 in the
 *              reference, :52), else with the points at k * (n / sample_k), itself skipped (:83-101); a match is the row
 *              [keypoint index of j, keypoint index of the target] under the image pair (frame of j, frame of the target)
 * n_img = number of images (every frame index must be < n_img).  Returns the table sizes; synchronises `stream`.
 * psfm_matches_copy copies the tables to HOST buffers (any may be NULL):
 *   kp_off (n_img+1) i64, kp_xy (n_kp,2) f64 -- keypoints of image i = kp_xy[kp_off[i] .. kp_off[i+1])
 *   pair_key (n_pairs) i64 = src_image * n_img + tgt_image, ascending;  pair_off (n_pairs+1) i64 into rows;
 *   pair_first (n_pairs) i64 = position of the pair's first match in the reference's loop order (its dict order);
 *   rows (n_matches,2) i32, inside a pair in the reference's loop order. */
psfm_status psfm_traj_to_matches(psfm_ctx* ctx, int n_img, int sample_k, const uint8_t* labels_dev, int64_t* n_kp_host,
                                 int64_t* n_matches_host, int64_t* n_pairs_host, void* stream);
psfm_status psfm_matches_copy(psfm_ctx* ctx, int64_t* kp_off_host, double* kp_xy_host, int64_t* pair_key_host,
                              int64_t* pair_off_host, int64_t* pair_first_host, int32_t* rows_host, void* stream);

/* ONE sequence over several processes / GPUs, exactly (psfm_dist.connect_sharded drives these).
 * The tracks are split by the row band of the stride-r grid they are born on: this process owns the births on grid points
 * [g0, g1) and runs every frame for its tracks.  `maps`: caller-owned DEVICE buffer of 2 x map_pitch bytes (map_pitch >=
 * G + 1); after psfm_shard_step(frame) the slab (frame & 1) holds, stamped, the grid points within distance sample_ratio of
 * one of this process's surviving tracks (+ byte G: a track survived) -- the caller all-reduces (max) those G + 1 bytes over
 * the ranks before the next step reads them.  The solve of a frame runs as export (this process's sums: kind 0 the fused
 * solve with k iterations -> k x 13 doubles, 1 / 2 the launch chain's first / next iteration -> 13) -> the caller combines the
 * ranks' sums in rank order -> control on the totals (same numbers, same decision on every rank; synchronises and
 * reports done / redo / statistics).  redo: a fused solve met something it did not speculate -> restore, then the chain
 * (export 1, control, export 2, control, ... until done, write-back).  psfm_shard_finish leaves the own trajectories as the
 * usual result (psfm_result_copy), sorted by (last valid time, birth frame, birth grid index); ids over all ranks follow
 * from those keys.  No collective is issued by the library. */
psfm_status psfm_shard_begin(psfm_ctx* ctx, int n_flows, int h, int w, int sample_ratio, int64_t g0, int64_t g1, int optimize,
                             uint8_t* maps, int64_t map_pitch, void* stream);
psfm_status psfm_shard_step(psfm_ctx* ctx, const float* flow, const uint8_t* occ, int frame, void* stream);
psfm_status psfm_shard_solve_export(psfm_ctx* ctx, const float* flow01, const float* flow12, const float* flow02,
                                    const uint8_t* occ02, int frame, int kind, int k, double* sums_out, void* stream);
/* psfm_shard_step(frame) + psfm_shard_solve_export(frame, kind 0, k) as ONE launch (track_optimize.py:31-50 for the own tracks:
 * the chain step of the frame and the fused solve of its tracks, sums exported); frame >= 1, flow12 = the frame's own forward
 * flow, occ = its occlusion map.  The frame's marks are exchanged behind it like psfm_shard_step's.  sums_out NULL (a shard that is
 * the whole sequence only, see psfm_shard_solve_local): nothing is exported, the launch runs the control step on its own totals
 * and no psfm_shard_solve_control* call follows; a solve that did not go as speculated raises the device-side stall flag, which
 * reaches psfm_shard_peek_stall with every 8th frame (and psfm_shard_window_state at once). */
psfm_status psfm_shard_frame(psfm_ctx* ctx, const float* flow01, const float* flow12, const float* flow02, const uint8_t* occ,
                             const uint8_t* occ02, int frame, int k, double* sums_out, void* stream);
psfm_status psfm_shard_solve_control(psfm_ctx* ctx, int frame, int kind, int k, const double* totals, int32_t* done_host,
                                     int32_t* redo_host, psfm_solve_stats* stats_host, void* stream);
/* The fused path without a host round trip per solve: psfm_shard_solve_control_async runs the control step of a fused export
 * (kind 0, k iterations) and returns; psfm_shard_window_state synchronises once for a whole window of frames -- the stalled
 * frame (a solve that did not go as speculated; every later launch of the context has been a no-op since) or -1, and the
 * statistics of the window's solves.  The caller then redoes the stalled solve (restore + the per-iteration protocol) and
 * re-runs the frames behind it. */
psfm_status psfm_shard_solve_control_async(psfm_ctx* ctx, int frame, int k, const double* totals, void* stream);
psfm_status psfm_shard_window_state(psfm_ctx* ctx, int f_lo, int f_hi, psfm_solve_stats* stats_host, int32_t* stalled_frame,
                                    void* stream);
/* The redo of a stalled solve without a host round trip per trust-region iteration: psfm_shard_solve_control_chain_async enqueues the
 * control step behind an export of kind 1 / 2 and returns; the caller enqueues a batch of rounds (export 2 -> exchange -> control)
 * ahead and asks once per batch with psfm_shard_solve_poll (synchronises).  Rounds behind the one that ended the solve are no-ops. */
psfm_status psfm_shard_solve_control_chain_async(psfm_ctx* ctx, int frame, int kind, const double* totals, void* stream);
psfm_status psfm_shard_solve_poll(psfm_ctx* ctx, int32_t* done_host, psfm_solve_stats* stats_host, void* stream);
/* A shard that is the WHOLE sequence (g0 = 0, g1 = every grid point: ONE rank, the windowed engine for one long sequence on one GPU;
 * PSFM_ERR_ARG for a band): its solves need no exchange, so solves whose steps get rejected run like psfm_connect's
 * (track_optimize.py:49-50 -> trajectory_optimize.cpp:74-82) instead of export -> exchange -> control once per trust-region
 * iteration.  psfm_shard_solve_local ENQUEUES the solve of `frame` behind psfm_shard_step(frame): the resident solve -- ONE launch per
 * solve, inside the context's resident budget (psfm_ctx_set_resident_budget; without one: `unroll` launches of one iteration each);
 * a solve that is not done behind its launches raises the device-side stall flag (psfm_shard_window_state).
 * psfm_shard_solve_redo_local redoes a stalled solve to termination and writes it back (synchronises; chain_stalled: the solve had
 * been enqueued by psfm_shard_solve_local, not by a fused export).  Same sums in the same order as the exchange form: same positions. */
psfm_status psfm_shard_solve_local(psfm_ctx* ctx, const float* flow01, const float* flow12, const float* flow02, const uint8_t* occ02,
                                   int frame, int unroll, void* stream);
psfm_status psfm_shard_solve_redo_local(psfm_ctx* ctx, const float* flow01, const float* flow12, const float* flow02,
                                        const uint8_t* occ02, int frame, int chain_stalled, psfm_solve_stats* stats_host, void* stream);
/* ONE solve over several ranks with no host and no collective library in its loop (csrc/psfm_shard.hip, csrc/psfm_solver.hip: PcPeers):
 * every rank runs the resident solve of trajectory_optimize.cpp:74-82 on its own tracks, and the second hop of the launch's all-reduce
 * writes each leader's sums into EVERY rank's granule area through peer-mapped pointers (IPC handles between processes -- P2P over
 * xGMI between GPUs -- plain pointers between host threads of one process).  Set-up once per context: psfm_shard_peer_area (this rank's
 * area + its 64-byte IPC handle), psfm_shard_peer_open (another process's area), psfm_shard_peer_connect (world <= 8, rank, every rank's
 * area as this process addresses it, every rank's launch size from psfm_shard_solve_blocks; 0 blocks anywhere = PSFM_ERR_ARG: keep the
 * exchange form).  Per frame psfm_shard_solve_peer(frame, epoch) ENQUEUES this rank's launch; all ranks pass the same epoch (a counter
 * they advance together, 1 .. 2^20 - 1, never reused on an area: psfm_shard_peer_epoch).  A launch that gives up (a rank missing, not co-resident) raises the stall flag on every rank --
 * psfm_shard_window_state reports it and the caller redoes that solve with psfm_shard_solve_export / _control. */
psfm_status psfm_shard_peer_area(psfm_ctx* ctx, void** area_dev, void* ipc_handle_64, void* stream);
psfm_status psfm_shard_peer_open(psfm_ctx* ctx, const void* ipc_handle_64, int peer_rank, void** mapped);
/* The last epoch a launch of this context tagged its area's granules with.  A driver starts a run from the maximum over the ranks (an
 * area outlives the engine object that drove it: starting over would let stale granules match) and resets every rank's area
 * (reset != 0, between two collectives) before the 20-bit count wraps. */
psfm_status psfm_shard_peer_epoch(psfm_ctx* ctx, uint32_t* last_epoch, int reset, void* stream);
psfm_status psfm_shard_peer_connect(psfm_ctx* ctx, int world, int rank, void* const* areas, const int32_t* n_blocks);
psfm_status psfm_shard_solve_blocks(psfm_ctx* ctx, int32_t* n_blocks);
psfm_status psfm_shard_solve_peer(psfm_ctx* ctx, const float* flow01, const float* flow12, const float* flow02, const uint8_t* occ02,
                                  int frame, uint32_t epoch, void* stream);

/* the stall flag as of the last control step the device has completed, without synchronising (-1: none) */
psfm_status psfm_shard_peek_stall(psfm_ctx* ctx, int32_t* stalled_frame);
psfm_status psfm_shard_solve_restore(psfm_ctx* ctx, int frame, void* stream);
psfm_status psfm_shard_solve_writeback(psfm_ctx* ctx, int frame, const psfm_solve_stats* stats, void* stream);
psfm_status psfm_shard_solve_record(psfm_ctx* ctx, const psfm_solve_stats* stats);
psfm_status psfm_shard_finish(psfm_ctx* ctx, psfm_track_info* info_host, void* stream);
/* keys_dev (n_traj) i64 DEVICE: (last valid time << 47) | (birth frame << 31) | birth grid index of every trajectory of the
 * result, ascending -- the order key of full_trajs (SURVEY a-17); a trajectory's id over all ranks = its key's rank. */
psfm_status psfm_result_keys(psfm_ctx* ctx, int sample_ratio, int w, int64_t* keys_dev, void* stream);

/* Per-kernel device time of the last psfm_track / psfm_flow_check when profiling is enabled with
 * psfm_ctx_set_profiling(ctx, 1): HIP events recorded on the launch stream around every launch of
 * the named kernel family (enable = N > 1: only every N-th per-frame chain_step launch is timed, which keeps the
 * event overhead out of a throughput measurement; a persistent launch is always timed and counts as ONE launch).
 * kind: 0 flow_check, 1 chain_step, 2 respawn, 3 solver, 4 finalize.
 * Returns the accumulated milliseconds and the number of launches. */
psfm_status psfm_ctx_set_profiling(psfm_ctx* ctx, int enable);
psfm_status psfm_profile_get(psfm_ctx* ctx, int kind, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PSFM_H_ */
