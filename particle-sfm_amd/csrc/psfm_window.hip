// psfm_window.hip -- SURVEY f-4: the motion-segmentation window tensors straight from the device-resident result.
//
// Reference: motion_seg/load_cut_seq.py:60-89 cuts a sequence into windows and, per window, calls
// TrajectorySet::sample_inside_window (optimize/src/trajectory_base.cpp:127-185): trajectories of the saved set (ids
// and the min-length filter of main_connect_point_trajectories.py:56-61) with at least `min_length` observations on
// the window's frames, in ascending id order, as zero-padded (K,L) x / y arrays + a presence mask; more than
// `max_num_tracks` -> a random subset (std::random_shuffle, unseeded in the reference; seeded here).  load_cut_seq then
// resizes the coordinates to the network input size and normalises them to [0,1]
// (motion_seg/core/dataset/data_utils.py:74-89).  Here the windows are contiguous frame ranges (what load_cut_seq
// passes) and everything is computed from the CSR result that psfm_track / psfm_connect left in HBM: no track.npy round
// trip, no per-trajectory Python objects.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include "psfm_device.h"
#include "psfm_internal.h"

#define PW_BLOCK 256

// observations of trajectory (birth, len) on frames [f0, f0 + L)
__device__ __forceinline__ int psfm_overlap(int birth, int len, int f0, int L)
{
    const int lo = birth > f0 ? birth : f0;
    const int hi = (birth + len) < (f0 + L) ? (birth + len) : (f0 + L);
    return hi > lo ? hi - lo : 0;
}

__global__ __launch_bounds__(PW_BLOCK) void psfm_window_flag_kernel(const int* __restrict__ birth, const int* __restrict__ len,
                                                                  int64_t n, int f0, int L, int traj_min_len, int min_length,
                                                                  uint8_t* __restrict__ flag, int* __restrict__ iota)
{
    const int64_t i = (int64_t)blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int ln = len[i];
    const int ov = psfm_overlap(birth[i], ln, f0, L);
    flag[i] = (uint8_t)((ln >= traj_min_len) & (ov > 0) & (ov >= min_length));
    iota[i] = (int)i;
}

// random subset: key = hash(seed, id); the smallest `max_num` keys win, in key order (a seeded shuffle)
__global__ __launch_bounds__(PW_BLOCK) void psfm_window_hash_kernel(const int* __restrict__ ids, int64_t k, unsigned long long seed,
                                                                  unsigned long long* __restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= k) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(ids[i] + 1);   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    keys[i] = z ^ (z >> 31);
}

struct PsfmWindowArgs {
    const int* ids; const int* birth; const int* len; const int64_t* off; const double2* xy;
    int64_t K; int f0, L;
    double ratio_w, ratio_h, in_w, in_h; int normalise;
    int* ids_out; double2* xy_raw; double2* xy_norm; double* mask_absent;
};

// one thread per (track k, window frame j); j fastest: consecutive threads read consecutive points of a trajectory
__global__ __launch_bounds__(PW_BLOCK) void psfm_window_gather_kernel(PsfmWindowArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * PW_BLOCK + threadIdx.x;
    if (e >= a.K * a.L) return;
    const int64_t k = e / a.L;
    const int j = (int)(e - k * a.L);
    const int id = a.ids[k];
    const int t = a.f0 + j, b = a.birth[id];
    const bool present = (t >= b) & (t < b + a.len[id]);
    double2 p = make_double2(0.0, 0.0);
    if (present) p = a.xy[a.off[id] + (t - b)];
    if (j == 0 && a.ids_out) a.ids_out[k] = id;
    if (a.xy_raw) a.xy_raw[e] = p;
    if (a.mask_absent) a.mask_absent[e] = present ? 0.0 : 1.0;          // load_cut_seq.py:67,81: (1 - masks).astype(float)
    if (a.xy_norm) {
        // data_utils.py:74-89: pt /= (raw / target); pt /= target; clip to [0,1] -- the same two f64 divisions, in order
        double x = p.x / a.ratio_w, y = p.y / a.ratio_h;
        x = x / a.in_w; y = y / a.in_h;
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
        a.xy_norm[e] = make_double2(x, y);
    }
}

extern "C" psfm_status psfm_window_sample(psfm_ctx* c, int frame0, int n_frames, int traj_min_len, int min_length,
                                          int64_t max_num_tracks, uint64_t seed, int raw_h, int raw_w, int in_h, int in_w,
                                          int64_t capacity, int32_t* ids_out, double* xy_raw, double* xy_norm,
                                          double* mask_absent, int64_t* k_host, void* stream)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    if (n_frames < 1 || max_num_tracks < 0 || capacity < 0 || !k_host || (xy_norm && (raw_h < 1 || raw_w < 1 || in_h < 1 || in_w < 1))) {
        psfm_set_error("psfm_window_sample: bad argument (n_frames=%d max_num_tracks=%lld capacity=%lld)", n_frames,
                       (long long)max_num_tracks, (long long)capacity);
        return PSFM_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = c->res_n_traj;
    *k_host = 0;
    if (n == 0) return PSFM_OK;
    psfm_status st;
    // workspace: flags (n u8) | iota (n i32) | selected ids (n i32) | count (8 B) | hash keys (2n u64) | shuffled ids (n i32)
    const size_t o_flag = 0, o_iota = ((size_t)n + 255) / 256 * 256, o_sel = o_iota + 4 * (size_t)n, o_cnt = o_sel + 4 * (size_t)n,
                 o_keys = o_cnt + 256, o_ids2 = o_keys + 16 * (size_t)n, total = o_ids2 + 4 * (size_t)n;
    if ((st = c->win_ws.ensure(total)) != PSFM_OK) return st;
    char* ws = (char*)c->win_ws.p;
    uint8_t* flag = (uint8_t*)(ws + o_flag);
    int* iota = (int*)(ws + o_iota);
    int* sel = (int*)(ws + o_sel);
    size_t* d_cnt = (size_t*)(ws + o_cnt);
    unsigned long long* keys = (unsigned long long*)(ws + o_keys);
    int* ids2 = (int*)(ws + o_ids2);
    hipLaunchKernelGGL(psfm_window_flag_kernel, dim3((unsigned)((n + PW_BLOCK - 1) / PW_BLOCK)), dim3(PW_BLOCK), 0, s,
                       c->res_birth.as<int>(), c->res_len.as<int>(), n, frame0, n_frames, traj_min_len, min_length, flag, iota);
    PSFM_HIP(hipGetLastError());
    size_t tmp = 0;
    PSFM_HIP(rocprim::select(nullptr, tmp, iota, flag, sel, d_cnt, (size_t)n, s));
    if ((st = c->sort_tmp.ensure(tmp)) != PSFM_OK) return st;
    PSFM_HIP(rocprim::select(c->sort_tmp.p, tmp, iota, flag, sel, d_cnt, (size_t)n, s));
    size_t* h_cnt = (size_t*)((char*)c->host_pinned + 256);
    PSFM_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    int64_t K = (int64_t)*h_cnt;
    const int* ids = sel;
    if (K > max_num_tracks) {   // trajectory_base.cpp:150-153, with a seed
        hipLaunchKernelGGL(psfm_window_hash_kernel, dim3((unsigned)((K + PW_BLOCK - 1) / PW_BLOCK)), dim3(PW_BLOCK), 0, s, sel, K,
                           (unsigned long long)seed, keys);
        PSFM_HIP(hipGetLastError());
        size_t tmp2 = 0;
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, tmp2, keys, keys + n, sel, ids2, (size_t)K, 0u, 64u, s));
        if ((st = c->sort_tmp.ensure(tmp2)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, tmp2, keys, keys + n, sel, ids2, (size_t)K, 0u, 64u, s));
        ids = ids2;
        K = max_num_tracks;
    }
    *k_host = K;
    if (K == 0 || (!ids_out && !xy_raw && !xy_norm && !mask_absent)) return PSFM_OK;   // count only
    if (K > capacity) {
        psfm_set_error("psfm_window_sample: %lld trajectories selected, output capacity %lld", (long long)K, (long long)capacity);
        return PSFM_ERR_CAPACITY;
    }
    PsfmWindowArgs a;
    a.ids = ids; a.birth = c->res_birth.as<int>(); a.len = c->res_len.as<int>(); a.off = c->res_off.as<int64_t>();
    a.xy = c->res_xy.as<double2>();
    a.K = K; a.f0 = frame0; a.L = n_frames;
    a.normalise = xy_norm != nullptr;
    a.ratio_w = a.normalise ? (double)raw_w / (double)in_w : 1.0;
    a.ratio_h = a.normalise ? (double)raw_h / (double)in_h : 1.0;
    a.in_w = (double)in_w; a.in_h = (double)in_h;
    a.ids_out = ids_out; a.xy_raw = (double2*)xy_raw; a.xy_norm = (double2*)xy_norm; a.mask_absent = mask_absent;
    const int64_t total_e = K * (int64_t)n_frames;
    hipLaunchKernelGGL(psfm_window_gather_kernel, dim3((unsigned)((total_e + PW_BLOCK - 1) / PW_BLOCK)), dim3(PW_BLOCK), 0, s, a);
    PSFM_HIP(hipGetLastError());
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}

// ------------------------------------------------------------------------------------------------
// The saved trajectory set (main_connect_point_trajectories.py:56-61): trajectories of length >= traj_min_len, ids =
// indices into the full list.  Filtering on the host means a boolean gather over ~5e7 points; here the CSR is
// compacted in HBM (flag -> select -> scan -> one wave per kept trajectory copies its run) and only what is kept
// crosses PCIe.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PW_BLOCK) void psfm_filter_flag_kernel(const int* __restrict__ len, int64_t n, int min_len,
                                                                  uint8_t* __restrict__ flag, int* __restrict__ iota)
{
    const int64_t i = (int64_t)blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i >= n) return;
    flag[i] = (uint8_t)(len[i] >= min_len);
    iota[i] = (int)i;
}
__global__ __launch_bounds__(PW_BLOCK) void psfm_filter_meta_kernel(const int* __restrict__ ids, int64_t k, const int* __restrict__ birth,
                                                                  const int* __restrict__ len, int* __restrict__ birth_out,
                                                                  int* __restrict__ len_out, int64_t* __restrict__ len64)
{
    const int64_t i = (int64_t)blockIdx.x * PW_BLOCK + threadIdx.x;
    if (i > k) return;
    if (i == k) { len64[i] = 0; return; }
    const int id = ids[i];
    birth_out[i] = birth[id];
    len_out[i] = len[id];
    len64[i] = (int64_t)len[id];
}
__global__ __launch_bounds__(PW_BLOCK) void psfm_filter_copy_kernel(const int* __restrict__ ids, int64_t k, const int* __restrict__ len,
                                                                  const int64_t* __restrict__ off, const int64_t* __restrict__ off_out,
                                                                  const double2* __restrict__ xy, double2* __restrict__ xy_out)
{
    const int lane = threadIdx.x & (PSFM_WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * PW_BLOCK + threadIdx.x) / PSFM_WAVE, nw = (int64_t)gridDim.x * (PW_BLOCK / PSFM_WAVE);
    for (int64_t t = wave; t < k; t += nw) {
        const int id = ids[t];
        const int n = len[id];
        const int64_t src = off[id], dst = off_out[t];
        for (int j = lane; j < n; j += PSFM_WAVE) xy_out[dst + j] = xy[src + j];
    }
}

extern "C" psfm_status psfm_result_filter(psfm_ctx* c, int traj_min_len, int64_t* n_traj_host, int64_t* n_points_host, void* stream)
{
    if (!c || !n_traj_host || !n_points_host) { psfm_set_error("psfm_result_filter: NULL argument"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = c->res_n_traj;
    *n_traj_host = 0; *n_points_host = 0;
    c->flt_n_traj = c->flt_n_points = 0;
    if (n == 0) return PSFM_OK;
    psfm_status st;
    const size_t o_iota = ((size_t)n + 255) / 256 * 256, o_cnt = o_iota + 4 * (size_t)n, total = o_cnt + 256;
    if ((st = c->win_ws.ensure(total)) != PSFM_OK) return st;
    if ((st = c->flt_ids.ensure(4 * (size_t)n)) != PSFM_OK) return st;
    char* ws = (char*)c->win_ws.p;
    uint8_t* flag = (uint8_t*)ws;
    int* iota = (int*)(ws + o_iota);
    size_t* d_cnt = (size_t*)(ws + o_cnt);
    hipLaunchKernelGGL(psfm_filter_flag_kernel, dim3((unsigned)((n + PW_BLOCK - 1) / PW_BLOCK)), dim3(PW_BLOCK), 0, s,
                       c->res_len.as<int>(), n, traj_min_len, flag, iota);
    PSFM_HIP(hipGetLastError());
    size_t tmp = 0;
    PSFM_HIP(rocprim::select(nullptr, tmp, iota, flag, c->flt_ids.as<int>(), d_cnt, (size_t)n, s));
    if ((st = c->sort_tmp.ensure(tmp)) != PSFM_OK) return st;
    PSFM_HIP(rocprim::select(c->sort_tmp.p, tmp, iota, flag, c->flt_ids.as<int>(), d_cnt, (size_t)n, s));
    size_t* h_cnt = (size_t*)((char*)c->host_pinned + 256);
    PSFM_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(size_t), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    const int64_t k = (int64_t)*h_cnt;
    c->flt_n_traj = k;
    *n_traj_host = k;
    if (k == 0) return PSFM_OK;
    if ((st = c->flt_birth.ensure(4 * (size_t)k)) != PSFM_OK) return st;
    if ((st = c->flt_len.ensure(4 * (size_t)k)) != PSFM_OK) return st;
    if ((st = c->flt_off.ensure(8 * (size_t)(k + 1))) != PSFM_OK) return st;
    if ((st = c->scan_tmp.ensure(8 * (size_t)(k + 1))) != PSFM_OK) return st;
    hipLaunchKernelGGL(psfm_filter_meta_kernel, dim3((unsigned)((k + 1 + PW_BLOCK - 1) / PW_BLOCK)), dim3(PW_BLOCK), 0, s,
                       c->flt_ids.as<int>(), k, c->res_birth.as<int>(), c->res_len.as<int>(), c->flt_birth.as<int>(),
                       c->flt_len.as<int>(), c->scan_tmp.as<int64_t>());
    PSFM_HIP(hipGetLastError());
    size_t scan_bytes = 0;
    PSFM_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(k + 1),
                                     rocprim::plus<int64_t>(), s));
    if ((st = c->sort_tmp.ensure(scan_bytes)) != PSFM_OK) return st;
    PSFM_HIP(rocprim::exclusive_scan(c->sort_tmp.p, scan_bytes, c->scan_tmp.as<int64_t>(), c->flt_off.as<int64_t>(), (int64_t)0,
                                     (size_t)(k + 1), rocprim::plus<int64_t>(), s));
    int64_t* h_np = (int64_t*)((char*)c->host_pinned + 264);
    PSFM_HIP(hipMemcpyAsync(h_np, c->flt_off.as<int64_t>() + k, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    const int64_t np_keep = *h_np;
    c->flt_n_points = np_keep;
    *n_points_host = np_keep;
    if ((st = c->flt_xy.ensure(16 * (size_t)(np_keep > 0 ? np_keep : 1))) != PSFM_OK) return st;
    hipLaunchKernelGGL(psfm_filter_copy_kernel, dim3(4096), dim3(PW_BLOCK), 0, s, c->flt_ids.as<int>(), k, c->res_len.as<int>(),
                       c->res_off.as<int64_t>(), c->flt_off.as<int64_t>(), c->res_xy.as<double2>(), c->flt_xy.as<double2>());
    PSFM_HIP(hipGetLastError());
    return PSFM_OK;
}

extern "C" psfm_status psfm_result_filtered_copy(psfm_ctx* c, int32_t* ids_host, int32_t* birth_host, int32_t* len_host,
                                                 int64_t* off_host, double* xy_host, void* stream)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t k = c->flt_n_traj, np_keep = c->flt_n_points;
    if (k > 0) {
        if (ids_host) PSFM_HIP(hipMemcpyAsync(ids_host, c->flt_ids.p, 4 * (size_t)k, hipMemcpyDeviceToHost, s));
        if (birth_host) PSFM_HIP(hipMemcpyAsync(birth_host, c->flt_birth.p, 4 * (size_t)k, hipMemcpyDeviceToHost, s));
        if (len_host) PSFM_HIP(hipMemcpyAsync(len_host, c->flt_len.p, 4 * (size_t)k, hipMemcpyDeviceToHost, s));
        if (off_host) PSFM_HIP(hipMemcpyAsync(off_host, c->flt_off.p, 8 * (size_t)(k + 1), hipMemcpyDeviceToHost, s));
        if (xy_host && np_keep > 0) PSFM_HIP(hipMemcpyAsync(xy_host, c->flt_xy.p, 16 * (size_t)np_keep, hipMemcpyDeviceToHost, s));
    } else if (off_host) {
        off_host[0] = 0;
    }
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}
