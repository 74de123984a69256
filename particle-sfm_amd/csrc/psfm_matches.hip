// psfm_matches.hip -- SURVEY f-3: sfm/matches_from_flow.py:51-118 (traj_to_matches) on the device.
//
// The reference walks every trajectory of track.npy in Python: each kept point (labels == 0 when remove_dynamic) becomes a
// keypoint of its frame's image -- keypoint index = how many points that image already has, i.e. its rank among the points
// of that frame in trajectory (id) order (:67-81) -- and point j of a trajectory with n kept points is matched with every
// other point when n <= K = 20, otherwise with the K points at k * (n // K), itself skipped (:83-101).  A match is filed
// under the ordered image pair (frame of j, frame of the target) as the row [keypoint index of j, keypoint index of the
// target]; rows keep the order the loops produce them, and an image's pairs appear in order of first use.
//
// Here the same tables come straight from the saved set that psfm_result_filter left in HBM (the CSR of trajectories of
// length >= traj_min_len; a trajectory's frames are birth .. birth + len - 1):
//   keep / compaction  flag -> exclusive scan (dynamic points dropped)
//   keypoints          stable radix sort of the kept points by frame; rank inside the frame = keypoint index
//   matches            count per point -> scan -> one thread per point emits its <= K matches at their place in the
//                      reference's loop order e; stable radix sort of (image pair key, e); pair boundaries by flag + scan
// Output tables (psfm_matches_copy): kp_off (n_img+1), kp_xy (n_kept,2); pair_key (src * n_img + tgt, ascending),
// pair_off (n_pairs+1), pair_first (loop-order position of the pair's first match -> dict order), rows (n_matches,2) i32.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "psfm_device.h"
#include "psfm_internal.h"

#define PM_BLOCK 256

// owner trajectory of flat point p: the last t with off[t] <= p
__device__ __forceinline__ int pm_owner(const int64_t* __restrict__ off, int64_t k, int64_t p)
{
    int64_t lo = 0, hi = k;            // off[lo] <= p < off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return (int)lo;
}

__global__ __launch_bounds__(PM_BLOCK) void pm_keep_kernel(const uint8_t* __restrict__ labels, int64_t n_pts, int64_t* __restrict__ keep)
{
    const int64_t p = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (p > n_pts) return;
    keep[p] = (p < n_pts && !(labels && labels[p] != 0)) ? 1 : 0;      // (entry n_pts: the scan's total)
}

// kept point q (compacted index): its frame, its flat point, its trajectory
__global__ __launch_bounds__(PM_BLOCK) void pm_compact_kernel(const uint8_t* __restrict__ labels, const int64_t* __restrict__ q_of,
                                                             int64_t n_pts, const int64_t* __restrict__ off, int64_t k,
                                                             const int* __restrict__ birth, unsigned* __restrict__ kfr,
                                                             int64_t* __restrict__ kpt, int* __restrict__ ktraj,
                                                             unsigned* __restrict__ iota, int* __restrict__ bad, int n_img)
{
    const int64_t p = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (p >= n_pts || (labels && labels[p] != 0)) return;
    const int64_t q = q_of[p];
    const int t = pm_owner(off, k, p);
    const int f = birth[t] + (int)(p - off[t]);
    // a frame outside the image list is an argument error, reported by the host once this pass is over; the entry is still
    // written (frame 0) -- the sort and the kernels behind it run before the host looks at the flag and must stay in bounds
    const bool outside = f < 0 || f >= n_img;
    if (outside) *bad = 1;
    kfr[q] = outside ? 0u : (unsigned)f;
    kpt[q] = p;
    ktraj[q] = t;
    iota[q] = (unsigned)q;
}

__global__ __launch_bounds__(PM_BLOCK) void pm_kp_off_kernel(const unsigned* __restrict__ fsorted, int64_t n_kept, int n_img,
                                                            int64_t* __restrict__ kp_off)
{
    const int i = blockIdx.x * PM_BLOCK + threadIdx.x;
    if (i > n_img) return;
    int64_t lo = 0, hi = n_kept;       // first s with fsorted[s] >= i
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (fsorted[mid] < (unsigned)i) lo = mid + 1; else hi = mid;
    }
    kp_off[i] = lo;
}

// s-th point of the frame-sorted order: keypoint index of its point, its coordinates; and how many matches the point emits
__global__ __launch_bounds__(PM_BLOCK) void pm_kp_kernel(const unsigned* __restrict__ fsorted, const unsigned* __restrict__ order,
                                                        int64_t n_kept, const int64_t* __restrict__ kp_off,
                                                        const int64_t* __restrict__ kpt, const double2* __restrict__ xy,
                                                        int* __restrict__ kp_ind, double2* __restrict__ kp_xy)
{
    const int64_t s = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (s >= n_kept) return;
    const unsigned q = order[s];
    kp_ind[q] = (int)(s - kp_off[fsorted[s]]);
    kp_xy[s] = xy[kpt[q]];
}

// kept points of trajectory t: q in [q_of[off[t]], q_of[off[t+1]])
__global__ __launch_bounds__(PM_BLOCK) void pm_count_kernel(const int* __restrict__ ktraj, const int64_t* __restrict__ q_of,
                                                           const int64_t* __restrict__ off, int64_t n_kept, int sample_k,
                                                           int64_t* __restrict__ m_cnt)
{
    const int64_t q = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (q > n_kept) return;
    if (q == n_kept) { m_cnt[q] = 0; return; }
    const int t = ktraj[q];
    const int64_t t0 = q_of[off[t]];
    const int64_t n = q_of[off[t + 1]] - t0, j = q - t0;
    int64_t m;
    if (n <= sample_k) m = n - 1;
    else {
        const int64_t stride = n / sample_k;
        m = sample_k - ((j % stride == 0 && j / stride < sample_k) ? 1 : 0);
    }
    m_cnt[q] = m;
}

__global__ __launch_bounds__(PM_BLOCK) void pm_emit_kernel(const int* __restrict__ ktraj, const int64_t* __restrict__ q_of,
                                                          const int64_t* __restrict__ off, int64_t n_kept, int sample_k,
                                                          const int64_t* __restrict__ m_off, const unsigned* __restrict__ kfr,
                                                          const int* __restrict__ kp_ind, int n_img,
                                                          unsigned long long* __restrict__ key, unsigned long long* __restrict__ val,
                                                          int2* __restrict__ rows_u)
{
    const int64_t q = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (q >= n_kept) return;
    const int t = ktraj[q];
    const int64_t t0 = q_of[off[t]];
    const int64_t n = q_of[off[t + 1]] - t0, j = q - t0;
    const int64_t reps = n <= sample_k ? n : sample_k, stride = n <= sample_k ? 1 : n / sample_k;
    int64_t e = m_off[q];
    const unsigned long long fsrc = kfr[q];
    const int ksrc = kp_ind[q];
    for (int64_t r = 0; r < reps; ++r) {
        const int64_t tl = r * stride;
        if (tl == j) continue;                               // :89-90 / :97-98
        const int64_t tq = t0 + tl;
        key[e] = fsrc * (unsigned long long)n_img + kfr[tq];
        val[e] = (unsigned long long)e;
        rows_u[e] = make_int2(ksrc, kp_ind[tq]);
        ++e;
    }
}

__global__ __launch_bounds__(PM_BLOCK) void pm_rows_kernel(const unsigned long long* __restrict__ key_s, const unsigned long long* __restrict__ val_s,
                                                          int64_t n_m, const int2* __restrict__ rows_u, int2* __restrict__ rows,
                                                          int64_t* __restrict__ flag)
{
    const int64_t s = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (s > n_m) return;
    if (s == n_m) { flag[s] = 0; return; }
    rows[s] = rows_u[val_s[s]];
    flag[s] = (s == 0 || key_s[s] != key_s[s - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(PM_BLOCK) void pm_pairs_kernel(const unsigned long long* __restrict__ key_s, const unsigned long long* __restrict__ val_s,
                                                           int64_t n_m, const int64_t* __restrict__ gid, int64_t* __restrict__ pair_key,
                                                           int64_t* __restrict__ pair_off, int64_t* __restrict__ pair_first)
{
    const int64_t s = (int64_t)blockIdx.x * PM_BLOCK + threadIdx.x;
    if (s > n_m) return;
    if (s == n_m) { pair_off[gid[s]] = n_m; return; }
    if (s == 0 || key_s[s] != key_s[s - 1]) {
        const int64_t g = gid[s];
        pair_key[g] = (int64_t)key_s[s];
        pair_off[g] = s;
        pair_first[g] = (int64_t)val_s[s];
    }
}

static psfm_status pm_scan(psfm_ctx* c, int64_t* in, int64_t* out, size_t n, hipStream_t s)
{
    size_t bytes = 0;
    psfm_status st;
    PSFM_HIP(rocprim::exclusive_scan(nullptr, bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, n, rocprim::plus<int64_t>(), s));
    if ((st = c->sort_tmp.ensure(bytes)) != PSFM_OK) return st;
    PSFM_HIP(rocprim::exclusive_scan(c->sort_tmp.p, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), s));
    return PSFM_OK;
}

static int pm_bits(unsigned long long v)   // bits needed for values < v
{
    int b = 1;
    while (b < 64 && (1ull << b) < v) ++b;
    return b;
}

static unsigned pm_grid(int64_t n) { return (unsigned)((n + PM_BLOCK - 1) / PM_BLOCK); }

extern "C" psfm_status psfm_traj_to_matches(psfm_ctx* c, int n_img, int sample_k, const uint8_t* labels, int64_t* n_kp_host,
                                            int64_t* n_matches_host, int64_t* n_pairs_host, void* stream)
{
    if (!c || !n_kp_host || !n_matches_host || !n_pairs_host || n_img < 1 || sample_k < 1) {
        psfm_set_error("psfm_traj_to_matches: bad argument (n_img=%d sample_k=%d)", n_img, sample_k);
        return PSFM_ERR_ARG;
    }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    *n_kp_host = *n_matches_host = *n_pairs_host = 0;
    c->mt_n_kp = c->mt_n_m = c->mt_n_pairs = 0; c->mt_n_img = n_img;
    psfm_status st;
    if ((st = c->mt_kp_off.ensure(8 * (size_t)(n_img + 1))) != PSFM_OK) return st;
    PSFM_HIP(hipMemsetAsync(c->mt_kp_off.p, 0, 8 * (size_t)(n_img + 1), s));
    const int64_t k = c->flt_n_traj, n_pts = c->flt_n_points;
    if (k == 0 || n_pts == 0) { PSFM_HIP(hipStreamSynchronize(s)); return PSFM_OK; }
    if (n_pts >= 0xffffffffll) { psfm_set_error("psfm_traj_to_matches: more than 2^32 points"); return PSFM_ERR_ARG; }
    const int64_t* off = c->flt_off.as<int64_t>();
    int64_t* h = (int64_t*)((char*)c->host_pinned + 272);    // [0] scalar read-back, [1] bad-frame flag
    // ---- keep + compaction ----
    if ((st = c->mt_q.ensure(8 * (size_t)(n_pts + 1))) != PSFM_OK) return st;
    if ((st = c->scan_tmp.ensure(8 * (size_t)(n_pts + 1))) != PSFM_OK) return st;
    int64_t* q_of = c->mt_q.as<int64_t>();
    hipLaunchKernelGGL(pm_keep_kernel, dim3(pm_grid(n_pts + 1)), dim3(PM_BLOCK), 0, s, labels, n_pts, c->scan_tmp.as<int64_t>());
    if ((st = pm_scan(c, c->scan_tmp.as<int64_t>(), q_of, (size_t)(n_pts + 1), s)) != PSFM_OK) return st;
    PSFM_HIP(hipMemcpyAsync(h, q_of + n_pts, 8, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    const int64_t n_kept = h[0];
    c->mt_n_kp = n_kept;
    *n_kp_host = n_kept;
    if (n_kept == 0) return PSFM_OK;
    // per kept point: frame, flat point, trajectory, keypoint index; frame-sorted copies
    const size_t a4 = ((size_t)n_kept * 4 + 255) / 256 * 256, a8 = ((size_t)n_kept * 8 + 255) / 256 * 256;
    if ((st = c->mt_pts.ensure(5 * a4 + a8 + 256)) != PSFM_OK) return st;
    char* w = (char*)c->mt_pts.p;
    unsigned* kfr = (unsigned*)w; unsigned* fsorted = (unsigned*)(w + a4); unsigned* iota = (unsigned*)(w + 2 * a4);
    unsigned* order = (unsigned*)(w + 3 * a4); int* ktraj = (int*)(w + 4 * a4); int64_t* kpt = (int64_t*)(w + 5 * a4);
    int* bad = (int*)(w + 5 * a4 + a8);
    if ((st = c->mt_kp_ind.ensure(4 * (size_t)n_kept)) != PSFM_OK) return st;
    if ((st = c->mt_kp_xy.ensure(16 * (size_t)n_kept)) != PSFM_OK) return st;
    PSFM_HIP(hipMemsetAsync(bad, 0, 4, s));
    hipLaunchKernelGGL(pm_compact_kernel, dim3(pm_grid(n_pts)), dim3(PM_BLOCK), 0, s, labels, (const int64_t*)q_of, n_pts, off, k,
                       c->flt_birth.as<int>(), kfr, kpt, ktraj, iota, bad, n_img);
    {
        size_t bytes = 0;
        const int bits = pm_bits((unsigned long long)n_img);
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kfr, fsorted, iota, order, (size_t)n_kept, 0, bits, s));
        if ((st = c->sort_tmp.ensure(bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, bytes, kfr, fsorted, iota, order, (size_t)n_kept, 0, bits, s));
    }
    hipLaunchKernelGGL(pm_kp_off_kernel, dim3(pm_grid(n_img + 1)), dim3(PM_BLOCK), 0, s, (const unsigned*)fsorted, n_kept, n_img,
                       c->mt_kp_off.as<int64_t>());
    hipLaunchKernelGGL(pm_kp_kernel, dim3(pm_grid(n_kept)), dim3(PM_BLOCK), 0, s, (const unsigned*)fsorted, (const unsigned*)order, n_kept,
                       (const int64_t*)c->mt_kp_off.as<int64_t>(), (const int64_t*)kpt, c->flt_xy.as<double2>(), c->mt_kp_ind.as<int>(),
                       c->mt_kp_xy.as<double2>());
    // ---- matches: count, scan, emit in loop order ----
    if ((st = c->mt_moff.ensure(8 * (size_t)(n_kept + 1))) != PSFM_OK) return st;
    if ((st = c->scan_tmp.ensure(8 * (size_t)(n_kept + 1))) != PSFM_OK) return st;
    hipLaunchKernelGGL(pm_count_kernel, dim3(pm_grid(n_kept + 1)), dim3(PM_BLOCK), 0, s, (const int*)ktraj, (const int64_t*)q_of, off, n_kept,
                       sample_k, c->scan_tmp.as<int64_t>());
    if ((st = pm_scan(c, c->scan_tmp.as<int64_t>(), c->mt_moff.as<int64_t>(), (size_t)(n_kept + 1), s)) != PSFM_OK) return st;
    PSFM_HIP(hipMemcpyAsync(h, c->mt_moff.as<int64_t>() + n_kept, 8, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipMemcpyAsync(h + 1, bad, 4, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    if ((int)(h[1] & 0xffffffff) != 0) { psfm_set_error("psfm_traj_to_matches: a trajectory has a frame outside [0, n_img=%d)", n_img); return PSFM_ERR_ARG; }
    const int64_t n_m = h[0];
    c->mt_n_m = n_m;
    *n_matches_host = n_m;
    if (n_m == 0) return PSFM_OK;
    const size_t m8 = ((size_t)n_m * 8 + 255) / 256 * 256;
    if ((st = c->mt_keys.ensure(5 * m8)) != PSFM_OK) return st;
    if ((st = c->mt_rows.ensure(8 * (size_t)n_m)) != PSFM_OK) return st;
    char* mk = (char*)c->mt_keys.p;
    unsigned long long* key = (unsigned long long*)mk; unsigned long long* key_s = (unsigned long long*)(mk + m8);
    unsigned long long* val = (unsigned long long*)(mk + 2 * m8); unsigned long long* val_s = (unsigned long long*)(mk + 3 * m8);
    int2* rows_u = (int2*)(mk + 4 * m8);
    hipLaunchKernelGGL(pm_emit_kernel, dim3(pm_grid(n_kept)), dim3(PM_BLOCK), 0, s, (const int*)ktraj, (const int64_t*)q_of, off, n_kept,
                       sample_k, (const int64_t*)c->mt_moff.as<int64_t>(), (const unsigned*)kfr, (const int*)c->mt_kp_ind.as<int>(), n_img, key,
                       val, rows_u);
    {
        size_t bytes = 0;
        const int bits = pm_bits((unsigned long long)n_img * (unsigned long long)n_img);
        PSFM_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key, key_s, val, val_s, (size_t)n_m, 0, bits, s));
        if ((st = c->sort_tmp.ensure(bytes)) != PSFM_OK) return st;
        PSFM_HIP(rocprim::radix_sort_pairs(c->sort_tmp.p, bytes, key, key_s, val, val_s, (size_t)n_m, 0, bits, s));
    }
    // ---- rows in (pair, loop) order; pair boundaries ----
    if ((st = c->scan_tmp.ensure(8 * (size_t)(n_m + 1))) != PSFM_OK) return st;
    if ((st = c->mt_gid.ensure(8 * (size_t)(n_m + 1))) != PSFM_OK) return st;
    hipLaunchKernelGGL(pm_rows_kernel, dim3(pm_grid(n_m + 1)), dim3(PM_BLOCK), 0, s, (const unsigned long long*)key_s,
                       (const unsigned long long*)val_s, n_m, (const int2*)rows_u, c->mt_rows.as<int2>(), c->scan_tmp.as<int64_t>());
    if ((st = pm_scan(c, c->scan_tmp.as<int64_t>(), c->mt_gid.as<int64_t>(), (size_t)(n_m + 1), s)) != PSFM_OK) return st;
    PSFM_HIP(hipMemcpyAsync(h, c->mt_gid.as<int64_t>() + n_m, 8, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    const int64_t n_pairs = h[0];
    c->mt_n_pairs = n_pairs;
    *n_pairs_host = n_pairs;
    if ((st = c->mt_pairs.ensure(8 * (size_t)(3 * n_pairs + 1))) != PSFM_OK) return st;
    int64_t* pk = c->mt_pairs.as<int64_t>();
    hipLaunchKernelGGL(pm_pairs_kernel, dim3(pm_grid(n_m + 1)), dim3(PM_BLOCK), 0, s, (const unsigned long long*)key_s,
                       (const unsigned long long*)val_s, n_m, (const int64_t*)c->mt_gid.as<int64_t>(), pk, pk + n_pairs, pk + 2 * n_pairs + 1);
    PSFM_HIP(hipGetLastError());
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}

extern "C" psfm_status psfm_matches_copy(psfm_ctx* c, int64_t* kp_off_host, double* kp_xy_host, int64_t* pair_key_host,
                                         int64_t* pair_off_host, int64_t* pair_first_host, int32_t* rows_host, void* stream)
{
    if (!c) { psfm_set_error("ctx is NULL"); return PSFM_ERR_ARG; }
    PSFM_HIP(hipSetDevice(c->device));
    PsfmGate gate(c->device, 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_kp = c->mt_n_kp, n_m = c->mt_n_m, n_p = c->mt_n_pairs;
    if (kp_off_host && c->mt_kp_off.p) PSFM_HIP(hipMemcpyAsync(kp_off_host, c->mt_kp_off.p, 8 * (size_t)(c->mt_n_img + 1), hipMemcpyDeviceToHost, s));
    if (kp_xy_host && n_kp > 0) PSFM_HIP(hipMemcpyAsync(kp_xy_host, c->mt_kp_xy.p, 16 * (size_t)n_kp, hipMemcpyDeviceToHost, s));
    if (n_p > 0) {
        const int64_t* pk = c->mt_pairs.as<int64_t>();
        if (pair_key_host) PSFM_HIP(hipMemcpyAsync(pair_key_host, pk, 8 * (size_t)n_p, hipMemcpyDeviceToHost, s));
        if (pair_off_host) PSFM_HIP(hipMemcpyAsync(pair_off_host, pk + n_p, 8 * (size_t)(n_p + 1), hipMemcpyDeviceToHost, s));
        if (pair_first_host) PSFM_HIP(hipMemcpyAsync(pair_first_host, pk + 2 * n_p + 1, 8 * (size_t)n_p, hipMemcpyDeviceToHost, s));
    } else if (pair_off_host) {
        pair_off_host[0] = 0;
    }
    if (rows_host && n_m > 0) PSFM_HIP(hipMemcpyAsync(rows_host, c->mt_rows.p, 8 * (size_t)n_m, hipMemcpyDeviceToHost, s));
    PSFM_HIP(hipStreamSynchronize(s));
    return PSFM_OK;
}
