// Host build of the DEVICE's arithmetic (particle-sfm_amd/csrc/psfm_device.h, psfm_chain.h through tests/host/shim) for
// tests/test_device_arith_host.py: the fp32 sampler, the flow_check verdict of a pixel (error-map form, mask-only form, interior
// form), and a whole track() run whose per-track arithmetic -- the step (gathers + blend + bounds + occlusion test), the folded
// EDT respawn rule (psfm_block_grid), the id key -- is the device's own code, with plain lists around it where the kernels
// have lanes, ballots and atomics.  Test infrastructure.
#include <algorithm>
#include <vector>

#include "psfm_chain.h"
#include "psfm_pc_control.h"

// tests/host/pc_chain_host.cpp (linked into the same library): the device's launch chain for a batch of tracks
extern "C" int pc_host_chain_solve(long n, const double* x0, const double* ref1, const double* ref2, const double* scale,
                                   const float* flow, int H, int W, int pair, double* x_out, int* stats, double* costs);

extern "C" void psfm_host_grid_sample(const float* map, int H, int W, const float* xy, long n, float* out)
{
    const float cw = (float)((W - 1) / 2.0), ch = (float)((H - 1) / 2.0);
    for (long i = 0; i < n; ++i) {
        const PsfmTaps t = psfm_taps(xy[2 * i], xy[2 * i + 1], cw, ch, H, W);
        const float2 v = psfm_sample_flow((const float2*)map, H, W, t);
        out[2 * i] = v.x; out[2 * i + 1] = v.y;
    }
}

extern "C" void psfm_host_grid_sample1(const float* map, int H, int W, const float* xy, long n, float* out)
{
    const float cw = (float)((W - 1) / 2.0), ch = (float)((H - 1) / 2.0);
    for (long i = 0; i < n; ++i) out[i] = psfm_sample_f32(map, H, W, psfm_taps(xy[2 * i], xy[2 * i + 1], cw, ch, H, W));
}

// form 0: error map + mask (true division, square root); 1: mask only (fast exact division, threshold under the root);
// 2: mask only, the wave-uniform interior form wherever the four taps lie inside the map (the general form elsewhere)
extern "C" void psfm_host_flow_check(const float* F, const float* B, int H, int W, float thres, int form, uint8_t* occ, float* err)
{
    PsfmFcParams q;
    q.H = H; q.W = W; q.cw = (float)((W - 1) / 2.0); q.ch = (float)((H - 1) / 2.0);
    q.rcw = psfm_rcp_host(q.cw); q.rch = psfm_rcp_host(q.ch); q.thres = thres; q.t2 = psfm_sq_threshold(thres);
    const float2* f2 = (const float2*)F;
    const float2* b2 = (const float2*)B;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float2 f = f2[(long)y * W + x];
            float e = 0.0f;
            uint8_t o;
            if (form == 0) o = psfm_flow_check_px<true>(b2, x, y, f, q, &e);
            else if (form == 1) o = psfm_flow_check_px<false>(b2, x, y, f, q, &e);
            else {
                const float X = __fadd_rn((float)x, f.x), Y = __fadd_rn((float)y, f.y);
                const PsfmTaps t = psfm_taps_t<true>(X, Y, q.cw, q.ch, q.rcw, q.rch, H, W);
                const bool interior = (t.x0 >= 0) & (t.x0 + 1 < W) & (t.y0 >= 0) & (t.y0 + 1 < H);
                o = interior ? psfm_flow_check_px_interior(b2, X, Y, f, t, q) : psfm_flow_check_px<false>(b2, x, y, f, q, &e);
            }
            occ[(long)y * W + x] = o;
            if (form == 0) err[(long)y * W + x] = e;
        }
}

struct HostFrame {
    const float2* flow; const uint8_t* occ;
    int H, W; float cw, ch, rcw, rch;
    int ratio, GW, GH;
    uint8_t* blocked_cur; uint8_t stamp_cur;
    PsfmFastDiv rdiv;
};
struct HostTrack { int birth, gidx; std::vector<double2> pts; int last; };

// track.py:24-50 with the device's arithmetic.  Returns the number of trajectories (ids = order of the key (last valid
// time, birth frame, birth grid index), psfm_key); points into xy (cap_points x 2).
// flows2 / occs2 != NULL: track_optimize.py:24-53 -- behind the step of frame t >= 1 the tracks with three buffered points
// (times t-1, t, t+1) are optimised together (trajectory.py:161-194): references and weight from the fp32 sampler at p0
// (restating pc_init_tracks of psfm_solver.hip with the device's sampler), the solve by the device's launch chain.
// iters_out (n_flows - 1 entries, may be NULL): iterations of every solve.
extern "C" long psfm_host_track(const float* const* flows, const uint8_t* const* occs, int n_flows, int H, int W, int ratio,
                                int* birth_out, int* len_out, double* xy_out, long cap_tracks, long cap_points, long* n_points_out,
                                const float* const* flows2, const uint8_t* const* occs2, int* iters_out, int* terms_out)
{
    HostFrame a;
    a.H = H; a.W = W; a.cw = (float)((W - 1) / 2.0); a.ch = (float)((H - 1) / 2.0);
    a.rcw = psfm_rcp_host(a.cw); a.rch = psfm_rcp_host(a.ch);
    a.ratio = ratio; a.GW = (W + ratio - 1) / ratio; a.GH = (H + ratio - 1) / ratio; a.rdiv = psfm_fastdiv_make((unsigned)ratio);
    const int G = a.GW * a.GH;
    std::vector<uint8_t> marks_prev(G, 0), marks_cur(G, 0);
    std::vector<HostTrack> active, done;
    bool any_prev = true;
    for (int t = 0; t < n_flows; ++t) {
        // ---- births (trajectory.py:99-115 at t = 0; :129-152 afterwards: grid points farther than `ratio` from every track) ----
        for (int g = 0; g < G; ++g) {
            bool birth = t == 0 ? true : marks_prev[g] == 0;
            // no survivor at all: SciPy's EDT measures to a phantom feature at (y = -1, x = 0) -- grid point 0 stays empty
            if (t > 0 && !any_prev) birth = g == 0 ? ((0 + 1) * (0 + 1) + 0 * 0) > ratio * ratio : true;
            if (!birth) continue;
            HostTrack k;
            k.birth = t; k.gidx = g; k.last = -1;
            k.pts.push_back(make_double2((double)((g % a.GW) * ratio), (double)((g / a.GW) * ratio)));
            active.push_back(k);        // (behind the older tracks, in grid order: the active list stays sorted by (birth, grid index))
        }
        // ---- step_forward + extend_all for every active track (trajectory.py:45-62, :129-147) ----
        std::fill(marks_cur.begin(), marks_cur.end(), 0);
        a.flow = (const float2*)flows[t]; a.occ = occs[t]; a.blocked_cur = marks_cur.data(); a.stamp_cur = 1;
        bool any = false;
        std::vector<HostTrack> next;
        next.reserve(active.size());
        for (auto& k : active) {
            const double2 p = k.pts.back();
            const PsfmStepLoads L = psfm_step_issue(a, p);
            const PsfmStep s = psfm_step_finish(a, p, L);
            if (s.alive) {
                k.pts.push_back(s.next);
                psfm_block_grid<0, false>(a, (int)s.next.x, (int)s.next.y);
                any = true;
                next.push_back(std::move(k));
            } else {
                k.last = t;
                done.push_back(std::move(k));
            }
        }
        active.swap(next);
        marks_prev.swap(marks_cur);
        any_prev = any;
        if (flows2 && t + 1 >= 2) {      // track_optimize.py:49-50
            const float2* flow01 = (const float2*)flows[t - 1];
            const float2* flow02 = (const float2*)flows2[t - 1];
            const uint8_t* occ02 = occs2[t - 1];
            std::vector<HostTrack*> sel;
            for (auto& k : active) if (k.birth <= t - 1) sel.push_back(&k);
            const long n = (long)sel.size();
            std::vector<double> x0(4 * n), r1(2 * n), r2(2 * n), sc(n), xo(4 * n);
            for (long i = 0; i < n; ++i) {
                const std::vector<double2>& q = sel[i]->pts;
                const double2 p0 = q[q.size() - 3], p1 = q[q.size() - 2], p2 = q[q.size() - 1];
                const PsfmTaps tp = psfm_taps((float)p0.x, (float)p0.y, a.cw, a.ch, H, W);
                const PsfmTapIdx ki = psfm_tap_idx(H, W, tp);
                const float2 f01 = psfm_sample_flow(flow01, ki, tp);
                const float2 f02 = psfm_sample_flow(flow02, ki, tp);
                const float o02 = psfm_sample_mask(occ02, ki, tp);
                const float nrm = sqrtf(__fadd_rn(__fmul_rn(f02.x, f02.x), __fmul_rn(f02.y, f02.y)));
                const float sf = __fmul_rn(__fsub_rn(1.0f, o02), nrm < 20.0f ? 1.0f : 0.0f);      // trajectory.py:179
                sc[i] = (double)sf;
                r1[2 * i] = p0.x + (double)f01.x; r1[2 * i + 1] = p0.y + (double)f01.y;
                r2[2 * i] = p0.x + (double)f02.x; r2[2 * i + 1] = p0.y + (double)f02.y;
                x0[4 * i] = p1.x; x0[4 * i + 1] = p1.y; x0[4 * i + 2] = p2.x; x0[4 * i + 3] = p2.y;
            }
            int st[7] = {0, 0, -1, 0, 0, 0, 0};
            double costs[2];
            if (n > 0) pc_host_chain_solve(n, x0.data(), r1.data(), r2.data(), sc.data(), flows[t], H, W, 1, xo.data(), st, costs);
            if (iters_out) iters_out[t - 1] = n > 0 ? st[0] : -1;
            if (terms_out) terms_out[t - 1] = n > 0 ? st[2] : -1;
            for (long i = 0; i < n; ++i) {
                std::vector<double2>& q = sel[i]->pts;
                q[q.size() - 2] = make_double2(xo[4 * i], xo[4 * i + 1]);
                q[q.size() - 1] = make_double2(xo[4 * i + 2], xo[4 * i + 3]);
            }
        }
    }
    for (auto& k : active) { k.last = n_flows; done.push_back(std::move(k)); }   // clear_active (trajectory.py:154-158)
    // ---- ids: rank under the device's key ----
    int shift_b = 0; while ((1 << shift_b) < G) ++shift_b;
    int bits_t = 0; while ((1 << bits_t) < n_flows + 2) ++bits_t;
    const int shift_d = shift_b + bits_t;
    std::vector<std::pair<unsigned long long, long>> order;
    for (long i = 0; i < (long)done.size(); ++i) order.push_back({psfm_key(done[i].last, done[i].birth, done[i].gidx, shift_b, shift_d), i});
    std::sort(order.begin(), order.end());
    long np = 0;
    if ((long)done.size() > cap_tracks) return -1;
    for (long r = 0; r < (long)order.size(); ++r) {
        const HostTrack& k = done[order[r].second];
        const int len = k.last - k.birth + 1;
        if ((long)k.pts.size() < len || np + len > cap_points) return -2;
        birth_out[r] = k.birth; len_out[r] = len;
        for (int j = 0; j < len; ++j) { xy_out[2 * (np + j)] = k.pts[j].x; xy_out[2 * (np + j) + 1] = k.pts[j].y; }
        np += len;
    }
    *n_points_out = np;
    return (long)done.size();
}
