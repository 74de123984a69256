// Host build of the RESIDENT solve's bookkeeping (particle-sfm_amd/csrc/psfm_pc_resident.h + psfm_pc_core.h + psfm_pc_control.h)
// for tests/test_pc_chain_host.py: one whole path-consistency solve of a batch of tracks the way psfm_pc_resident_kernel<NS> runs
// it -- n_blocks blocks of PC_BLOCK threads, every block with its list of lanes (pc_list_plan: the chunks it compacts, banded or
// not; `part` says which rows take part), every thread with NS slots (slot k = entry k * PC_BLOCK + t) and a streamed tail behind
// them (entries from NS * PC_BLOCK on: state in memory, the launch chain's per-entry functions, candidates ping-ponging between the
// iterate buffers), a round = candidate -> evaluate ahead -> sums | refresh, the blocks' sums added in the tree order of
// pc_tree_totals, every block running the same control step; accepted: slots take the candidate (recomputed) and (u', d');
// give-up: a block that leaves in a round takes the whole solve with it and NOTHING has been written.  The loop bodies restate the
// kernel's; every function they call is the device's own code.  What this cannot see: the order in which a block adds its threads'
// terms (DPP row tree on the device, thread order here) and the granule hand-off itself.
// Test infrastructure (compiled by the test with g++ -O2 -mfma -ffp-contract=off).
#include <vector>

#include "psfm_pc_resident.h"

// returns 0: done (x_out, stats, costs filled); 1: ran out of rounds; 2: gave up (a block quit: nothing written, x_out untouched)
// world > 1 (owner[i] = the rank row i belongs to): the same solve spread over `world` ranks the way psfm_shard_solve_peer runs it -- every
// rank its own launch of n_blocks blocks over the rows it owns, every rank's tree total, the totals added in rank order (pc_peer_totals),
// every block of every rank running the same control step.  quit_block counts through the ranks' blocks (rank * n_blocks + block).
static int pc_host_resident_solve_ranks(long n, const unsigned char* part, const unsigned char* owner, int world, const double* x0,
                                        const double* ref1, const double* ref2, const double* scale, const float* flow, int H, int W,
                                        int n_blocks_per_rank, int NS, int banded, int init_inside, int quit_block, int quit_round,
                                        double* x_out, int* stats, double* costs, int* info);

extern "C" int pc_host_resident_solve(long n, const unsigned char* part, const double* x0, const double* ref1, const double* ref2,
                                      const double* scale, const float* flow, int H, int W, int n_blocks, int NS, int banded,
                                      int init_inside, int quit_block, int quit_round, double* x_out, int* stats, double* costs,
                                      int* info /* [0] longest list, [1] streamed entries, [2] rounds, [3] empty blocks */)
{
    return pc_host_resident_solve_ranks(n, part, nullptr, 1, x0, ref1, ref2, scale, flow, H, W, n_blocks, NS, banded, init_inside, quit_block,
                                        quit_round, x_out, stats, costs, info);
}

extern "C" int pc_host_peer_solve(long n, const unsigned char* part, const unsigned char* owner, int world, const double* x0, const double* ref1,
                                  const double* ref2, const double* scale, const float* flow, int H, int W, int n_blocks_per_rank, int NS,
                                  int banded, int init_inside, int quit_block, int quit_round, double* x_out, int* stats, double* costs, int* info)
{
    return pc_host_resident_solve_ranks(n, part, owner, world, x0, ref1, ref2, scale, flow, H, W, n_blocks_per_rank, NS, banded, init_inside,
                                        quit_block, quit_round, x_out, stats, costs, info);
}

static int pc_host_resident_solve_ranks(long n, const unsigned char* part, const unsigned char* owner, int world, const double* x0,
                                        const double* ref1, const double* ref2, const double* scale, const float* flow, int H, int W,
                                        int n_blocks_per_rank, int NS, int banded, int init_inside, int quit_block, int quit_round,
                                        double* x_out, int* stats, double* costs, int* info)
{
    const int n_blocks = world * n_blocks_per_rank;       // all ranks' blocks, rank-major: block lb = rank * n_blocks_per_rank + b
    const PcF2* F = (const PcF2*)flow;
    if (NS < 1) NS = 1;
    if (NS > 3) NS = 3;
    // ---- the blocks' lists (pc_build_list: chunks of the plan, the participating lanes of a chunk in lane order) ----
    std::vector<std::vector<long>> lists((size_t)n_blocks);
    for (int b = 0; b < n_blocks; ++b) {
        const int rank = b / n_blocks_per_rank;
        const PcListPlan plan = pc_list_plan(b % n_blocks_per_rank, n_blocks_per_rank, (int)n, banded);
        for (int q = plan.first; pc_list_chunk_ok(plan, q); q += plan.step)
            for (int t = 0; t < PC_BLOCK; ++t) {
                const long i = (long)(plan.band0 + q) * PC_BLOCK + t;
                if (i < n && (!part || part[i]) && (!owner || owner[i] == rank)) lists[(size_t)b].push_back(i);
            }
    }
    {   // (every participating row is in exactly one list)
        std::vector<int> seen((size_t)n, 0);
        for (auto& l : lists) for (long i : l) seen[(size_t)i] += 1;
        for (long i = 0; i < n; ++i) if (seen[(size_t)i] != ((!part || part[i]) ? 1 : 0)) return 9;
    }
    info[0] = info[1] = info[2] = info[3] = 0;
    for (auto& l : lists) {
        if ((int)l.size() > info[0]) info[0] = (int)l.size();
        if ((long)l.size() > (long)NS * PC_BLOCK) info[1] += (int)(l.size() - (size_t)NS * PC_BLOCK);
        if (l.empty()) info[3] += 1;
    }
    // ---- state: slots per (block, thread, k); for the streamed entries the chain's memory (constants, iterate buffers 1 / 2) ----
    struct Th { PcSlot T[3]; double next[3][8]; };
    std::vector<std::vector<Th>> blk((size_t)n_blocks, std::vector<Th>(PC_BLOCK));
    std::vector<double> b1(4 * (size_t)n), b2(4 * (size_t)n), js(2 * (size_t)n);
    auto buf = [&](int m) -> double* { return m == 0 ? const_cast<double*>(x0) : (m == 1 ? b1.data() : b2.data()); };
    std::vector<double> rows((size_t)n_blocks * PC_NSUM);
    double tot[PC_NSUM];
    PsfmSolveCtrl C;
    // ---- iteration 0 (pc_init_entry in batch mode: references and scale are given) ----
    const double mu0 = 1e-8;
    for (int b = 0; b < n_blocks; ++b) {
        double* acc = &rows[(size_t)b * PC_NSUM];
        for (int k = 0; k < PC_NSUM; ++k) acc[k] = 0.0;
        const std::vector<long>& lst = lists[(size_t)b];
        const long cnt = (long)lst.size();
        for (int t = 0; t < PC_BLOCK; ++t) {
            auto init_entry = [&](long i, PcSlot* slot) {
                const double s = scale[i];
                double r0[6], j0[4];
                pc_core_eval(F, H, W, x0 + 4 * i, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r0, j0);
                const PcConst c = pc_core_const(s, j0);
                if (!slot || !init_inside) { js[2 * i] = c.S0q; js[2 * i + 1] = c.S1q; }      // (constants to memory: streamed entries; a separate pc_init launch)
                double a1[PC_NSUM];
                for (int k = 0; k < PC_NSUM; ++k) a1[k] = 0.0;
                a1[SUM_CNT] += 1.0;
                a1[SUM_COST0] += pc_core_cost(r0);
                PcSys y;
                pc_core_system<true>(x0 + 4 * i, r0, j0, c, mu0, pc_core_iA22(c, mu0), a1, y, CH_QUD, CH_QDD);
                for (int k = 0; k < PC_NSUM; ++k) acc[k] = k == SUM_GMAX ? fmax(acc[k], a1[k]) : acc[k] + a1[k];
                if (slot && init_inside) pc_slot_fill(*slot, s, c, x0 + 4 * i, y);
            };
            for (int k = 0; k < NS; ++k)
                if (pc_slot_entry(k, t) < cnt) init_entry(lst[(size_t)pc_slot_entry(k, t)], &blk[(size_t)b][(size_t)t].T[k]);
            for (long p = pc_stream_first(NS, t); p < cnt; p += PC_BLOCK) init_entry(lst[(size_t)p], nullptr);
        }
    }
    { std::vector<int> nb((size_t)world, n_blocks_per_rank); pc_peer_totals(rows.data(), PC_NSUM, nb.data(), world, PC_NSUM, tot); }
    pc_chain_control(C, tot, 0);
    C.launches = 1;
    if (!init_inside && !C.done) {
        // behind a separate iteration-0 launch: the slots load constants and start values from memory and solve the system at x0
        for (int b = 0; b < n_blocks; ++b)
            for (int t = 0; t < PC_BLOCK; ++t)
                for (int k = 0; k < NS; ++k) {
                    if (pc_slot_entry(k, t) >= (long)lists[(size_t)b].size()) continue;
                    const long i = lists[(size_t)b][(size_t)pc_slot_entry(k, t)];
                    PcSlot& T = blk[(size_t)b][(size_t)t].T[k];
                    T.s = scale[i]; T.S0q = js[2 * i]; T.S1q = js[2 * i + 1];
                    for (int q = 0; q < 4; ++q) T.x[q] = x0[4 * i + q];
                    const PcTaps tp = pc_core_taps<true>(F, H, W, T.x);
                    pc_slot_start(T, tp, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], C.mu);
                }
    }
    // ---- the rounds ----
    int rounds = 0;
    while (!C.done && rounds < 2 * 200 + 64) {
        const double mu = C.mu, a = C.dl_a, b_ = C.dl_b;
        const int cur = C.cur;
        const bool refresh = C.kind_next != 0;
        const double mu_next = fmax(1e-8, 2.0 * mu / 10.0);
        const double* xc = buf(cur);
        double* xn = buf(pc_other(cur));
        for (int b = 0; b < n_blocks; ++b) {
            if (b == quit_block && rounds == quit_round) return 2;      // poison: every block leaves, nothing has been written
            double* acc = &rows[(size_t)b * PC_NSUM];
            for (int k = 0; k < PC_NSUM; ++k) acc[k] = 0.0;
            const std::vector<long>& lst = lists[(size_t)b];
            const long cnt = (long)lst.size();
            for (int t = 0; t < PC_BLOCK; ++t) {
                double a1[PC_NSUM];
                for (int k = 0; k < PC_NSUM; ++k) a1[k] = 0.0;
                Th& th = blk[(size_t)b][(size_t)t];
                for (int k = 0; k < NS; ++k) {
                    if (pc_slot_entry(k, t) >= cnt) continue;
                    const long i = lst[(size_t)pc_slot_entry(k, t)];
                    PcSlot& T = th.T[k];
                    if (refresh) {
                        pc_slot_refresh<true>(T, F, H, W, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], mu, a1);
                    } else {
                        double xe[4];
                        (void)pc_slot_candidate(T, a, b_, xe);
                        const PcTaps tp = pc_core_taps<true>(F, H, W, xe);
                        pc_slot_round(T, tp, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], a, b_, mu_next, a1, th.next[k]);
                    }
                }
                // the streamed tail: pc_refresh_entry / pc_ahead_entry of the launch chain on the entries beyond the slots
                for (long p = pc_stream_first(NS, t); p < cnt; p += PC_BLOCK) {
                    const long i = lst[(size_t)p];
                    const double s = scale[i];
                    PcConst c;
                    c.s = s; c.S0q = js[2 * i]; c.S1q = js[2 * i + 1]; c.H22 = fma(s, s, 1.0);
                    const double* x = xc + 4 * i;
                    double r[6], jac[4];
                    PcSys y;
                    pc_core_eval<true>(F, H, W, x, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, jac);
                    if (refresh) { pc_core_system<true>(x, r, jac, c, mu, pc_core_iA22(c, mu), a1, y, CH_QUD, CH_QDD); continue; }
                    double unused[PC_NSUM], xp[4];
                    for (int k = 0; k < PC_NSUM; ++k) unused[k] = 0.0;
                    pc_core_system<false>(x, r, jac, c, mu, pc_core_iA22(c, mu), unused, y, 0, 0);
                    pc_core_step<false, false>(x, r, jac, c, y, a, b_, a1, xp);
                    for (int k = 0; k < 4; ++k) xn[4 * i + k] = xp[k];
                    pc_core_eval<true>(F, H, W, xp, ref1[2 * i], ref1[2 * i + 1], ref2[2 * i], ref2[2 * i + 1], s, r, jac);
                    a1[SUM_COST] += pc_core_cost(r);
                    pc_core_system<true>(xp, r, jac, c, mu_next, pc_core_iA22(c, mu_next), a1, y, CH_QUD, CH_QDD);
                }
                for (int k = 0; k < PC_NSUM; ++k) acc[k] = k == SUM_GMAX ? fmax(acc[k], a1[k]) : acc[k] + a1[k];
            }
        }
        ++rounds;
        { std::vector<int> nb((size_t)world, n_blocks_per_rank); pc_peer_totals(rows.data(), PC_NSUM, nb.data(), world, PC_NSUM, tot); }
        const int cur0 = C.cur;
        pc_chain_control(C, tot, 1);
        C.launches += 1;
        if (C.cur != cur0) {         // accepted: x <- the candidate (recomputed), (u, d) <- what was solved there
            for (int b = 0; b < n_blocks; ++b)
                for (int t = 0; t < PC_BLOCK; ++t)
                    for (int k = 0; k < NS; ++k)
                        if (pc_slot_entry(k, t) < (long)lists[(size_t)b].size()) pc_slot_accept(blk[(size_t)b][(size_t)t].T[k], a, b_, blk[(size_t)b][(size_t)t].next[k]);
        }
    }
    info[2] = rounds;
    if (!C.done) return 1;
    // ---- write-back: slots write their own x when the solve moved, streamed entries come from the buffer the control block names ----
    const bool moved = pc_res_moved(C);
    const double* xs = buf(pc_res_stream_source(C));
    for (int b = 0; b < n_blocks; ++b) {
        const std::vector<long>& lst = lists[(size_t)b];
        const long cnt = (long)lst.size();
        for (int t = 0; t < PC_BLOCK; ++t) {
            for (int k = 0; k < NS; ++k) {
                if (pc_slot_entry(k, t) >= cnt) continue;
                const long i = lst[(size_t)pc_slot_entry(k, t)];
                for (int q = 0; q < 4; ++q) x_out[4 * i + q] = moved ? blk[(size_t)b][(size_t)t].T[k].x[q] : x0[4 * i + q];
            }
            for (long p = pc_stream_first(NS, t); p < cnt; p += PC_BLOCK) {
                const long i = lst[(size_t)p];
                for (int q = 0; q < 4; ++q) x_out[4 * i + q] = xs[4 * i + q];
            }
        }
    }
    stats[0] = C.iteration; stats[1] = C.successful; stats[2] = C.n_tracks == 0 ? -1 : C.termination; stats[3] = C.nonGN;
    stats[4] = C.launches; stats[5] = C.done; stats[6] = C.failed;
    if (C.failed) stats[2] = PSFM_TERM_FAILURE;
    costs[0] = C.initial_cost; costs[1] = C.x_cost;
    return 0;
}
