// psfm_pc_resident.h -- the bookkeeping of the RESIDENT solve (psfm_pc_resident_kernel in psfm_solver.hip: the launch chain's
// trust-region loop as ONE launch with the tracks' solver state on chip) that does not depend on lanes, LDS or granules: which lane
// chunks a block's list is made of, which list entry a thread's slot holds, what a slot keeps between rounds and what the four
// things that can happen to it in a round do to it -- filled at iteration 0, evaluated ahead at a candidate, accepted, refreshed for
// a raised mu -- where the totals of the blocks' sums are added in which order, and what the write-back takes from where.
// Host-compilable like psfm_pc_core.h / psfm_pc_control.h: tests/host/pc_resident_host.cpp drives exactly these functions with
// NS = 1..3 slots per thread, lists longer than the slots (the streamed tail) and blocks that give up, against the launch chain and
// the oracle (tests/test_pc_chain_host.py) -- without a GPU.  What stays device-only in psfm_solver.hip: the ballots that compact a
// chunk, the block sums through DPP / LDS, the two-hop all-reduce of tagged granules.
#pragma once
#include "psfm_pc_core.h"
#include "psfm_pc_control.h"

#ifndef PC_BLOCK
#define PC_BLOCK 256
#endif
#define PC_LEADERS 32           // first-level sums of the block tree (the resident solve's leaders)

// ---- the list of a block: the lane chunks (PC_BLOCK lanes each) it compacts, in this order (pc_build_list) ----
// By default XCD-BANDED: blocks go to the eight XCDs round-robin, each XCD has a private L2, lanes are (roughly) in image order --
// the blocks of XCD x = b % 8 share the x-th eighth of the chunks (block b takes chunks x * per + b / 8 + k * (blocks per XCD)), so
// the taps of an XCD's tracks come from one band of the flow field.  Not banded (or a grid that is not a multiple of 8, or fewer
// than 64 chunks): chunks b, b + n_blocks, ...
struct PcListPlan { int per, first, step, band0, nchunk; };
PC_HD PcListPlan pc_list_plan(int b, int n_blocks, int n, int want_banded)
{
    PcListPlan p;
    p.nchunk = (n + PC_BLOCK - 1) / PC_BLOCK;
    const bool banded = want_banded && (n_blocks & 7) == 0 && p.nchunk >= 64;
    p.per = banded ? (p.nchunk + 7) / 8 : p.nchunk;           // chunks of a band
    p.first = banded ? (b >> 3) : b;                          // first chunk of this block inside its band
    p.step = banded ? (n_blocks >> 3) : n_blocks;
    p.band0 = banded ? (b & 7) * p.per : 0;
    return p;
}
// the q-th candidate chunk of the plan is chunk band0 + q for q = first, first + step, ... while q < per and band0 + q < nchunk
PC_HD bool pc_list_chunk_ok(const PcListPlan& p, int q) { return q < p.per && p.band0 + q < p.nchunk; }

// ---- slots: slot k of thread t holds entry k * PC_BLOCK + t of the block's list (if the list is that long); entries from
//      NS * PC_BLOCK on are STREAMED: thread t walks NS * PC_BLOCK + t, + PC_BLOCK, ... through memory like the launch chain ----
PC_HD int pc_slot_entry(int k, int t) { return k * PC_BLOCK + t; }
PC_HD int pc_stream_first(int ns, int t) { return ns * PC_BLOCK + t; }

struct PcSlot {
    double s, S0q, S1q;            // weight, squared Jacobi scaling of columns 0, 1
    double x[4], u[4], d[4];       // the iterate, and the system's solution there for the mu in force
};

PC_HD PcConst pc_slot_const(const PcSlot& T)
{
    PcConst c;
    c.s = T.s; c.S0q = T.S0q; c.S1q = T.S1q; c.H22 = fma(T.s, T.s, 1.0);
    return c;
}

// the candidate x + a u + b d of a slot, as the launch chain forms it (pc_core_step<false, false>), and |x - x'|^2
PC_HD double pc_slot_candidate(const PcSlot& T, double a, double b, double xp[4])
{
    double v[PC_NSUM], r0[6], j0[4];      // (r0, j0, the constants: not read in this form of the step)
    PcSys y;
    PcConst c;
    for (int q = 0; q < 4; ++q) { y.u[q] = T.u[q]; y.d[q] = T.d[q]; }
    v[SUM_STEP2] = 0.0;
    pc_core_step<false, false>(T.x, r0, j0, c, y, a, b, v, xp);
    return v[SUM_STEP2];
}

// iteration 0 has produced the track's constants, start values and the system there (PcInit of psfm_solver.hip): into the slot
PC_HD void pc_slot_fill(PcSlot& T, double s, const PcConst& c, const double x[4], const PcSys& y)
{
    T.s = s; T.S0q = c.S0q; T.S1q = c.S1q;
    for (int q = 0; q < 4; ++q) { T.x[q] = x[q]; T.u[q] = y.u[q]; T.d[q] = y.d[q]; }
}

// behind a separate iteration-0 launch: constants and start values came from memory, the taps at x0 are in tp -- the system at x0 for
// the mu in force (nothing is added to any sum: iteration 0's launch has reduced them)
PC_HD void pc_slot_start(PcSlot& T, const PcTaps& tp, double r1x, double r1y, double r2x, double r2y, double mu)
{
    double r[6], jac[4], unused[PC_NSUM];
    PcSys y;
    for (int q = 0; q < PC_NSUM; ++q) unused[q] = 0.0;
    const PcConst c = pc_slot_const(T);
    pc_core_eval_taps(tp, T.x, r1x, r1y, r2x, r2y, T.s, r, jac);
    pc_core_system<false>(T.x, r, jac, c, mu, pc_core_iA22(c, mu), unused, y, 0, 0);
    for (int q = 0; q < 4; ++q) { T.u[q] = y.u[q]; T.d[q] = y.d[q]; }
}

// A ROUND, evaluate-ahead form: the candidate of (a, b) -- recomputed here: the same operations as where its taps were requested,
// the same bits --, its cost and, ahead of the decision, the system there for the mu an accepted step leaves (mu_next).  The
// track's terms go to acc[] (ONE term per track, in list order, like the launches'); (u', d') at the candidate to next[8].
PC_HD void pc_slot_round(const PcSlot& T, const PcTaps& tp, double r1x, double r1y, double r2x, double r2y, double a, double b, double mu_next,
                         double acc[PC_NSUM], double next[8])
{
    double xe[4], r[6], jac[4];
    PcSys y;
    acc[SUM_STEP2] += pc_slot_candidate(T, a, b, xe);
    const PcConst c = pc_slot_const(T);
    pc_core_eval_taps(tp, xe, r1x, r1y, r2x, r2y, T.s, r, jac);
    acc[SUM_COST] += pc_core_cost(r);
    pc_core_system<true>(xe, r, jac, c, mu_next, pc_core_iA22(c, mu_next), acc, y, CH_QUD, CH_QDD);
    for (int q = 0; q < 4; ++q) { next[q] = y.u[q]; next[4 + q] = y.d[q]; }
}

// ... refresh form: the system at x for the mu an invalid step has raised; (u, d) of the slot are replaced
template <bool PAIR>
PC_HD void pc_slot_refresh(PcSlot& T, const PcF2* F12, int H, int W, double r1x, double r1y, double r2x, double r2y, double mu, double acc[PC_NSUM])
{
    double r[6], jac[4];
    PcSys y;
    const PcConst c = pc_slot_const(T);
    pc_core_eval<PAIR>(F12, H, W, T.x, r1x, r1y, r2x, r2y, T.s, r, jac);
    pc_core_system<true>(T.x, r, jac, c, mu, pc_core_iA22(c, mu), acc, y, CH_QUD, CH_QDD);
    for (int q = 0; q < 4; ++q) { T.u[q] = y.u[q]; T.d[q] = y.d[q]; }
}

// the round's step was ACCEPTED: x <- the candidate (the same operations once more: the same bits), (u, d) <- what was solved there
PC_HD void pc_slot_accept(PcSlot& T, double a, double b, const double next[8])
{
    double xp[4];
    (void)pc_slot_candidate(T, a, b, xp);
    for (int q = 0; q < 4; ++q) { T.x[q] = xp[q]; T.u[q] = next[q]; T.d[q] = next[4 + q]; }
}

// ---- the order in which the blocks' sums are added (every form of the chain: launches, the resident solve, the sharded export),
// with L = min(PC_LEADERS, n_blocks) and Q = ceil(ceil(n_blocks / PC_LEADERS) / 4):
//     total = ((t_0 + t_1) + t_2) + t_3,      t_j  = S_{8j} + S_{8j+1} + ... + S_{8j+7}  (those below L, in order)
//     S_x   = ((s_x0 + s_x1) + s_x2) + s_x3,  s_xj = the sums of blocks x + PC_LEADERS m, m in [j Q, (j + 1) Q), in order
// SUM_GMAX by max.  rows[b * pitch + k] = sum k of block b.  (The device runs this as two hops / as pc_reduce_totals.) ----
PC_HD int pc_tree_q(int n_blocks) { return ((n_blocks + PC_LEADERS - 1) / PC_LEADERS + 3) / 4; }
PC_HD void pc_tree_totals(const double* rows, int pitch, int n_blocks, int n_sums, double tot[PC_NSUM])
{
    const int L = n_blocks < PC_LEADERS ? n_blocks : PC_LEADERS;
    const int Q = pc_tree_q(n_blocks);
    for (int k = 0; k < PC_NSUM; ++k) {
        if (k >= n_sums) { tot[k] = 0.0; continue; }
        const bool mx = k == SUM_GMAX;
        double t[4];
        for (int j = 0; j < 4; ++j) {
            double tv = 0.0;
            for (int x = 8 * j; x < 8 * j + 8 && x < L; ++x) {
                const int cnt = (n_blocks - x + PC_LEADERS - 1) / PC_LEADERS;
                double sj[4];
                for (int jj = 0; jj < 4; ++jj) {
                    double v = 0.0;
                    for (int u = 0; u < Q; ++u) {
                        const int m = jj * Q + u;
                        if (m >= cnt) continue;
                        const double val = rows[(long)(x + PC_LEADERS * m) * pitch + k];
                        v = mx ? fmax(v, val) : v + val;
                    }
                    sj[jj] = v;
                }
                const double S = mx ? fmax(fmax(fmax(sj[0], sj[1]), sj[2]), sj[3]) : ((sj[0] + sj[1]) + sj[2]) + sj[3];
                tv = mx ? fmax(tv, S) : tv + S;
            }
            t[j] = tv;
        }
        tot[k] = mx ? fmax(fmax(fmax(t[0], t[1]), t[2]), t[3]) : ((t[0] + t[1]) + t[2]) + t[3];
    }
}

// ---- ONE solve over several ranks (PcPeers in psfm_solver.hip): every rank forms its total T_r with the tree above over ITS blocks'
// rows, and every block of every rank adds the ranks' totals in rank order,
//     total = (...((T_0 + T_1) + T_2) ...) + T_{world-1}            (SUM_GMAX by max)
// -- the same numbers in the same order everywhere.  rows: rank r's block rows start at rows + (n_blocks[0] + ... + n_blocks[r-1]) * pitch.
PC_HD void pc_peer_totals(const double* rows, int pitch, const int* n_blocks, int world, int n_sums, double tot[PC_NSUM])
{
    double Tr[PC_NSUM];
    long start = 0;
    for (int r = 0; r < world; ++r) {
        pc_tree_totals(rows + start * pitch, pitch, n_blocks[r], n_sums, Tr);
        for (int k = 0; k < PC_NSUM; ++k) tot[k] = r == 0 ? Tr[k] : (k == SUM_GMAX ? fmax(tot[k], Tr[k]) : tot[k] + Tr[k]);
        start += n_blocks[r];
    }
}

// ---- the end of a solve: which buffer the positions of the solve come from.  A failed solve hands the parameters back as they
// came in (buffer 0, which no form of the chain writes before its write-back); otherwise a slot writes its own x, a streamed
// entry is copied from the iterate buffer the control block names ----
PC_HD bool pc_res_moved(const PsfmSolveCtrl& C) { return C.cur != 0 && !C.failed; }
PC_HD int pc_res_stream_source(const PsfmSolveCtrl& C) { return pc_res_moved(C) ? C.cur : 0; }
