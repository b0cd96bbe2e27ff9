// nra_joint.hip -- junction decomposition of the 2D (joint) grid (gfx950).
//
// For one read and one k1, the templates L + u1^k1 + mid + u2^k2 + R over k2 share
// L + u1^k1 + mid + u2^k2 as a prefix and R as a suffix, exactly like the 1D bank
// (nra_sweep.hip), so one forward sweep per (read, k1) and one reverse sweep over rev(R)
// per read replace one full DP per (read, k1, k2) cell.
//
// The 2D selector needs the reference's CIGAR-window score of the co-optimal alignment
// (tk.py:435-500), so cells are int32 = (score << 16) | (window + 0x8000) and every max is
// the oracle's lexicographic (score, window) max.  Both parts are additive along a path, so
// the pair decomposes at the junction like the score alone:
//
//   reverse sweep (DIR 0): reversed read vs rev(R); window payload of the R side: the
//     window reaches wr = min(10, |R|) bases into R (nanoRepeat_joint.py:447-448).  A
//     deletion run charges -4 on its first overlapped base and -2 on the others
//     (tk.py:480-485); read backwards that is -4 on the first base met (or on R[wr-1] when the
//     run comes in from outside the window) and -2 after.  At the column of R[0] every row
//     stores H, E_in, E2_in; the running maximum is A = the best alignment inside R.
//   forward sweep: read vs L + u1^k1 + mid + u2^k2hi, window open from a = max(0,|L|-10)
//     on.  At every requested k2 boundary each row combines with the R side:
//        H + Hb,   E_in + Eb_in + (q, +2),   E2_in + E2b_in + (q2, +2)        (H >= (0, 0): the floor, below)
//     -- a gap spanning the junction gets one open refunded, and +2 of window score, because
//     both sides charged a "first overlapped base" --  and V = max(S, B, A) is the cell's
//     (score, window score).
//
// The forward sweeps of one read share L + u1^k1 across its k1 values too.  A systolic wave is a
// deterministic machine: its whole state is 3R+7 registers per lane.  So the forward sweep is cut
// in two:
//   prefix sweep (DIR 1), one per read: read vs L + u1^k1max; at step t = |L| + m1*k1 - 1, for
//     every k1 of the read's cells, all lanes store their registers (lane 0 is about to take the
//     LAST column of L + u1^k1; lane l is l columns behind, still inside the shared prefix);
//   tail sweep (DIR 2), one per (read, k1) run of cells: loads that state and carries on with
//     that last column followed by mid + u2^k2hi -- bit for bit the uninterrupted sweep over
//     L + u1^k1 + mid + u2^k2hi, for 1 + |mid| + m2*k2hi steps + one per lane the read occupies instead of
//     all of them.
//
// Columns without window payload.  The window takes the last 10 bases of L and the first 10 of R; before
// it every cell's payload is the constant 0, so the first |L| - 10 columns of the prefix sweep and the first
// |R| - 10 of the reverse sweep are plain score cells: k_joint_pk16 sweeps them in packed int16, TWO reads per
// wave at 14.5 instead of 2 x 15.5 instructions per row and column (the cell of the 1D sweeps, nra_pk16.h, no
// origin bit), and leaves the wave state -- every lane's rows and hand-off values as they stand when lane 0 is
// about to take the first window column -- for the int32 sweep of each read to resume from (`resume`).  The
// L side depends on the read alone, not on the cell list: it is kept from one grid round to the next.
//
// Preconditions checked on the host (else the cell goes to the brute-force kernel):
// |L| >= 1, |R| >= 2, the k2 values of a (read, k1) form an arithmetic progression.
#include "nra_pk16.h"

#ifndef NRA_PART
#define NRA_PART 0
#endif
#define NRA_HAS_PART(n) (NRA_PART == 0 || NRA_PART == (n))

#define JNEG (-(1 << 29))
#define JBIAS 0x8000
#define JFLAG_BOUNDARY 0x100
#define JFLAG_SNAPSHOT 0x200

// base of the strand-oriented read at oriented index idx (0..qlen-1)
template <bool HAS_N>
__device__ __forceinline__ int oriented_code(const NraDevRead& rd, const uint32_t* q2bit,
                                             const uint32_t* qnmask, int idx)
{
    if (idx < 0 || idx >= rd.qlen) return NRA_PAD_Q;
    uint32_t b = rd.qoff + (uint32_t)(rd.rc ? (rd.qlen - 1 - idx) : idx);
    int c = (q2bit[b >> 4] >> ((b & 15u) * 2u)) & 3u;
    if (rd.rc) c = 3 - c;
    if (HAS_N) {
        if ((qnmask[b >> 5] >> (b & 31u)) & 1u) c = NRA_CODE_N;
    }
    return c;
}

// Waves per SIMD to aim for (512 VGPRs per lane and SIMD): a sweep keeps 4 registers per row (H,
// E_in, E2_in, the read base) -- 7 in the tail sweep, which also holds the R side -- and ~45 more.
// Without the hint the compiler spends every register its occupancy bracket has (R = 18: 239).
// DIR: 0 reverse sweep over rev(R); 1 prefix sweep; 2 tail sweep; 3 reverse sweep extended over rev(u2)^k2hi with a
// column state per k2 (junction at the end of mid); 4 MID sweep (a tail that stops at the end of mid and leaves its
// column state); 5 prefix sweep that leaves COLUMN states -- every lane its registers as it is about to take the last
// column of L + u1^k1, on its own step -- for k_joint_midscan, which takes the MID part column by column.
constexpr int joint_waves(int R, int DIR)
{
    const int need = (DIR == 2 ? 7 : 4) * R + 70;
    const int w = 512 / need;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

template <int R, bool HAS_N, int DIR>
__global__ __launch_bounds__(WAVE, joint_waves(R, DIR)) void k_joint_sweep(int n_tasks, const NraJointTask* __restrict__ tasks,
                                                      const NraDevRead* __restrict__ reads,
                                                      const NraDevRegion* __restrict__ regions,
                                                      const uint8_t* __restrict__ pool,
                                                      const uint32_t* __restrict__ q2bit,
                                                      const uint32_t* __restrict__ qnmask,
                                                      NraScoreParams sp,
                                                      const int32_t* __restrict__ k1list,  // DIR 1: the read's k1 values, ascending
                                                      int32_t* __restrict__ state,     // wave states: (3R+7) x 64 per (read, k1)
                                                      int32_t* __restrict__ snap,      // 3 x int32 per read base
                                                      int32_t* __restrict__ read_a,    // A per read (packed)
                                                      int32_t* __restrict__ cell_score,
                                                      int32_t* __restrict__ cell_wscore,
                                                      const int32_t* __restrict__ pstate)   // DIR 0 / 1: packed states
{
    constexpr bool FWD = DIR == 1 || DIR == 2 || DIR >= 4;      // read vs the template left to right
    constexpr bool TAIL = DIR == 2 || DIR == 4;                 // resumes from a prefix sweep's wave state
    constexpr bool EXT = DIR == 3, MID = DIR == 4;
    constexpr bool PRE = DIR == 1 || DIR == 5, COLPRE = DIR == 5;
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    const NraJointTask tk = tasks[task];
    const NraDevRead rd = reads[tk.read];
    const NraDevRegion rg = regions[rd.region];
    const int Q = rd.qlen;
    const int lenR = rg.l3;
    const int wr = imin(10, lenR);                      // window bases inside R

    // template of this sweep, in the sweep's own column numbers v = 0, 1, ...:
    //   DIR 0: rev(R);  DIR 1: L + u1^k1max up to the last state dump;
    //   DIR 2: v = 0 is column t0 = |L| + m1*k1 - 1 of piece 1, then mid + u2^k2hi.
    const uint8_t* __restrict__ p1 = pool + (FWD ? rg.p1_off : rg.pr_off);
    const uint8_t* __restrict__ p2 = pool + rg.p2_off;
    const int t0 = TAIL ? rg.l1 + rg.m1 * tk.k1 - 1 : 0;                       // real column of v = 0
    int si = 0;                                                                // DIR 1: next dump
    auto t0_of = [&](int i) { return rg.l1 + rg.m1 * k1list[tk.k1_off + i] - 1; };   // last column of L + u1^k1_i
    int next_t = PRE ? t0_of(0) : -1;
    const int t_last = PRE ? t0_of(tk.nk1 - 1) : 0;
    const int cmid = 1 + rg.l2;                                                // columns of a MID sweep
    const int ncols = DIR == 0 ? lenR : PRE ? t_last : MID ? cmid
                      : (EXT ? lenR - 1 : rg.l2) + 1 + rg.m2 * (tk.k2lo + tk.k2step * (tk.n2 - 1));
    const int vfirst = (EXT ? lenR - 1 : rg.l2) + rg.m2 * tk.k2lo;             // DIR 2 / 3: first boundary
    const int vstep = rg.m2 * tk.k2step;
    const int wa = imax(0, rg.l1 - 10);                                        // window start (forward)
    constexpr int NSTATE = NRA_JOINT_NSTATE(R);

    // the lane's read bases, four per register (byte-select compares cost nothing extra)
    uint32_t qcp[(R + 3) / 4];
#pragma unroll
    for (int i = 0; i < (R + 3) / 4; ++i) qcp[i] = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int r = lane * R + i;
        qcp[i >> 2] |= (uint32_t)oriented_code<HAS_N>(rd, q2bit, qnmask, FWD ? r : (r < Q ? Q - 1 - r : -1)) << (8 * (i & 3));
    }

    const int sA = sp.match << 16, sB = -(sp.mismatch << 16), sN = -(sp.ambi << 16);
    const int o1 = -(sp.open1 << 16), x1 = -(sp.ext1 << 16);
    const int o2 = -(sp.open2 << 16), x2 = -(sp.ext2 << 16);
    const int fresh = JBIAS;

    // Cells keep H plus the cost of opening a vertical gap in their own column, Hq = H + fo1 (and Hq2 = H + fo2):
    // that is what F takes, and -- the open of a horizontal gap in the NEXT column costing the same but for the
    // window's first column (forward) or two (reverse) -- also what E of the next column takes, and the diagonal
    // and the floor take it with the constant folded into the substitution scores: 16.5 instructions per cell
    // instead of 20.  fo1, fo2 differ by the window payload (0 or -4) from column to column, never between the
    // two gap pieces: Hq2 = Hq + (o2 - o1).
    // The floor of the local alignment (an alignment may start anywhere) lives in the vertical-gap state, as in the
    // packed cell (nra_pk16.h, FF): F never falls below `fresh`, so every H is >= fresh and neither the diagonal
    // nor the junction combine needs a max with it; lane 0's inputs get the floor through the DPP moves' `old`
    // operand.  15.5 instructions per cell, + 5 per row on a tail sweep's boundary steps.
    const int o21 = o2 - o1;

    // forward sweep: the R side, row r pairs with reverse-sweep row Q-2-r.  The stored values carry
    // one JBIAS each; the combine adds two packed words, so one bias is taken out here, and the
    // gap-spanning terms get the refunds (one gap open, +2 window).  Boundary columns lie inside the window
    // (fo1 = o1 - 4): the forward H arrives as Hq, so the R side's H takes that constant out again.
    const int fo1_win = o1 - 4;
    int Hbo[DIR == 2 ? R : 1], Ebo[DIR == 2 ? R : 1], E2bo[DIR == 2 ? R : 1];
    if (DIR == 2) {
        const int q1 = (sp.open1 - sp.ext1) << 16, q2 = (sp.open2 - sp.ext2) << 16;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = lane * R + i;
            const int a = Q - 2 - r;
            if (a >= 0) {
                const int32_t* s3 = snap + ((size_t)rd.qoff + a) * 3;
                Hbo[i] = s3[0] - JBIAS - fo1_win;
                Ebo[i] = s3[1] - JBIAS + q1 + 2;
                E2bo[i] = s3[2] - JBIAS + q2 + 2;
            } else { Hbo[i] = JNEG; Ebo[i] = JNEG; E2bo[i] = JNEG; }
        }
    }

    // (Hq2 is kept in registers while a wave's 256 arch VGPRs hold 8 -- tail sweep -- or 5 values per row, else
    // recomputed from Hq where E2 needs it)
    constexpr bool KEEP_HQ2 = (DIR == 2 ? 8 : 5) * R + 45 <= 256;
    int Hq[R], Hq2[KEEP_HQ2 ? R : 1], E[R], E2[R];     // after a step: H(i,j) + fo1 / + fo2, and E_in / E2_in of column j
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = fresh + o1; E[i] = JNEG; E2[i] = JNEG; if (KEEP_HQ2) Hq2[i] = fresh + o2; }
    int Hbot = fresh + o1, Fout = fresh, F2out = JNEG, Hup_prev = fresh + o1;     // H = 0 left of and above the matrix
    int M = fresh;                      // running lexicographic max of this lane's cells ((0, 0) to start)
    int accS = JNEG, accB = JNEG;
    int tt = NRA_PAD_T;
    int j = t0 - lane;                  // real template column of this lane's next cell
    int ncur = 0;                       // boundary counter, meaningful in the output lane only
    int nsnap = 0;                      // EXT: boundary columns this lane has passed

    if (TAIL) {
        // resume: the registers of every lane as the prefix sweep left them before step t0
        const int32_t* __restrict__ sv = state + tk.state + lane;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            Hq[i] = sv[(size_t)i * 64]; E[i] = sv[(size_t)(R + i) * 64]; E2[i] = sv[(size_t)(2 * R + i) * 64];
            if (KEEP_HQ2) Hq2[i] = Hq[i] + o21;
        }
        Hbot = sv[(size_t)(3 * R) * 64]; Fout = sv[(size_t)(3 * R + 1) * 64]; F2out = sv[(size_t)(3 * R + 2) * 64];
        Hup_prev = sv[(size_t)(3 * R + 3) * 64]; M = sv[(size_t)(3 * R + 4) * 64];
        accB = sv[(size_t)(3 * R + 5) * 64]; tt = sv[(size_t)(3 * R + 6) * 64];
    }

    // resume behind k_joint_pk16: lane l has finished column step0 - 1 - l; values are (score + BIAS) halves,
    // H is stored minus the gap open there too (nra_pk16.h; no payload before the window), the accumulator
    // chains start empty (no boundary before step0)
    int step0 = 0;
    if (!TAIL) {
        if (tk.resume) {
            step0 = NRA_JOINT_PACKED_COLS(FWD ? rg.l1 : lenR);
            const int32_t* __restrict__ pv = pstate + tk.pstate + lane;
            const int hi = tk.phalf;
            auto cell_of = [&](int v) { return (((hi ? half_hi(v) : half_lo(v)) - BIAS) << 16) + JBIAS; };
#pragma unroll
            for (int i = 0; i < R; ++i) {
                Hq[i] = cell_of(pv[(size_t)i * 64]);
                if (KEEP_HQ2) Hq2[i] = Hq[i] + o21;
                E[i] = cell_of(pv[(size_t)(R + i) * 64]);
                E2[i] = cell_of(pv[(size_t)(2 * R + i) * 64]);
            }
            Hup_prev = cell_of(pv[(size_t)(3 * R) * 64]);
            Fout = cell_of(pv[(size_t)(3 * R + 1) * 64]);
            F2out = cell_of(pv[(size_t)(3 * R + 2) * 64]);
            M = cell_of(pv[(size_t)(3 * R + 3) * 64]);
            Hbot = Hq[R - 1];
            const int col = step0 - 1 - lane;
            tt = col >= 0 ? (int)p1[col] : NRA_PAD_T;
            j = step0 - lane;
        }
    }

    // DIR 0 / 2 drain the pipeline as far as the read goes: the lanes below its last row hold padding rows only
    // (every value there lies below one of a real row), so the boundary outputs leave through the lane of the
    // last row and the loop ends `last_lane` steps after the last column; DIR 1 leaves at its last dump (step t_last)
    // One flat step loop (a chunk loop around a 64-step loop made the compiler keep two copies of
    // the row registers).  Every 64 steps the lanes fetch the next 64 template columns.
    const int last_lane = imin(63, imax(Q - 1, 0) / R);
    const int nsteps = DIR == 1 ? ncols + 1 : COLPRE ? ncols + last_lane + 1 : ncols + last_lane;
    int my_next = COLPRE && lane <= last_lane ? next_t + lane : 0x7fffffff;    // COLPRE: the step of this lane's next dump
    int feed = NRA_PAD_T;
#pragma unroll 1   // unrolling the step loop twice takes minutes to compile at R >= 20
    for (int step = step0; step < nsteps; ++step) {
        {
            if (((step - step0) & 63) == 0) {
                const int col = step + lane;
                feed = NRA_PAD_T;
                if (col < ncols) {
                    if (TAIL) {
                        feed = col == 0 ? p1[t0] : p2[col - 1];
                        if (MID) { if (col == ncols - 1) feed |= JFLAG_SNAPSHOT | JFLAG_BOUNDARY; }
                        else if (col >= vfirst && (col - vfirst) % vstep == 0 && (col - vfirst) / vstep < tk.n2) feed |= JFLAG_BOUNDARY;
                    } else {
                        feed = p1[col];          // (EXT: rev(R) is followed by rev(u2)^k2max in the pool)
                        if (DIR == 0 && col == lenR - 1) feed |= JFLAG_SNAPSHOT | JFLAG_BOUNDARY;
                        if (EXT && col >= vfirst && (col - vfirst) % vstep == 0) feed |= JFLAG_SNAPSHOT | JFLAG_BOUNDARY;
                    }
                }
            }
            if (DIR == 1) {
                if (step == next_t) {                // wave-uniform: dump the wave before this step
                    int32_t* __restrict__ sv = state + tk.state + (size_t)si * (NSTATE * 64) + lane;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        sv[(size_t)i * 64] = Hq[i]; sv[(size_t)(R + i) * 64] = E[i]; sv[(size_t)(2 * R + i) * 64] = E2[i];
                    }
                    sv[(size_t)(3 * R) * 64] = Hbot; sv[(size_t)(3 * R + 1) * 64] = Fout; sv[(size_t)(3 * R + 2) * 64] = F2out;
                    sv[(size_t)(3 * R + 3) * 64] = Hup_prev; sv[(size_t)(3 * R + 4) * 64] = M;
                    sv[(size_t)(3 * R + 5) * 64] = accB; sv[(size_t)(3 * R + 6) * 64] = tt;
                    if (++si >= tk.nk1) break;
                    next_t = rg.l1 + rg.m1 * k1list[tk.k1_off + si] - 1;
                }
            }
            if (COLPRE) {
                if (step == my_next) {               // this lane is about to take the last column of L + u1^k1: its column state
                    // lane-major, 16-byte pieces: [Hq | E_in | E2_in | Hup_prev, M] (one lane stores per step: wide stores)
                    constexpr int N4 = NRA_JOINT_COLSTATE(R) / 4;
                    int4* __restrict__ sv = reinterpret_cast<int4*>(state + tk.state + (size_t)si * (NSTATE * 64) + (size_t)lane * (4 * N4));
                    int v[4 * N4];
#pragma unroll
                    for (int i = 0; i < 4 * N4; ++i)
                        v[i] = i < R ? Hq[i] : i < 2 * R ? E[i - R] : i < 3 * R ? E2[i - 2 * R] : i == 3 * R ? Hup_prev : i == 3 * R + 1 ? M : 0;
#pragma unroll
                    for (int i = 0; i < N4; ++i) sv[i] = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
                    ++si;
                    my_next = si < tk.nk1 ? t0_of(si) + lane : 0x7fffffff;
                }
            }
            int F = dpp_shr1(fresh, Fout);
            int F2 = dpp_shr1(JNEG, F2out);
            tt = dpp_shr1(feed, tt);
            feed = dpp_rol1(feed);
            const int accS_in = dpp_shr1(JNEG, accS);
            const int accB_in = dpp_shr1(JNEG, accB);
            const int tcode = tt & 0xff;

            // window payload increments of THIS column (tk.py:464-485; mirrored for the reverse sweep), and the
            // vertical-gap open of the PREVIOUS column, which the stored Hq carry (fo_prev)
            int pe, pn, eo, ex, fo, fx, fo_prev;
            if (FWD) {
                const bool inw = j >= wa;                                    // every column from wa on
                pe = inw ? 2 : 0; pn = inw ? -4 : 0;
                eo = inw ? -4 : 0; ex = inw ? (j == wa ? -4 : -2) : 0;        // deletion onto base j
                fo = inw ? -4 : 0; fx = inw ? -2 : 0;                         // insertion at ref_pos = j+1 > wa
                // The one column where opening a horizontal gap (eo of column wa) and the vertical open of the
                // column before (0) differ: the lane re-bases what it holds of column wa - 1, once per sweep.
                fo_prev = inw ? -4 : 0;
                if (j == wa) {
#pragma unroll
                    for (int i = 0; i < R; ++i) { Hq[i] -= 4; if (KEEP_HQ2) Hq2[i] -= 4; }
                    Hup_prev -= 4;
                }
            } else {
                const int p = lenR - 1 - j;                                  // forward position inside R (EXT: < 0 inside u2^k2)
                const bool real = EXT ? j < ncols : p >= 0;                  // a template column, not pipeline padding
                const bool inw = p < wr && real;
                pe = inw ? 2 : 0; pn = inw ? -4 : 0;
                eo = inw ? -4 : 0; ex = inw ? (p == wr - 1 ? -4 : -2) : 0;   // read backwards: first base met
                const bool fin = p < wr - 1 && real;                         // insertion before R[p] (or inside u2^k2)
                fo = fin ? -4 : 0; fx = fin ? -2 : 0;
                fo_prev = (p + 1 < wr - 1 && (EXT ? j - 1 < ncols : p + 1 >= 0)) ? -4 : 0;
            }
            const int fo1p = o1 + fo_prev;
            const int s_eq = sA + pe - fo1p, s_ne = sB + pn - fo1p, n_eq = sN + pe - fo1p, n_ne = sN + pn - fo1p;
            const int ex1 = x1 + ex, ex2 = x2 + ex;
            const int fo1 = o1 + fo, fx1 = x1 + fx, fx2 = x2 + fx;
            const int dq = !FWD ? eo - fo_prev : 0;                          // reverse: two columns at the window's edge
#define NRA_SUBST(i, out)                                                                          \
            {                                                                                      \
                const int qc_ = (int)((qcp[(i) >> 2] >> (8 * ((i) & 3))) & 0xffu);                 \
                const bool eq_ = qc_ == tcode;                                                     \
                out = eq_ ? s_eq : s_ne;                                                           \
                if (HAS_N) {                                                                       \
                    if ((qc_ | tcode) & 4) out = eq_ ? n_eq : n_ne;                                \
                }                                                                                  \
            }
            int sc;
            NRA_SUBST(0, sc);
            int d = Hup_prev + sc;
            Hup_prev = dpp_shr1(fresh + fo1, Hbot);              // lane 0: the row above the read, H = 0
            int h = JNEG, h_prev = JNEG;
            const bool at_boundary = (tt & JFLAG_BOUNDARY) != 0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                int d_next = d;
                if (i + 1 < R) {
                    NRA_SUBST(i + 1, sc);
                    d_next = Hq[i] + sc;
                }
                // E(i,j) from column j-1, lazily: E_in stays in the register for the combine
                const int ein = !FWD ? imax(E[i] + ex1, Hq[i] + dq) : imax(E[i] + ex1, Hq[i]);
                const int hq2_prev = KEEP_HQ2 ? Hq2[i] : Hq[i] + o21;
                const int e2in = !FWD ? imax(E2[i] + ex2, hq2_prev + dq) : imax(E2[i] + ex2, hq2_prev);
                h = imax(imax(d, ein), F);
                h = imax(imax(h, e2in), F2);
                if (i & 1) M = imax(imax(M, h_prev), h);                     // two rows per 3-input max
                else if (i == R - 1) M = imax(M, h);
                else h_prev = h;
                E[i] = ein;
                E2[i] = e2in;
                const int hq = h + fo1;
                Hq[i] = hq;
                const int hq2 = hq + o21;
                if (KEEP_HQ2) Hq2[i] = hq2;
                F = imax(imax(F + fx1, hq), fresh);
                F2 = imax(F2 + fx2, hq2);
                d = d_next;
            }
#undef NRA_SUBST
            int tS = JNEG;
            if (DIR == 2) {
                if (__builtin_amdgcn_ballot_w64(at_boundary) != 0) {
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int t1 = Hq[i] + Hbo[i];
                        const int t2 = E[i] + Ebo[i];
                        const int t3 = E2[i] + E2bo[i];
                        tS = imax(imax(tS, t1), imax(t2, t3));
                    }
                }
            } else if (DIR == 0 && (tt & JFLAG_SNAPSHOT)) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int a = lane * R + i;
                    if (a < Q) {
                        int32_t* s3 = snap + ((size_t)rd.qoff + a) * 3;
                        s3[0] = Hq[i] - fo1; s3[1] = E[i]; s3[2] = E2[i];
                    }
                }
            } else if ((EXT || MID) && (tt & JFLAG_SNAPSHOT)) {
                // the lane's rows as they stand on a junction column: planes [H | E_in | E2_in] of Q rows.  EXT: slot
                // nsnap (the lane's own count of boundaries passed) of the read's k2 list, H without the vertical open;
                // MID: the one slot of this (read, k1), values as the junction combine takes them
                if (EXT) {
                    // planes of 64 * R rows (the padding rows too: no guard; nobody reads them), the lane's R rows of a plane
                    // in one piece: 16-byte stores where R allows (a lane is alone on its column: nothing to coalesce with)
                    int32_t* __restrict__ dst = snap + tk.state + (size_t)nsnap * (3 * 64 * R) + lane * R;
                    if (R % 4 == 0) {
#pragma unroll
                        for (int i = 0; i < R / 4 * 4; i += 4) {
                            *reinterpret_cast<int4*>(dst + i) = make_int4(Hq[i] - fo1, Hq[i + 1] - fo1, Hq[i + 2] - fo1, Hq[i + 3] - fo1);
                            *reinterpret_cast<int4*>(dst + 64 * R + i) = make_int4(E[i], E[i + 1], E[i + 2], E[i + 3]);
                            *reinterpret_cast<int4*>(dst + 2 * 64 * R + i) = make_int4(E2[i], E2[i + 1], E2[i + 2], E2[i + 3]);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < R; ++i) { dst[i] = Hq[i] - fo1; dst[64 * R + i] = E[i]; dst[2 * 64 * R + i] = E2[i]; }
                    }
                } else {
                    int32_t* dst = snap + tk.pstate;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int a = lane * R + i;
                        if (a < Q) { dst[a] = Hq[i]; dst[Q + a] = E[i]; dst[2 * Q + a] = E2[i]; }
                    }
                }
                ++nsnap;
            }
            Hbot = Hq[R - 1]; Fout = F; F2out = F2;
            accS = imax(accS_in, tS);
            accB = imax(accB_in, M);

            if (!PRE && lane == last_lane && at_boundary) {
                if (DIR == 0) {
                    read_a[tk.read] = accB;                  // best alignment inside R (packed)
                } else if (EXT) {
                    read_a[tk.out + ncur++] = accB;          // A(k2): best alignment inside u2^k2 + R
                } else if (MID) {
                    read_a[tk.out] = accB;                   // B(k1): best alignment inside L + u1^k1 + mid
                } else {
                    const int n = ncur++;
                    const int A = read_a[tk.read];
                    const int V = imax(imax(accS, accB), A);
                    const int scv = V >> 16;
                    const int lo = sp.min_score > 1 ? sp.min_score : 1;
                    const int idx = tk.out + n;
                    if (scv >= lo) { cell_score[idx] = scv; cell_wscore[idx] = (V & 0xffff) - JBIAS; }
                    else { cell_score[idx] = -1; cell_wscore[idx] = 0; }
                }
            }
            ++j;
        }
    }
}

// ------------------------------------------------------------------------------------
// k_joint_pk16: the payload-free columns of a prefix (dir 1) or reverse (dir 0) sweep, two reads per wave in
// packed int16 cells.  One systolic cell per lane, skew 1, lane-to-lane hand-off through a one-slot LDS ring
// (k_sweep_ring's, nra_sweep.hip).  Runs exactly `cols` steps: lane l ends on column cols - 1 - l.
template <int R, bool HAS_N>
__global__ __launch_bounds__(WAVE) void k_joint_pk16(int n_tasks, const NraJointPairTask* __restrict__ tasks,
                                                     const NraDevRead* __restrict__ reads,
                                                     const NraDevRegion* __restrict__ regions,
                                                     const uint8_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ q2bit,
                                                     const uint32_t* __restrict__ qnmask,
                                                     NraScoreParams sp, int dir_arg,
                                                     int32_t* __restrict__ pstate_arg,
                                                     int32_t* __restrict__ pstate_l)
{
    __shared__ int4 ring[64];
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    NraJointPairTask tk = tasks[task];
    // dir_arg < 0: a launch that mixes the two sides, L-side tasks marked in bit 63 of their state index (states at pstate_l)
    const int dir = dir_arg < 0 ? (int)(tk.state >> 63) : dir_arg;
    int32_t* __restrict__ pstate = dir_arg < 0 && dir ? pstate_l : pstate_arg;
    tk.state &= ~(1ull << 63);
    const int ra = tk.read_a, rb = tk.read_b >= 0 ? tk.read_b : tk.read_a;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[rda.region];
    const uint8_t* __restrict__ piece = pool + (dir ? rg.p1_off : rg.pr_off);
    const int cols = NRA_JOINT_PACKED_COLS(dir ? rg.l1 : rg.l3);

    const int o1 = sp.open1, o2 = sp.open2;
    const int P1 = 0x00010001;
    const int v_floor = (BIAS - o1) * P1;
    const int v_o1 = o1 * P1, v_e1 = sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = sp.ext2 * P1;
    const int NEG1 = NEGB * P1;
    // substitution scores + o1 (the diagonal is read from Hq = H - o1): all in [0, 127] (host: joint_pack_ok)
    const int s_match = sp.match + o1, s_mis = o1 - sp.mismatch, s_ambi = o1 - sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);               // selector 4: padding row, 5: N in the read
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;
    auto column_table = [&](int col) {
        int t = tbl_mis4;
        if (col >= 0 && col < cols) {
            const int code = piece[col];
            t = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
        }
        return t;
    };
    auto selector = [&](const NraDevRead& rd, int r) {
        const int c = oriented_code<HAS_N>(rd, q2bit, qnmask, dir ? r : (r < rd.qlen ? rd.qlen - 1 - r : -1));
        return c == NRA_PAD_Q ? 4 : (c == NRA_CODE_N ? 5 : c);
    };
    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i)
        qc[i] = selector(rda, lane * R + i) | (0x0c << 8) | (selector(rdb, lane * R + i) << 16) | (0x0c << 24);

    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }
    ring[lane] = make_int4(v_floor, NEG1, NEG1, lane == 0 ? column_table(0) : tbl_mis4);
    ring_order();

    int Hup_prev = v_floor, M = BIAS * P1;
    int F = NEG1, F2 = NEG1;
    int feed = tbl_mis4;
    const int wr = (lane + 1) & 63;
#pragma unroll 1
    for (int step = 0; step < cols; ++step) {
        if ((step & 63) == 0) feed = column_table(step + 1 + wr);      // lane 63 hands out column step + 1
        const int4 in = ring[lane];
        F = pmaxi(in.y, BIAS * P1); F2 = in.z;              // the floor lives in F (sweep_cell, FF); lane 0 takes constants
        sweep_cell<0, R, R, false, true>(Hq, Hq2, E, E2, qc, Hup_prev, F, F2, M, in.w, tbl_hi, BIAS * P1, v_e1, v_e2, v_o1, v_o2);
        Hup_prev = pmaxi(in.x, v_floor);
        ring[wr] = make_int4(Hq[R - 1], F, F2, in.w);
        if (lane == 63) ring[0] = make_int4(v_floor, NEG1, NEG1, feed);
        ring_order();                                       // the next step's load stays behind these stores
        feed = dpp_rol1(feed);
    }
    int32_t* __restrict__ pv = pstate + tk.state + lane;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        pv[(size_t)i * 64] = Hq[i]; pv[(size_t)(R + i) * 64] = E[i]; pv[(size_t)(2 * R + i) * 64] = E2[i];
    }
    pv[(size_t)(3 * R) * 64] = Hup_prev; pv[(size_t)(3 * R + 1) * 64] = F; pv[(size_t)(3 * R + 2) * 64] = F2;
    pv[(size_t)(3 * R + 3) * 64] = M;
}

// ------------------------------------------------------------------------------------
template <int DIR>
static int launch_joint(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                        const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                        const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                        const int32_t* k1list, int32_t* state,
                        int32_t* snap, int32_t* read_a, int32_t* cell_score, int32_t* cell_wscore,
                        const int32_t* pstate)
{
    if (n_tasks <= 0) return 0;
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_joint_sweep<r, true, DIR><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state, snap, read_a, cell_score, cell_wscore, pstate); \
        else k_joint_sweep<r, false, DIR><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state, snap, read_a, cell_score, cell_wscore, pstate);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}

#if NRA_HAS_PART(7)
extern "C" int nra_launch_joint_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                    const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                    const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                    int32_t* snap, int32_t* read_a, const int32_t* pstate)
{
    return launch_joint<0>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, nullptr, nullptr,
                           snap, read_a, nullptr, nullptr, pstate);
}
#endif
#if NRA_HAS_PART(8)
extern "C" int nra_launch_joint_prefix(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                       const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                       const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                       const int32_t* k1list, int32_t* state, const int32_t* pstate)
{
    return launch_joint<1>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state,
                           nullptr, nullptr, nullptr, nullptr, pstate);
}
#endif
#if NRA_HAS_PART(10)
extern "C" int nra_launch_joint_tail(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                     const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                     const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                     int32_t* state, int32_t* snap, int32_t* read_a, int32_t* cell_score,
                                     int32_t* cell_wscore)
{
    return launch_joint<2>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, nullptr, state,
                           snap, read_a, cell_score, cell_wscore, nullptr);
}
#endif
// ------------------------------------------------------------------------------------
// k_joint_midscan: the MID part of a routed grid -- the last prefix column and `mid`, 1 + |mid| columns per (read, k1) --
// column by column with ALL rows at once instead of as a systolic sweep (whose pipeline fill and drain, one step per
// lane, are four fifths of a 14-column sweep).  One wave per read, its k1 values in turn; lane l holds rows
// [l*R, l*R+R).  Input: the column states the prefix sweep left (DIR 5: every lane's registers as it is about to
// take the last column of L + u1^k1); output: the column state at the end of mid (for k_joint_combine) and B(k1).
//
// A column: the diagonal and the two horizontal-gap states are row-local (the row above is one register or one DPP
// move away); the two vertical-gap states are max-plus recurrences down the rows,
//     F(g) = max(F(g-1) + fx, Hn(g-1) + fo [, floor]),     Hn = max(diagonal, E, E2)  (the cell without its vertical gaps)
// -- H(g-1) in place of Hn(g-1) adds only "a vertical gap right after a vertical gap", which a single gap of the
// cheaper-to-extend piece beats strictly -- and so prefix maxima:  F(g) = fx*(g-1) + max_{k<g} (Hn(k) + fo - fx*k):
// an in-lane running maximum, one exclusive scan across the lanes (DPP row shifts and broadcasts), two adds.
__device__ __forceinline__ int scan_excl_max(int x)        // exclusive prefix maximum across the 64 lanes (lane 0: JNEG * 2)
{
    const int NEG = 2 * JNEG;
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x111 /*row_shr:1*/, 0xf, 0xf, false));
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x112 /*row_shr:2*/, 0xf, 0xf, false));
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x114 /*row_shr:4*/, 0xf, 0xf, false));
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x118 /*row_shr:8*/, 0xf, 0xf, false));
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x142 /*row_bcast:15*/, 0xa, 0xf, false));
    x = imax(x, __builtin_amdgcn_update_dpp(NEG, x, 0x143 /*row_bcast:31*/, 0xc, 0xf, false));
    return __builtin_amdgcn_update_dpp(NEG, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}

template <int R, bool HAS_N>
__global__ __launch_bounds__(WAVE) void k_joint_midscan(int n_tasks, const NraJointTask* __restrict__ tasks,
                                                        const NraDevRead* __restrict__ reads,
                                                        const NraDevRegion* __restrict__ regions,
                                                        const uint8_t* __restrict__ pool,
                                                        const uint32_t* __restrict__ q2bit,
                                                        const uint32_t* __restrict__ qnmask, NraScoreParams sp,
                                                        const int32_t* __restrict__ k1list,
                                                        const int32_t* __restrict__ state,
                                                        int32_t* __restrict__ fsnap, int32_t* __restrict__ fb,
                                                        const NraGridRow* __restrict__ rows)
{
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    const NraJointTask tk = tasks[task];
    // rows != NULL (a refinement routed on the device, k_joint_refine_route): the task is the read's k1 values number
    // k1_off ... k1_off + nk1 - 1 of its row (unit step), as many of them as the row has
    int nk1 = tk.nk1, k1_first = 0;
    if (rows) {
        const NraGridRow row = rows[tk.read];
        nk1 = imin(nk1, row.n1 - tk.k1_off);
        k1_first = row.k1lo + tk.k1_off;
        if (nk1 <= 0) return;
    }
    const NraDevRead rd = reads[tk.read];
    const NraDevRegion rg = regions[rd.region];
    const int Q = rd.qlen;
    const uint8_t* __restrict__ p1 = pool + rg.p1_off;
    const uint8_t* __restrict__ p2 = pool + rg.p2_off;
    const int wa = imax(0, rg.l1 - 10);
    const int cmid = 1 + rg.l2;
    constexpr int NSTATE = NRA_JOINT_NSTATE(R);
    const int last_lane = imin(63, imax(Q - 1, 0) / R);

    uint32_t qcp[(R + 3) / 4];
#pragma unroll
    for (int i = 0; i < (R + 3) / 4; ++i) qcp[i] = 0;
#pragma unroll
    for (int i = 0; i < R; ++i)
        qcp[i >> 2] |= (uint32_t)oriented_code<HAS_N>(rd, q2bit, qnmask, lane * R + i) << (8 * (i & 3));

    const int sA = sp.match << 16, sB = -(sp.mismatch << 16), sN = -(sp.ambi << 16);
    const int o1 = -(sp.open1 << 16), x1 = -(sp.ext1 << 16);
    const int o2 = -(sp.open2 << 16), x2 = -(sp.ext2 << 16);
    const int fresh = JBIAS;
    const int o21 = o2 - o1;
    const int g0 = lane * R;                                  // the lane's first row
    __shared__ int tr[R * 64];                                // one plane of a column state, to transpose it for the store

    for (int s = 0; s < nk1; ++s) {
        const int k1 = rows ? k1_first + s : k1list[tk.k1_off + s];
        const int t0 = rg.l1 + rg.m1 * k1 - 1;
        const int slot = (k1 - tk.k1) / tk.k2step;           // (the prefix sweep may have left more column states than this list)
        constexpr int N4 = NRA_JOINT_COLSTATE(R) / 4;
        const int4* __restrict__ sv = reinterpret_cast<const int4*>(state + tk.state + (size_t)slot * (NSTATE * 64) + (size_t)lane * (4 * N4));
        int cs[4 * N4];
#pragma unroll
        for (int i = 0; i < N4; ++i) { const int4 q = sv[i]; cs[4 * i] = q.x; cs[4 * i + 1] = q.y; cs[4 * i + 2] = q.z; cs[4 * i + 3] = q.w; }
        int Hq[R], E[R], E2[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { Hq[i] = cs[i]; E[i] = cs[R + i]; E2[i] = cs[2 * R + i]; }
        int Hup = cs[3 * R], M = cs[3 * R + 1];
        int fo1_last = o1;
        for (int v = 0; v < cmid; ++v) {
            const int j = t0 + v;
            const int tcode = v == 0 ? p1[t0] : p2[v - 1];
            // window payload of this column (k_joint_sweep, forward rules)
            const bool inw = j >= wa;
            const int pe = inw ? 2 : 0, pn = inw ? -4 : 0;
            const int ex = inw ? (j == wa ? -4 : -2) : 0;
            const int fo = inw ? -4 : 0, fx = inw ? -2 : 0;
            const int fo_prev = inw ? -4 : 0;
            // the diagonal of the lane's first row: H of the row above it in the PREVIOUS column = the last row of the lane
            // above as it stands now (the first column's comes with the column state; above the read H = 0)
            if (v > 0) Hup = dpp_shr1(fresh + fo1_last, Hq[R - 1]);
            if (j == wa) {
#pragma unroll
                for (int i = 0; i < R; ++i) Hq[i] -= 4;
                Hup -= 4;
            }
            const int fo1p = o1 + fo_prev;
            const int s_eq = sA + pe - fo1p, s_ne = sB + pn - fo1p, n_eq = sN + pe - fo1p, n_ne = sN + pn - fo1p;
            const int ex1 = x1 + ex, ex2 = x2 + ex;
            const int fo1 = o1 + fo, fo2 = fo1 + o21, fx1 = x1 + fx, fx2 = x2 + fx;
            int hn[R];
            int run1 = 2 * JNEG, run2 = 2 * JNEG;                      // in-lane running maxima of Hn(k) + fo - fx*k
            int c1[R], c2[R];
            int d = Hup;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int qc_ = (int)((qcp[i >> 2] >> (8 * (i & 3))) & 0xffu);
                const bool eq_ = qc_ == tcode;
                int sc = eq_ ? s_eq : s_ne;
                if (HAS_N) { if ((qc_ | tcode) & 4) sc = eq_ ? n_eq : n_ne; }
                const int dd = d + sc;
                d = Hq[i];                                             // the diagonal of the row below: H(i, j-1)
                const int ein = imax(E[i] + ex1, Hq[i]);
                const int e2in = imax(E2[i] + ex2, Hq[i] + o21);
                E[i] = ein; E2[i] = e2in;
                hn[i] = imax(imax(dd, ein), e2in);
                c1[i] = run1; c2[i] = run2;                            // maxima over the lane's rows ABOVE row i
                run1 = imax(run1, hn[i] + fo1 - fx1 * (g0 + i));
                run2 = imax(run2, hn[i] + fo2 - fx2 * (g0 + i));
            }
            fo1_last = fo1;
            const int above1 = scan_excl_max(run1), above2 = scan_excl_max(run2);     // ... over the rows of the lanes above
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int g = g0 + i;
                const int F = imax(fresh, imax(above1, c1[i]) + fx1 * (g - 1));
                const int F2 = imax(above2, c2[i]) + fx2 * (g - 1);
                const int h = imax(imax(hn[i], F), F2);
                M = imax(M, h);
                Hq[i] = h + fo1;
            }
        }
        // the column state at the end of mid, planes [Hq | E_in | E2_in] of Q rows, and B(k1)
        // (through LDS: a lane holds R consecutive rows, a store instruction should cover 64 consecutive ones -- written
        // straight from the registers the 80-byte pieces of neighbouring lanes share lines and the counter read 2.6 x the bytes)
        int32_t* __restrict__ fs = fsnap + tk.pstate + (size_t)s * 3 * Q;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < R; ++i) tr[g0 + i] = pl == 0 ? Hq[i] : pl == 1 ? E[i] : E2[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int a = 64 * i + lane;
                if (a < Q) fs[(size_t)pl * Q + a] = tr[a];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        int best = lane <= last_lane ? M : 2 * JNEG;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = imax(best, __shfl_xor(best, off, 64));
        if (lane == 0) fb[tk.out + s] = best;
    }
}

#if NRA_HAS_PART(22)
extern "C" int nra_launch_joint_prefix_cols(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                            const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                            const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                            const int32_t* k1list, int32_t* state, const int32_t* pstate)
{
    return launch_joint<5>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state,
                           nullptr, nullptr, nullptr, nullptr, pstate);
}
#endif
#if NRA_HAS_PART(23)
extern "C" int nra_launch_joint_midscan(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                        const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                        const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                        const int32_t* k1list, const int32_t* state, int32_t* fsnap, int32_t* fb,
                                        const NraGridRow* rows)
{
    if (n_tasks <= 0) return 0;
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_joint_midscan<r, true><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state, fsnap, fb, rows); \
        else k_joint_midscan<r, false><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, k1list, state, fsnap, fb, rows);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}
#endif
#if NRA_HAS_PART(20)
extern "C" int nra_launch_joint_bwd_ext(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                        const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                        const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                        int32_t* rsnap, int32_t* ra, const int32_t* pstate)
{
    return launch_joint<3>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, nullptr, nullptr,
                           rsnap, ra, nullptr, nullptr, pstate);
}
#endif
#if NRA_HAS_PART(21)
extern "C" int nra_launch_joint_mid(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                    const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                    const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                    int32_t* state, int32_t* fsnap, int32_t* fb)
{
    return launch_joint<4>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, nullptr, state,
                           fsnap, fb, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------
// k_joint_combine: the cells of a routed grid from the column states on either side of the junction at the end of mid.
// One wave per read; rows across the lanes (row = 64*t + lane: coalesced plane loads).  Cell (k1_i, k2_n):
//   S = max over rows r of  Hq_f(r) + H_b(a) - fo1_win,  E_f(r) + E_b(a) + (q, +2),  E2_f(r) + E2_b(a) + (q2, +2),   a = Q-2-r
//   V = max(S, B(k1_i), A(k2_n))  ->  score, window score        (the arithmetic of k_joint_sweep's tail combine)
template <int IB, int NB>
__device__ __forceinline__ void joint_combine_tiles(const NraJointCombineTask& tk, const int Q, const int lane, NraScoreParams sp,
                                                    const int32_t* __restrict__ fsnap, const int32_t* __restrict__ rsnap,
                                                    const int32_t* __restrict__ fb, const int32_t* __restrict__ ra,
                                                    int32_t* __restrict__ cell_score, int32_t* __restrict__ cell_wscore)
{
    const int o1 = -(sp.open1 << 16);
    const int fo1_win = o1 - 4;
    const int q1 = (sp.open1 - sp.ext1) << 16, q2 = (sp.open2 - sp.ext2) << 16;
    const int cH = -JBIAS - fo1_win, cE = -JBIAS + q1 + 2, cE2 = -JBIAS + q2 + 2;      // as the tail sweep's prologue
    const int lo = sp.min_score > 1 ? sp.min_score : 1;
    // Tiles of IB k1 values x NB k2 values, rows in chunks of CH per lane (128 rows): a chunk of both sides stays in
    // registers for the tile's IB * NB cells -- every plane is read once per tile row / column instead of once per cell
    // (8.7 -> ~4 GB per config-3 step) --, its loads are independent (one dependent trip per cell and row group makes the
    // loop latency-bound), a lane keeps its partial maxima of the tile's cells and the wave reduces once per cell.
    constexpr int CH = 2;
    for (int n0 = 0; n0 < tk.n2; n0 += NB) {
        for (int i0 = 0; i0 < tk.n1; i0 += IB) {
            int tS[IB][NB];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) tS[ii][nn] = JNEG;
            for (int r0 = 0; r0 < Q - 1; r0 += 64 * CH) {
                int hb[NB][CH], eb[NB][CH], e2b[NB][CH];
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    const int n = imin(n0 + nn, tk.n2 - 1);           // (past the last k2 / k1: the last one again, not written)
                    const int P = tk.rs_plane;                        // rows of an R-side plane (padded to the wave's 64 * R)
                    const int32_t* __restrict__ rs = rsnap + tk.rs + (size_t)(tk.rs_first + n * tk.rs_stride) * 3 * P;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int r = r0 + 64 * c + lane;
                        const bool ok = r < Q - 1;                    // row Q-1 has no partner on the R side
                        const int a = ok ? Q - 2 - r : 0;
                        hb[nn][c] = ok ? rs[a] + cH : JNEG; eb[nn][c] = ok ? rs[P + a] + cE : JNEG;
                        e2b[nn][c] = ok ? rs[2 * P + a] + cE2 : JNEG;
                    }
                }
#pragma unroll
                for (int ii = 0; ii < IB; ++ii) {
                    const int i = imin(i0 + ii, tk.n1 - 1);
                    const int32_t* __restrict__ fs = fsnap + tk.fs + (size_t)i * 3 * Q;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int r = r0 + 64 * c + lane;
                        const int rr = r < Q ? r : 0;
                        const int hf = fs[rr], ef = fs[Q + rr], e2f = fs[2 * Q + rr];
#pragma unroll
                        for (int nn = 0; nn < NB; ++nn)
                            tS[ii][nn] = imax(imax(tS[ii][nn], hf + hb[nn][c]), imax(ef + eb[nn][c], e2f + e2b[nn][c]));
                    }
                }
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                for (int nn = 0; nn < NB; ++nn) {
                    int t = tS[ii][nn];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) t = imax(t, __shfl_xor(t, off, 64));
                    if (lane == 0 && i0 + ii < tk.n1 && n0 + nn < tk.n2) {
                        const int V = imax(imax(t, fb[tk.fb + i0 + ii]), ra[tk.ra + tk.rs_first + (n0 + nn) * tk.rs_stride]);
                        const int scv = V >> 16;
                        const int idx = tk.out + (i0 + ii) * tk.n2 + n0 + nn;
                        if (scv >= lo) { cell_score[idx] = scv; cell_wscore[idx] = (V & 0xffff) - JBIAS; }
                        else { cell_score[idx] = -1; cell_wscore[idx] = 0; }
                    }
                }
        }
    }
}

__global__ __launch_bounds__(WAVE) void k_joint_combine(int n_tasks, const NraJointCombineTask* __restrict__ tasks,
                                                        const NraDevRead* __restrict__ reads, NraScoreParams sp,
                                                        const int32_t* __restrict__ fsnap,
                                                        const int32_t* __restrict__ rsnap,
                                                        const int32_t* __restrict__ fb, const int32_t* __restrict__ ra,
                                                        int32_t* __restrict__ cell_score,
                                                        int32_t* __restrict__ cell_wscore,
                                                        const NraGridRow* __restrict__ rows)
{
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    NraJointCombineTask tk = tasks[task];
    if (rows) {
        // a refinement routed on the device: the read's cells are its row's n1 x n2 (unit steps); the task holds the
        // first repeat count of the R side's kept slots in rs_first
        const NraGridRow row = rows[tk.read];
        tk.n1 = row.n1; tk.n2 = row.n2;
        tk.rs_first = row.k2lo - tk.rs_first; tk.rs_stride = 1;
        if (row.n1 <= 0 || row.n2 <= 0) return;
    }
    const int Q = reads[tk.read].qlen;
    // the tile that takes all the read's k1 values in one pass, if there is one: every plane is then read once
    if (tk.n1 <= 8) joint_combine_tiles<8, 6>(tk, Q, lane, sp, fsnap, rsnap, fb, ra, cell_score, cell_wscore);
    else joint_combine_tiles<12, 6>(tk, Q, lane, sp, fsnap, rsnap, fb, ra, cell_score, cell_wscore);
}

extern "C" int nra_launch_joint_combine(hipStream_t st, int n_tasks, const NraJointCombineTask* tasks,
                                        const NraDevRead* reads, NraScoreParams sp, const int32_t* fsnap,
                                        const int32_t* rsnap, const int32_t* fb, const int32_t* ra,
                                        int32_t* cell_score, int32_t* cell_wscore, const NraGridRow* rows)
{
    if (n_tasks <= 0) return 0;
    k_joint_combine<<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, sp, fsnap, rsnap, fb, ra, cell_score, cell_wscore, rows);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// k_joint_refine_route: the routing of the reference's round 3 (nanoRepeat_joint.py:315-330) on the device, from the
// per-read results the selector of the grid before it (round 2) left there: size_a = sum_ka / n_ties (the mean of the
// tied cells, :473-474, in float64 like numpy's), and read r takes the unit-step counts k with
//     max(size_a - buf_a, lo_a[r]) <= k < min(size_a + buf_a, hi_a[r])          (lo, hi: its round-1 range)
// on both axes -- at most 2 buf_a of them; none for a read without a round-2 size.  (The global grid the reference
// walks, range(max(0, int(min size - buf)), int(max size + buf + 2)) at :298-303, holds every such k when lo >= 0.)
// Every k lies among the counts the grid's sweeps kept column states at (route_grid's keep rows); a row that does not
// is counted in words[1] (the host then fails the fetch) and gets no cells.
__global__ void k_joint_refine_route(int n_reads, const uint8_t* __restrict__ status, const int32_t* __restrict__ n_ties,
                                     const int64_t* __restrict__ sum_k1, const int64_t* __restrict__ sum_k2,
                                     const double* __restrict__ lo1, const double* __restrict__ hi1,
                                     const double* __restrict__ lo2, const double* __restrict__ hi2,
                                     int buf1, int buf2, const NraGridRow* __restrict__ keep,
                                     NraGridRow* __restrict__ rows, uint32_t* __restrict__ cell_cnt,
                                     const int32_t* __restrict__ rowspad, const NraDevRead* __restrict__ reads,
                                     const NraDevRegion* __restrict__ regions,
                                     unsigned long long* __restrict__ words)      // [0] cells, [1] bad rows, [2] executed, [3] algorithmic cells
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    NraGridRow row{0, 0, 0, 0};
    if (status[r] == 0 && n_ties[r] > 0 && rowspad[r] > 0) {
        const double nt = (double)n_ties[r];
        const double s1 = (double)sum_k1[r] / nt, s2 = (double)sum_k2[r] / nt;
        const double a1 = fmax(s1 - (double)buf1, lo1[r]), b1 = fmin(s1 + (double)buf1, hi1[r]);
        const double a2 = fmax(s2 - (double)buf2, lo2[r]), b2 = fmin(s2 + (double)buf2, hi2[r]);
        if (a1 < b1 && a2 < b2) {
            const int k1lo = imax((int)ceil(a1), 0), k1hi = (int)ceil(b1);      // first count >= a, first count >= b
            const int k2lo = imax((int)ceil(a2), 0), k2hi = (int)ceil(b2);
            if (k1hi > k1lo && k2hi > k2lo) {
                const NraGridRow k = keep[r];
                const bool inside = k1lo >= k.k1lo && k1hi <= k.k1lo + k.n1 && k2lo >= k.k2lo && k2hi <= k.k2lo + k.n2 &&
                                    k1hi - k1lo <= 2 * buf1 && k2hi - k2lo <= 2 * buf2;
                if (inside) row = NraGridRow{k1lo, k1hi - k1lo, k2lo, k2hi - k2lo};
                else atomicAdd(&words[1], 1ull);
            }
        }
    }
    rows[r] = row;
    cell_cnt[r] = (uint32_t)(row.n1 * row.n2);
    if (row.n1 > 0) {
        const NraDevRead rd = reads[r];
        const NraDevRegion rg = regions[rd.region];
        const long long n1 = row.n1, n2 = row.n2;
        const long long sk1 = n1 * row.k1lo + n1 * (n1 - 1) / 2, sk2 = n2 * row.k2lo + n2 * (n2 - 1) / 2;
        atomicAdd(&words[0], (unsigned long long)(n1 * n2));
        atomicAdd(&words[2], (unsigned long long)((long long)rowspad[r] * n1 * (1 + rg.l2)));          // the MID scans
        atomicAdd(&words[3], (unsigned long long)((long long)rd.qlen * (n1 * n2 * ((long long)rg.l1 + rg.l2 + rg.l3) +
                                                                       rg.m1 * sk1 * n2 + rg.m2 * sk2 * n1)));
    }
}

extern "C" int nra_launch_joint_refine_route(hipStream_t st, int n_reads, const uint8_t* status, const int32_t* n_ties,
                                             const int64_t* sum_k1, const int64_t* sum_k2, const double* lo1,
                                             const double* hi1, const double* lo2, const double* hi2, int buf1, int buf2,
                                             const NraGridRow* keep, NraGridRow* rows, uint32_t* cell_cnt,
                                             const int32_t* rowspad, const NraDevRead* reads, const NraDevRegion* regions,
                                             unsigned long long* words)
{
    if (n_reads <= 0) return 0;
    k_joint_refine_route<<<(n_reads + 255) / 256, 256, 0, st>>>(n_reads, status, n_ties, sum_k1, sum_k2, lo1, hi1, lo2, hi2,
                                                                 buf1, buf2, keep, rows, cell_cnt, rowspad, reads, regions, words);
    return (int)hipGetLastError();
}
#endif
#if NRA_HAS_PART(17)
extern "C" int nra_launch_joint_pk16(int R, int has_n, hipStream_t st, int n_tasks, const NraJointPairTask* tasks,
                                     const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                     const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp, int dir,
                                     int32_t* pstate, int32_t* pstate_l)
{
    if (n_tasks <= 0) return 0;
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_joint_pk16<r, true><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, dir, pstate, pstate_l); \
        else k_joint_pk16<r, false><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, dir, pstate, pstate_l);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}
#endif
