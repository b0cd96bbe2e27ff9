// nra_sweep.hip -- junction decomposition of the 1D candidate bank (gfx950).
//
// The K candidates of a read, L + unit^k + R for k in [kmin,kmax], share L + unit^k as a
// prefix and R as a suffix.  An optimal local alignment of the read against candidate k
// either lies entirely in L+unit^k, or entirely in unit^k+R, or consumes R[0].  So with
//
//   B_k = best score entirely inside L + unit^k          (forward sweep, running maximum)
//   A_k = best score entirely inside unit^k + R          (reverse sweep, running maximum)
//   S_k = best score of an alignment that consumes R[0]  (junction combine, below)
//
// Score(k) = max(S_k, B_k, A_k) is the oracle's score exactly (optimal DP decomposes at a
// node column; a gap spanning the junction is handled per gap piece by refunding one open).
// Both sweeps are ONE alignment-sized DP per read instead of K:
//
//   reverse sweep (DIR 0): reversed read vs rev(R).  At its last column every lane stores H, E_in,
//     E2_in of its rows (the "R side" of the junction: best paths starting at node (i, R[0]) --
//     closed, or inside a piece-1 / piece-2 gap); the maximum over all its cells is A = the best
//     score entirely inside R.
//   forward sweep (DIR 1): read vs L + unit^kmax.  At every unit boundary L + m*k - 1 each
//     row combines its three states with the stored R side:
//        H + Hb,   E_in + Eb_in + q,   E2_in + E2b_in + q2        -> S_k
//     and the running maximum gives B_k.
//
// The origin bit.  A_k is only needed to know whether an optimal alignment STARTS at a column >= |L|,
// and that is one bit of payload: all scores are doubled and the low bit of a state says "the best
// path into this state (largest bit among co-optimal ones) starts at column >= |L|".  An
// alignment starting at column j enters through max(H(i-1,j-1), 0): the 0 becomes 0|1 for
// j >= |L| (a flag travelling with the template column), every other operation adds even
// numbers, and max() on 2*score+bit is the lexicographic max.  So the forward sweep alone
// yields (Score, bit) of S_k and B_k, the reverse sweep never needs rev(unit)^kmax, and
//   left  flank (tstart < |L|)    passes  iff  the bit of V = max(S_k, B_k, 2A+1) is 0
//                                              (oracle: the largest tstart among co-optimal paths wins)
//   right flank (tend > |L|+m*k)  passes  if  B_k < Score;  fails if B_k == Score > S_k;
//                                 B_k == S_k == Score is ambiguous -> explicit extents DP.
//
// Two kernels run these sweeps: k_sweep_ring (reads <= 3072 bases, unit <= 8 bases: two reads per wave
// in the int16 halves of every VGPR, lane-to-lane hand-off through an LDS ring, the combine on every
// m-th step) and k_sweep_pk16 (DPP hand-off: longer units; and -- CHAIN -- reads of any length as
// chained row blocks in int32 cells, one read per wave).  E is updated lazily in both (E_in of the
// current column stays in the register, which the combine needs).
#include "nra_pk16.h"

#ifndef NRA_PART
#define NRA_PART 0
#endif
#define NRA_HAS_PART(n) (NRA_PART == 0 || NRA_PART == (n))

#define FLAG_BOUNDARY 0x00000080      // bit 7 of table byte 0
#define FLAG_SNAPSHOT 0x00008000      // bit 7 of table byte 1 (reverse sweep)
#define FLAG_INREP    0x80008000      // bits 7 of table bytes 1 and 3 (forward sweep, origin bit): column >= |L|

// A value parked in an accumulation register (AGPR).  A wave addresses 256 arch VGPRs; a forward sweep with
// R rows per lane wants 8R + ~35, so from R = 28 on something has to live in the 256 AGPRs.  The compiler's
// own choice costs a v_accvgpr move at every use of whatever it picked; the R side of the junction is read
// on every m-th step only, so it is parked there by hand.
__device__ __forceinline__ int agpr_put(int v) { int a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ int agpr_get(int a) { int v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

// selector byte of one query row for v_perm_b32: 0..3 base, 4 padding row, 5 N
template <bool HAS_N>
__device__ __forceinline__ int sweep_query_sel(const NraDevRead& rd, const uint32_t* q2bit,
                                               const uint32_t* qnmask, int gi, bool rev)
{
    if (gi >= rd.qlen) return 4;
    uint32_t b = rd.qoff + (uint32_t)(rev ? (rd.qlen - 1 - gi) : gi);
    int c = (q2bit[b >> 4] >> ((b & 15u) * 2u)) & 3u;
    if (HAS_N) {
        if ((qnmask[b >> 5] >> (b & 31u)) & 1u) c = 5;
    }
    return c;
}

// junction combine of one cell's rows at a boundary column (values biased twice).  The forward H is
// taken as it is, not as max(H, 0): a term with H < 0 is below Hb alone, an alignment inside R, and
// that is in the final maximum anyway (A, or A_k for chained reads).
template <int OFF, int N, int R, bool W = false, bool PARK = false>
__device__ __forceinline__ int sweep_combine(const int (&Hq)[R], const int (&E)[R], const int (&E2)[R],
                                             const int (&Hbo)[R], const int (&Ebo)[R],
                                             const int (&E2bo)[R], int tS)
{
    int t3_prev = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int i = OFF + n;
        const int t1 = Hq[i] + (PARK ? agpr_get(Hbo[i]) : Hbo[i]);
        const int t2 = E[i] + (PARK ? agpr_get(Ebo[i]) : Ebo[i]);
        const int t3 = E2[i] + (PARK ? agpr_get(E2bo[i]) : E2bo[i]);
        tS = mx3<W>(tS, t1, t2);
        if (n & 1) tS = mx3<W>(tS, t3_prev, t3);
        else if (n == N - 1) tS = mx2<W>(tS, t3);
        else t3_prev = t3;
    }
    return tS;
}

// R side of the junction, as the reverse sweep's registers hold it (both reads packed, biased):
// per task and row block three planes [H - o1 | E_in | E2_in] of R x 64 dwords, [row of the lane][lane],
// so that every store of the reverse sweep and every load of the forward sweep is one coalesced line.
template <int OFF, int N, int R>
__device__ __forceinline__ void sweep_snapshot(const int (&Hq)[R], const int (&E)[R], const int (&E2)[R],
                                               int32_t* __restrict__ snap_blk, int lane)
{
    int32_t* p = snap_blk + lane;
    asm volatile("" : "+v"(p));    // keeps the 3N store addresses out of the step loop's registers
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int i = OFF + n;
        p[(0 * R + i) * 64] = Hq[i];
        p[(1 * R + i) * 64] = E[i];
        p[(2 * R + i) * 64] = E2[i];
    }
}

// The same for the half-wave kernel: lane-major (NRA_SNAP_LANE_STRIDE), p = the lane's own piece, 16-byte aligned.
template <int R>
__device__ __forceinline__ void sweep_snapshot_lane(const int (&Hq)[R], const int (&E)[R], const int (&E2)[R],
                                                    int32_t* __restrict__ p)
{
    constexpr int N4 = NRA_SNAP_LANE_STRIDE(R) / 4;
    int v[4 * N4];
#pragma unroll
    for (int i = 0; i < 4 * N4; ++i) v[i] = i < R ? Hq[i] : i < 2 * R ? E[i - R] : i < 3 * R ? E2[i - 2 * R] : 0;
    int4* __restrict__ p4 = reinterpret_cast<int4*>(p);
#pragma unroll
    for (int i = 0; i < N4; ++i) p4[i] = make_int4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// CHAIN = the WIDE sweeps: reads longer than one register block (3072 bases) are swept as consecutive row
// blocks of 64*R rows by one wave, ONE read per wave in plain int32 cells (2*score + origin bit: no range
// limit worth naming), the last cell of a block leaving its per-column hand-off in a scratch strip that
// the next block's first cell picks up.  A launch has at most as many waves as there are strips and walks
// its tasks with a grid stride.
template <int R, bool HAS_N, int DIR, bool CHAIN>
__device__ __forceinline__ void sweep_dpp_task(int task, const NraSweepTask* __restrict__ tasks,
                                               const NraDevRead* __restrict__ reads,
                                               const NraDevRegion* __restrict__ regions,
                                               const uint8_t* __restrict__ pool,
                                               const uint32_t* __restrict__ q2bit,
                                               const uint32_t* __restrict__ qnmask,
                                               const NraScoreParams& sp,
                                               const int32_t* __restrict__ kmin_arr,
                                               const int32_t* __restrict__ kmax_arr,
                                               const uint32_t* __restrict__ coff,
                                               int32_t* __restrict__ snap,
                                               int32_t* __restrict__ read_a,
                                               int32_t* __restrict__ cand_score,
                                               uint8_t* __restrict__ cand_flag,
                                               int32_t* chain_buf, int chain_cap)
{
    constexpr bool W = CHAIN;             // wide cells
    constexpr int SC = 2;                 // origin-bit scheme: doubled scores, reverse sweep over rev(R) only
    constexpr int BIASW = W ? 0 : BIAS;
    const int lane = threadIdx.x;
    const NraSweepTask tk = tasks[task];
    const bool has_b = tk.read_b >= 0;
    const int ra = tk.read_a, rb = has_b ? tk.read_b : tk.read_a;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[rda.region];
    const int m = rg.m1;
    const int flank = DIR ? rg.l1 : rg.l3;
    const uint8_t* __restrict__ piece = pool + (DIR ? rg.p1_off : rg.pr_off);
    const int ncols = DIR == 0 ? flank : flank + m * tk.kmax;
    // boundary column of k = kmin (>= 0: flank >= 1); the reverse sweep has one: R[0]
    const int jfirst = DIR == 0 ? flank - 1 : flank + m * tk.kmin - 1;
    const int kmin_a = kmin_arr[ra], kmax_a = kmax_arr[ra];
    const int kmin_b = kmin_arr[rb], kmax_b = kmax_arr[rb];
    const uint32_t coff_a = coff[ra], coff_b = coff[rb];
    int32_t* __restrict__ snap_task = snap + tk.snap_off;

    const int o1 = SC * sp.open1, o2 = SC * sp.open2;
    const int P1 = W ? 1 : 0x00010001;
    const int v_floor = (BIASW - o1) * P1;                  // max(H,0) - o1 (even: the origin bit is free)
    const int v_o1 = o1 * P1, v_e1 = SC * sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = SC * sp.ext2 * P1;
    const int NEG1 = W ? -(1 << 28) : NEGB * P1;            // "minus infinity", biased once / twice
    const int NEG2 = W ? -(1 << 28) : 2 * NEGB * P1;
    // substitution scores + o1 (the diagonal is read from Hq = H - o1): all in [0, 127]
    const int s_match = SC * sp.match + o1, s_mis = o1 - SC * sp.mismatch, s_ambi = o1 - SC * sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);               // selector 4: padding row, 5: N in the read
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;

    // Score(k) and the flank verdict leave through lane 63, one boundary column at a time.  They
    // are collected in a lane-indexed register -- rotated one lane down per boundary, the newest value in
    // lane 63 -- and written 64 candidates at a time, one coalesced line per read.
    int out_a = 0, out_b = 0;
    int n_out = 0;                                          // wave-uniform: boundaries collected
    int kcur = tk.kmin;                                     // wave-uniform: k of the next boundary to leave
    auto flush = [&](int n_valid) {
        // lane l holds k = kcur - 64 + l; the oldest n_valid values sit in lanes [64 - n_valid, 64)
        const int k = kcur - 64 + lane;
        if (lane < 64 - n_valid) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && !has_b) break;
            const int lo_k = s2 ? kmin_b : kmin_a, hi_k = s2 ? kmax_b : kmax_a;
            if (k < lo_k || k > hi_k) continue;
            const uint32_t idx = (s2 ? coff_b : coff_a) + (uint32_t)(k - lo_k);
            const int v = s2 ? out_b : out_a;
            cand_score[idx] = v >> 2;                       // (score << 2) | verdict; -1: below min_dp_score
            cand_flag[idx] = (uint8_t)(v & 3);
        }
    };

    // CHAIN: row blocks of 64*R, one after the other in this wave; the strip belongs to the wave (blockIdx)
    const int n_blk = CHAIN ? (imax(rda.qlen, rdb.qlen) + 64 * R - 1) / (64 * R) : 1;
    volatile int32_t* strip = CHAIN ? chain_buf + (size_t)blockIdx.x * 10 * chain_cap : nullptr;
  for (int blk = 0; blk < n_blk; ++blk) {
    const int row_base = blk * 64 * R;
    const bool first_blk = blk == 0, last_blk = blk == n_blk - 1;
    volatile int32_t* cin = CHAIN ? strip + ((blk + 1) & 1) * 5 * chain_cap : nullptr;
    volatile int32_t* cout = CHAIN ? strip + (blk & 1) * 5 * chain_cap : nullptr;

    int qc[R];       // v_perm selectors: {selA, zero, selB, zero}
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gi = row_base + lane * R + i;
        const int ca = sweep_query_sel<HAS_N>(rda, q2bit, qnmask, gi, DIR == 0);
        const int cb = W ? 0x0c : sweep_query_sel<HAS_N>(rdb, q2bit, qnmask, gi, DIR == 0);
        qc[i] = ca | (0x0c << 8) | (cb << 16) | (0x0c << 24);
    }

    // forward sweep: the R side of the junction, row r pairs with reverse-sweep row Q-2-r.  A row without
    // a partner (the read's last row, padding rows) gets "-1": its terms stay below B_k, which holds the
    // forward H of that very cell, so they never decide anything.
    int Hbo[DIR ? R : 1], Ebo[DIR ? R : 1], E2bo[DIR ? R : 1];
    if (DIR) {
        const int q1 = SC * (sp.open1 - sp.ext1), q2 = SC * (sp.open2 - sp.ext2);     // the refunded gap opens
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = row_base + lane * R + i;
            int h[2], e[2], e2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int a = (s ? rdb.qlen : rda.qlen) - 2 - r;
                if (a >= 0) {
                    const int ablk = CHAIN ? a / (64 * R) : 0;
                    const int w = a - ablk * 64 * R;
                    const int al = w / R, ai = w - al * R;
                    const int32_t* __restrict__ p = snap_task + ((size_t)ablk * 3 * R + ai) * 64 + al;
                    const int vh = p[0], ve = p[R * 64], ve2 = p[2 * R * 64];
                    h[s] = (W ? vh : (s ? half_hi(vh) : half_lo(vh))) + 2 * o1;    // Hq carries -o1 on either side
                    e[s] = (W ? ve : (s ? half_hi(ve) : half_lo(ve))) + q1;
                    e2[s] = (W ? ve2 : (s ? half_hi(ve2) : half_lo(ve2))) + q2;
                } else { h[s] = BIASW + o1 - SC; e[s] = BIASW - SC; e2[s] = BIASW - SC; }
            }
            Hbo[i] = W ? h[0] : pack2(h[0], h[1]);
            Ebo[i] = W ? e[0] : pack2(e[0], e[1]);
            E2bo[i] = W ? e2[0] : pack2(e2[0], e2[1]);
        }
    }

    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }

    // Every lane hosts TWO virtual systolic cells -- A = its first RA rows, B = the other RB --
    // one template column apart (cell v works on column t - v, lane l hosts cells 2l and 2l+1).
    // A is fed by lane l-1's B through DPP, B by this lane's A from the previous step, so each
    // step carries two independent dependency chains per wave: the sweeps run with only a few
    // waves per SIMD and need the instruction-level parallelism.
    constexpr int RA = (R + 1) / 2, RB = R - RA;
    int HbotA = v_floor, FoutA = NEG1, F2outA = NEG1, HupA_prev = v_floor, MA = BIASW * P1;
    int HbotB = v_floor, FoutB = NEG1, F2outB = NEG1, HupB_prev = v_floor, MB = BIASW * P1;
    int accS_A = NEG2, accB_A = NEG1, accS_B = NEG2, accB_B = NEG1;
    int ttA = tbl_mis4, ttB = tbl_mis4;                   // padding column: everything mismatches
    n_out = 0; kcur = tk.kmin;

    // One flat step loop (128 cells deep; a chunk loop around a 64-step loop costs registers: the
    // compiler keeps the row arrays twice).  Every 64 steps the lanes fetch the next 64 columns.
    const int nsteps = ncols + 127;                       // cell 127 finishes the last column at step ncols + 126
    int feed = tbl_mis4;
    // what enters cell 0 at each column: constants for the first row block, else the strip
    int inH = v_floor, inF = NEG1, inF2 = NEG1, inS = NEG2, inB = NEG1;
#pragma unroll 1   // two steps per trip would turn the hand-offs into renames but cost ~14 registers
    for (int step = 0; step < nsteps; ++step) {
        {
            if ((step & 63) == 0) {
                const int col = step + lane;
                feed = tbl_mis4;
                if (col < ncols) {
                    const int code = piece[col];
                    feed = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
                    if (col >= jfirst && (col - jfirst) % m == 0) feed |= FLAG_BOUNDARY;
                    if (DIR == 0 && col == flank - 1) feed |= FLAG_SNAPSHOT;
                    if (DIR == 1 && col >= flank) feed |= FLAG_INREP;
                }
                if (CHAIN) {
                    inH = v_floor; inF = NEG1; inF2 = NEG1; inS = NEG2; inB = NEG1;
                    if (!first_blk && col < ncols) {
                        inH = cin[col]; inF = cin[chain_cap + col]; inF2 = cin[2 * chain_cap + col];
                        inS = cin[3 * chain_cap + col]; inB = cin[4 * chain_cap + col];
                    }
                }
            }
            // inputs of cell A: cell B of the lane above, as it stood after the previous step
            const int hupA = dpp_shr1(CHAIN ? inH : v_floor, HbotB);
            int FA = dpp_shr1(CHAIN ? inF : NEG1, FoutB);
            int F2A = dpp_shr1(CHAIN ? inF2 : NEG1, F2outB);
            const int ttA_new = dpp_shr1(feed, ttB);      // lane 0 takes the next template column
            feed = dpp_rol1(feed);
            const int accS_Ain = dpp_shr1(CHAIN ? inS : NEG2, accS_B);
            const int accB_Ain = dpp_shr1(CHAIN ? inB : NEG1, accB_B);
            if (CHAIN) {
                inH = dpp_rol1(inH); inF = dpp_rol1(inF); inF2 = dpp_rol1(inF2);
                inS = dpp_rol1(inS); inB = dpp_rol1(inB);
            }
            // inputs of cell B: this lane's cell A after the previous step
            const int hupB = HbotA;
            int FB = FoutA, F2B = F2outA;
            const int ttB_new = ttA;
            const int accS_Bin = accS_A, accB_Bin = accB_A;

            // the empty alignment a path may start from at this column: score 0, origin bit from the flag
            const int floorA = DIR == 1 ? (int)((((unsigned)ttA_new >> 15) & (unsigned)P1) | (unsigned)v_floor) : v_floor;
            const int floorB = DIR == 1 ? (int)((((unsigned)ttB_new >> 15) & (unsigned)P1) | (unsigned)v_floor) : v_floor;
            sweep_cell<0, RA, R, W>(Hq, Hq2, E, E2, qc, HupA_prev, FA, F2A, MA, ttA_new & 0x7f7f7f7f, tbl_hi,
                                    floorA, v_e1, v_e2, v_o1, v_o2);
            HupA_prev = hupA; HbotA = Hq[RA - 1]; FoutA = FA; F2outA = F2A;
            if (RB > 0) {
                sweep_cell<RA, RB, R, W>(Hq, Hq2, E, E2, qc, HupB_prev, FB, F2B, MB, ttB_new & 0x7f7f7f7f, tbl_hi,
                                         floorB, v_e1, v_e2, v_o1, v_o2);
                HbotB = Hq[R - 1];
            } else {
                HbotB = hupB;                             // empty cell: hand everything through
            }
            HupB_prev = hupB; FoutB = FB; F2outB = F2B;
            ttA = ttA_new; ttB = ttB_new;

            const bool atA = (ttA & FLAG_BOUNDARY) != 0, atB = (ttB & FLAG_BOUNDARY) != 0;
            const unsigned long long at_mask = __builtin_amdgcn_ballot_w64(atA || atB);
            int tSA = NEG2, tSB = NEG2;
            if constexpr (DIR != 0) {
                if (at_mask != 0) {
                    tSA = sweep_combine<0, RA, R, W>(Hq, E, E2, Hbo, Ebo, E2bo, tSA);
                    tSB = sweep_combine<RA, RB, R, W>(Hq, E, E2, Hbo, Ebo, E2bo, tSB);
                }
            }
            accS_A = mx2<W>(accS_Ain, tSA);
            accB_A = mx2<W>(accB_Ain, MA);
            accS_B = mx2<W>(accS_Bin, tSB);
            accB_B = RB > 0 ? mx2<W>(accB_Bin, MB) : accB_Bin;  // an empty cell adds nothing of its own

            if (DIR == 0) {
                int32_t* __restrict__ snap_blk = snap_task + (size_t)blk * 3 * R * 64;
                if (ttA & FLAG_SNAPSHOT) sweep_snapshot<0, RA, R>(Hq, E, E2, snap_blk, lane);
                if (ttB & FLAG_SNAPSHOT) sweep_snapshot<RA, RB, R>(Hq, E, E2, snap_blk, lane);
            }
            if (CHAIN) {
                const int col_out = step - 127;           // the column cell 127 has just finished
                if (lane == 63 && !last_blk && col_out >= 0 && col_out < ncols) {
                    cout[col_out] = HbotB; cout[chain_cap + col_out] = FoutB; cout[2 * chain_cap + col_out] = F2outB;
                    cout[3 * chain_cap + col_out] = accS_B; cout[4 * chain_cap + col_out] = accB_B;
                }
            }
            // cell 127 (lane 63, cell B) has finished a boundary column: wave-uniform branch
            if (last_blk && (__builtin_amdgcn_ballot_w64(atB) >> 63) != 0) {
                if (DIR == 0) {
                    // the one boundary of the reverse sweep: A = best alignment inside R (doubled)
                    if (lane == 63) {
                        read_a[ra] = (W ? accB_B : half_lo(accB_B)) - BIASW;
                        if (has_b) read_a[rb] = half_hi(accB_B) - BIASW;
                    }
                } else {
                    int va = 0, vb = 0;
                    if (lane == 63) {
#pragma unroll
                        for (int s2 = 0; s2 < (W ? 1 : 2); ++s2) {
                            const int B = (W ? accB_B : (s2 ? half_hi(accB_B) : half_lo(accB_B))) - BIASW;
                            const int S = (W ? accS_B : (s2 ? half_hi(accS_B) : half_lo(accS_B))) - 2 * BIASW;
                            const int lo = sp.min_score > 1 ? sp.min_score : 1;
                            // packed 2*score + origin bit; an alignment inside R starts at a column >= |L|
                            const int V = imax(imax(S, B), read_a[s2 ? rb : ra] + 1);
                            const int best = V >> 1;
                            int flag = 1;
                            if (V & 1) flag = 0;                                  // an optimal alignment starts at >= |L|
                            else if ((B >> 1) >= best) flag = ((S >> 1) >= best) ? 2 : 0;   // one ends inside L+unit^k
                            const int v = ((best >= lo ? best : -1) << 2) | flag;
                            if (s2) vb = v; else va = v;
                        }
                    }
                    out_a = dpp_rol1(out_a); out_b = dpp_rol1(out_b);
                    if (lane == 63) { out_a = va; out_b = vb; }
                    ++kcur; ++n_out;
                    if (n_out == 64) { flush(64); n_out = 0; }
                }
            }
        }
    }
    if (DIR == 1 && last_blk && n_out > 0) flush(n_out);
  }   // row blocks
}

template <int R, bool HAS_N, int DIR, bool CHAIN>
__global__ __launch_bounds__(WAVE) void k_sweep_pk16(int n_tasks, const NraSweepTask* __restrict__ tasks,
                                                     const NraDevRead* __restrict__ reads,
                                                     const NraDevRegion* __restrict__ regions,
                                                     const uint8_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ q2bit,
                                                     const uint32_t* __restrict__ qnmask,
                                                     NraScoreParams sp,
                                                     const int32_t* __restrict__ kmin_arr,
                                                     const int32_t* __restrict__ kmax_arr,
                                                     const uint32_t* __restrict__ coff,
                                                     int32_t* __restrict__ snap,
                                                     int32_t* __restrict__ read_a,
                                                     int32_t* __restrict__ cand_score,
                                                     uint8_t* __restrict__ cand_flag,
                                                     int32_t* chain_buf, int chain_cap)
{
    // unchained: one task per wave; chained: the launch has one wave per scratch strip (trip count wave-uniform)
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x)
        sweep_dpp_task<R, HAS_N, DIR, CHAIN>(task, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin_arr, kmax_arr,
                                             coff, snap, read_a, cand_score, cand_flag, chain_buf, chain_cap);
}

// ------------------------------------------------------------------------------------
// k_sweep_ring: the same sweep with the lane-to-lane hand-off through an LDS ring.
//
// Lane l owns all R rows of its row block as ONE systolic cell and works on column t - skew*l at step t.
// What a cell leaves for the cell below -- H of its last row, the two vertical-gap states and the template
// column itself (its substitution table) -- is one 16-byte LDS write into the next lane's place of slot
// (t mod skew); the lane below reads it `skew` steps later (a wave's LDS operations execute in order, and
// every lane reads its place of a slot before any lane writes it again: no barrier).  Lane 63 also writes
// lane 0's place: the constants that enter row 0 and the next template column.
//
// skew = 1 for the reverse sweep.  The forward sweep uses skew = m, the unit length: then ALL lanes sit on
// a unit boundary in the same steps -- every m-th -- and the junction combine, the only work that is not
// the DP recurrence, runs under a wave-uniform branch 1/m of the time instead of on every step (with the
// 128 columns in flight of k_sweep_pk16 some lane is always on a boundary).  The per-boundary accumulators
// S and B travel through a second, one-slot LDS array on those steps only.  Price: the pipeline is 64*m
// columns deep instead of 128.  No DPP moves, one cell per lane: ~12 instead of ~35 instructions of
// per-step overhead.  (m <= SWEEP_RING_D; other regions and chained reads use k_sweep_pk16.)
//
// Measured and not kept (round 2, config 2 at 6.2 ms per step): persistent waves taking tasks from a global
// counter (1-6 waves per SIMD: 6.3-6.6 ms; the task loop keeps ~35 more registers live); issuing the
// hand-off's LDS read one step ahead (6.5 ms); a two- or three-column skew in the reverse sweep (6.6-6.7 ms);
// two or four waves (= tasks) per workgroup (6.1-6.3 ms, SWEEP_RING_WPB below).
#define SWEEP_RING_D NRA_SWEEP_RING_MAX_M
#define NRA_Q_SPIN_LIMIT (1u << 20)         // polls (~2 us each) before a waiting second part gives up: a hang guard, not a path

#ifndef SWEEP_RING_WPB
#define SWEEP_RING_WPB 1
#endif
// QUANTA (the sweeps in quanta, k_sweep_ringq): the body runs the steps [s_begin, s_end) of the sweep only.  `load`: it
// starts from the wave state another wave left in `qs` -- the lane's registers, its places of the ring and of the boundary
// accumulators, its pending outputs (lane-major 16-byte pieces) -- instead of the initial state; `store`: it leaves that state
// instead of finishing the sweep.  Cuts are multiples of 64 steps (the column tables are reloaded every 64 / 32 steps), the
// wave-uniform counters of the forward sweep follow from the step number, and a resumed wave is the uninterrupted sweep bit
// for bit (the joint prefix / tail pair works the same way).  COMB: this part of a forward sweep meets unit boundaries --
// it needs the R side's junction rows (the reverse sweep finished) and carries the combine; a part that ends before the
// first boundary step needs neither.  Without QUANTA the arguments are constants and the body is the whole sweep.
// (piece by piece, straight between the registers and memory: an array of the whole state in between costs the
// merged kernel 60 registers)
template <int R>
__device__ __forceinline__ int qstate_get(int i, const int (&Hq)[R], const int (&Hq2)[R], const int (&E)[R], const int (&E2)[R],
                                          int Hup_prev, int M)
{
    return i < R ? Hq[i] : i < 2 * R ? Hq2[i - R] : i < 3 * R ? E[i - 2 * R] : i < 4 * R ? E2[i - 3 * R] : i == 4 * R ? Hup_prev : i == 4 * R + 1 ? M : 0;
}
template <int R>
__device__ __forceinline__ void qstate_store(int32_t* __restrict__ q, const int (&Hq)[R], const int (&Hq2)[R], const int (&E)[R],
                                             const int (&E2)[R], int Hup_prev, int M, const int4* ring, const int2* racc,
                                             int out_a, int out_b, int lane)
{
    constexpr int NR = (4 * R + 2 + 3) / 4;            // 16-byte pieces of the registers; the ring's places follow, then the outputs
    int4* __restrict__ q4 = reinterpret_cast<int4*>(q + (size_t)lane * NRA_QSTATE_INTS(R));
#pragma unroll
    for (int i = 0; i < NR; ++i)
        q4[i] = make_int4(qstate_get<R>(4 * i, Hq, Hq2, E, E2, Hup_prev, M), qstate_get<R>(4 * i + 1, Hq, Hq2, E, E2, Hup_prev, M),
                          qstate_get<R>(4 * i + 2, Hq, Hq2, E, E2, Hup_prev, M), qstate_get<R>(4 * i + 3, Hq, Hq2, E, E2, Hup_prev, M));
#pragma unroll
    for (int sl = 0; sl < SWEEP_RING_D; ++sl) q4[NR + sl] = ring[sl * 64 + lane];
    const int2 acc = racc[lane];
    q4[NR + SWEEP_RING_D] = make_int4(out_a, out_b, acc.x, acc.y);
}
template <int R>
__device__ __forceinline__ void qstate_load(const int32_t* __restrict__ q, int (&Hq)[R], int (&Hq2)[R], int (&E)[R], int (&E2)[R],
                                            int& Hup_prev, int& M, int4* ring, int2* racc, int& out_a, int& out_b, int lane)
{
    constexpr int NR = (4 * R + 2 + 3) / 4;
    const int4* __restrict__ q4 = reinterpret_cast<const int4*>(q + (size_t)lane * NRA_QSTATE_INTS(R));
#pragma unroll
    for (int sl = 0; sl < SWEEP_RING_D; ++sl) ring[sl * 64 + lane] = q4[NR + sl];
    const int4 o = q4[NR + SWEEP_RING_D];
    out_a = o.x; out_b = o.y;
    racc[lane] = make_int2(o.z, o.w);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int4 x = q4[i];
        const int v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * i + c;
            if (k < R) Hq[k] = v[c];
            else if (k < 2 * R) Hq2[k - R] = v[c];
            else if (k < 3 * R) E[k - 2 * R] = v[c];
            else if (k < 4 * R) E2[k - 3 * R] = v[c];
            else if (k == 4 * R) Hup_prev = v[c];
            else if (k == 4 * R + 1) M = v[c];
        }
    }
}
// the forward sweep's wave-uniform counters at step s (what the loop below would have counted up to there): boundary steps
// so far, the next repeat count lane LAG + 1 puts out, outputs waiting for their flush of W
struct SweepCounters { int bidx, kcur, n_out; };
__device__ __forceinline__ SweepCounters sweep_counters_at(int s, int jfirst, int m, int kmin, int kmax, int lag, int w)
{
    SweepCounters c;
    c.bidx = (jfirst >= 0 && s > jfirst) ? (s - 1 - jfirst) / m + 1 : 0;
    int outs = c.bidx - lag;
    if (outs < 0) outs = 0;
    if (outs > kmax - kmin + 1) outs = kmax - kmin + 1 > 0 ? kmax - kmin + 1 : 0;
    c.kcur = kmin + outs;
    c.n_out = outs % w;
    return c;
}

template <int R, bool HAS_N, int DIR, bool COMB, bool QUANTA>
__device__ __forceinline__ void sweep_ring_body(const int task, const int lane, int4* ring, int2* racc,
                                                const NraSweepTask* __restrict__ tasks,
                                                const NraDevRead* __restrict__ reads,
                                                const NraDevRegion* __restrict__ regions,
                                                const uint8_t* __restrict__ pool,
                                                const uint32_t* __restrict__ q2bit,
                                                const uint32_t* __restrict__ qnmask,
                                                NraScoreParams sp,
                                                const int32_t* __restrict__ kmin_arr,
                                                const int32_t* __restrict__ kmax_arr,
                                                const uint32_t* __restrict__ coff,
                                                int32_t* __restrict__ snap,
                                                int32_t* __restrict__ read_a,
                                                int32_t* __restrict__ cand_score,
                                                uint8_t* __restrict__ cand_flag,
                                                int32_t* __restrict__ qs, const int s_begin, const int s_end,
                                                const bool load, const bool store)
{
    static_assert(DIR == 1 || !COMB, "only a forward sweep meets unit boundaries");
    constexpr int SC = 2;                 // origin-bit scheme: doubled scores
    const NraSweepTask tk = tasks[task];
    const bool has_b = tk.read_b >= 0;
    const int ra = tk.read_a, rb = has_b ? tk.read_b : tk.read_a;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[rda.region];
    const int m = rg.m1;
    const int flank = DIR ? rg.l1 : rg.l3;
    const uint8_t* __restrict__ piece = pool + (DIR ? rg.p1_off : rg.pr_off);
    const int ncols = DIR ? flank + m * tk.kmax : flank;
    const int jfirst = DIR ? flank + m * tk.kmin - 1 : flank - 1;     // boundary column of k = kmin
    const int skew = DIR ? m : 1;
    const int kmin_a = kmin_arr[ra], kmax_a = kmax_arr[ra];
    const int kmin_b = kmin_arr[rb], kmax_b = kmax_arr[rb];
    const uint32_t coff_a = coff[ra], coff_b = coff[rb];
    int32_t* __restrict__ snap_task = snap + tk.snap_off;

    const int o1 = SC * sp.open1, o2 = SC * sp.open2;
    const int P1 = 0x00010001;
    const int v_floor = (BIAS - o1) * P1;
    const int v_o1 = o1 * P1, v_e1 = SC * sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = SC * sp.ext2 * P1;
    const int NEG1 = NEGB * P1, NEG2 = 2 * NEGB * P1;
    const int s_match = SC * sp.match + o1, s_mis = o1 - SC * sp.mismatch, s_ambi = o1 - SC * sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;

    // the substitution table (+ flags) of template column `col`; padding outside the template
    auto column_table = [&](int col) {
        int t = tbl_mis4;
        if (col >= 0 && col < ncols) {
            const int code = piece[col];
            t = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
            if (DIR == 0 && col == flank - 1) t |= FLAG_SNAPSHOT;
            if (DIR == 1 && col + 1 >= flank) t |= FLAG_INREP;     // origin bit of an alignment starting at the NEXT column
        }
        return t;
    };

    int out_a = 0, out_b = 0;
    int n_out = 0, kcur = tk.kmin;                          // wave-uniform
    auto flush = [&](int n_valid) {
        const int k = kcur - 64 + lane;
        if (lane < 64 - n_valid) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && !has_b) break;
            const int lo_k = s2 ? kmin_b : kmin_a, hi_k = s2 ? kmax_b : kmax_a;
            if (k < lo_k || k > hi_k) continue;
            const uint32_t idx = (s2 ? coff_b : coff_a) + (uint32_t)(k - lo_k);
            const int v = s2 ? out_b : out_a;
            cand_score[idx] = v >> 2;
            cand_flag[idx] = (uint8_t)(v & 3);
        }
    };

    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gi = lane * R + i;
        const int ca = sweep_query_sel<HAS_N>(rda, q2bit, qnmask, gi, DIR == 0);
        const int cb = sweep_query_sel<HAS_N>(rdb, q2bit, qnmask, gi, DIR == 0);
        qc[i] = ca | (0x0c << 8) | (cb << 16) | (0x0c << 24);
    }
    constexpr bool PARK = R >= 28;        // the R side of the junction in AGPRs
    int Hbo[COMB ? R : 1], Ebo[COMB ? R : 1], E2bo[COMB ? R : 1];
    if (COMB) {
        const int q1 = SC * (sp.open1 - sp.ext1), q2 = SC * (sp.open2 - sp.ext2);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = lane * R + i;
            int h[2], e[2], e2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int a = (s ? rdb.qlen : rda.qlen) - 2 - r;
                if (a >= 0) {
                    const int al = a / R, ai = a - al * R;
                    const int32_t* __restrict__ p = snap_task + (size_t)ai * 64 + al;
                    const int vh = p[0], ve = p[R * 64], ve2 = p[2 * R * 64];
                    h[s] = (s ? half_hi(vh) : half_lo(vh)) + 2 * o1;
                    e[s] = (s ? half_hi(ve) : half_lo(ve)) + q1;
                    e2[s] = (s ? half_hi(ve2) : half_lo(ve2)) + q2;
                } else { h[s] = BIAS + o1 - SC; e[s] = BIAS - SC; e2[s] = BIAS - SC; }
            }
            Hbo[i] = PARK ? agpr_put(pack2(h[0], h[1])) : pack2(h[0], h[1]);
            Ebo[i] = PARK ? agpr_put(pack2(e[0], e[1])) : pack2(e[0], e[1]);
            E2bo[i] = PARK ? agpr_put(pack2(e2[0], e2[1])) : pack2(e2[0], e2[1]);
        }
    }
    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }

    // ring: padding columns everywhere, then lane 0's first `skew` columns
#pragma unroll
    for (int s = 0; s < SWEEP_RING_D; ++s) ring[s * 64 + lane] = make_int4(v_floor, NEG1, NEG1, tbl_mis4);
    racc[lane] = make_int2(NEG2, NEG1);
    if (lane < skew) ring[lane * 64] = make_int4(v_floor, NEG1, NEG1, column_table(lane));
    ring_order();

    int Hup_prev = v_floor, M = BIAS * P1;
    int feed = tbl_mis4;
    const int nsteps = ncols + 63 * skew;                   // lane 63 finishes the last column at step ncols - 1 + 63*skew
    const int wr = (lane + 1) & 63;
    const int s0 = QUANTA ? s_begin : 0, s1 = QUANTA ? (s_end < nsteps ? s_end : nsteps) : nsteps;
    if (QUANTA && load) {
        qstate_load<R>(qs, Hq, Hq2, E, E2, Hup_prev, M, ring, racc, out_a, out_b, lane);
        ring_order();
        const SweepCounters c = sweep_counters_at(s0, jfirst, m, tk.kmin, tk.kmax, 63, 64);
        kcur = c.kcur; n_out = c.n_out;
    }
    int slot = QUANTA ? s0 % skew : 0;                      // step mod skew
    // A (best alignment inside R, doubled) of the two reads: written by the reverse sweep, constant here
    const int a_of_a = COMB ? read_a[ra] : 0, a_of_b = COMB ? read_a[rb] : 0;
    int phase = jfirst % m;                                 // boundary steps: step mod m == phase, step >= jfirst
    int pcnt = QUANTA ? s0 % m : 0;                         // step mod m
    int bidx = QUANTA ? sweep_counters_at(s0, jfirst, m, tk.kmin, tk.kmax, 63, 64).bidx : 0;      // boundary steps so far
#pragma unroll 1
    for (int step = s0; step < s1; ++step) {
        if ((step & 63) == 0) feed = column_table(step + skew + wr);      // lane 63 hands out column step + skew
        const int4 in = ring[slot * 64 + lane];
        const int tt = in.w;
        // the floor lives in F (sweep_cell, FF): score 0 with the origin bit of the next column.  What enters a
        // lane's first row is at the floor already, but for lane 0, which takes constants: one max each
        const int fl = DIR == 1 ? (int)((((unsigned)tt >> 15) & (unsigned)P1) | (unsigned)(BIAS * P1)) : BIAS * P1;
        int F = pmaxi(in.y, fl), F2 = in.z;
        sweep_cell<0, R, R, false, true>(Hq, Hq2, E, E2, qc, Hup_prev, F, F2, M, tt & 0x7f7f7f7f, tbl_hi, fl, v_e1, v_e2, v_o1, v_o2);
        Hup_prev = pmaxi(in.x, fl - v_o1);
        ring[slot * 64 + wr] = make_int4(Hq[R - 1], F, F2, tt);
        if (lane == 63) ring[slot * 64] = make_int4(v_floor, NEG1, NEG1, feed);
        ring_order();                                       // the next steps' loads stay behind these stores
        feed = dpp_rol1(feed);
        if (++slot == skew) slot = 0;

        if constexpr (DIR == 0) {
            if (tt & FLAG_SNAPSHOT) sweep_snapshot<0, R, R>(Hq, E, E2, snap_task, lane);
        } else if constexpr (COMB) {
            if (pcnt == phase && step >= jfirst) {          // every lane is on a unit boundary: wave-uniform
                const int tS = sweep_combine<0, R, R, false, PARK>(Hq, E, E2, Hbo, Ebo, E2bo, NEG2);
                const int2 acc = racc[lane];
                const int accS = pmaxi(acc.x, tS), accB = pmaxi(acc.y, M);
                racc[wr] = make_int2(accS, accB);
                if (lane == 63) racc[0] = make_int2(NEG2, NEG1);
                ring_order();
                if (bidx >= 63 && kcur <= tk.kmax) {        // lane 63 is on the boundary of k = kcur
                    int va = 0, vb = 0;
                    if (lane == 63) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int B = (s2 ? half_hi(accB) : half_lo(accB)) - BIAS;
                            const int S = (s2 ? half_hi(accS) : half_lo(accS)) - 2 * BIAS;
                            const int lo = sp.min_score > 1 ? sp.min_score : 1;
                            // packed 2*score + origin bit; an alignment inside R starts at a column >= |L|
                            const int V = imax(imax(S, B), (s2 ? a_of_b : a_of_a) + 1);
                            const int best = V >> 1;
                            int flag = 1;
                            if (V & 1) flag = 0;                                  // an optimal alignment starts at >= |L|
                            else if ((B >> 1) >= best) flag = ((S >> 1) >= best) ? 2 : 0;
                            const int v = ((best >= lo ? best : -1) << 2) | flag;
                            if (s2) vb = v; else va = v;
                        }
                    }
                    out_a = dpp_rol1(out_a); out_b = dpp_rol1(out_b);
                    if (lane == 63) { out_a = va; out_b = vb; }
                    ++kcur; ++n_out;
                    if (n_out == 64) { flush(64); n_out = 0; }
                }
                ++bidx;
            }
            if (++pcnt == m) pcnt = 0;
        }
    }
    if (QUANTA && store) {
        qstate_store<R>(qs, Hq, Hq2, E, E2, Hup_prev, M, ring, racc, out_a, out_b, lane);
    } else if (DIR == 0) {
        // the short reverse sweep ends on its one boundary, R[0]: A = best alignment inside R (doubled)
        // = the maximum over every cell of the sweep
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) M = pmaxi(M, __shfl_xor(M, off, 64));
        if (lane == 0) {
            read_a[ra] = half_lo(M) - BIAS;
            if (has_b) read_a[rb] = half_hi(M) - BIAS;
        }
    } else if (COMB && n_out > 0) {
        flush(n_out);
    }
}

template <int R, bool HAS_N, int DIR>
__global__ __launch_bounds__(WAVE * SWEEP_RING_WPB) void k_sweep_ring(int n_tasks, const NraSweepTask* __restrict__ tasks,
                                                     const NraDevRead* __restrict__ reads,
                                                     const NraDevRegion* __restrict__ regions,
                                                     const uint8_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ q2bit,
                                                     const uint32_t* __restrict__ qnmask,
                                                     NraScoreParams sp,
                                                     const int32_t* __restrict__ kmin_arr,
                                                     const int32_t* __restrict__ kmax_arr,
                                                     const uint32_t* __restrict__ coff,
                                                     int32_t* __restrict__ snap,
                                                     int32_t* __restrict__ read_a,
                                                     int32_t* __restrict__ cand_score,
                                                     uint8_t* __restrict__ cand_flag)
{
    __shared__ int4 ring_all[SWEEP_RING_WPB * SWEEP_RING_D * 64];
    __shared__ int2 racc_all[SWEEP_RING_WPB * 64];
    // (with one wave per block everything below is a constant: LDS addresses stay immediates)
    const int wave_in_block = SWEEP_RING_WPB > 1 ? (int)(threadIdx.x >> 6) : 0;
    int4* ring = ring_all + wave_in_block * SWEEP_RING_D * 64;
    int2* racc = racc_all + wave_in_block * 64;
    const int task = blockIdx.x * SWEEP_RING_WPB + wave_in_block;
    if (task >= n_tasks) return;
    const int lane = SWEEP_RING_WPB > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    sweep_ring_body<R, HAS_N, DIR, DIR == 1, false>(task, lane, ring, racc, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin_arr, kmax_arr,
                                                    coff, snap, read_a, cand_score, cand_flag, nullptr, 0, 0, false, false);
}

// ------------------------------------------------------------------------------------
// k_sweep_ring32: the LDS-ring sweep with TWO read pairs of one region per wave, 32 lanes each.
//
// Most real cores are a few hundred bases: k_sweep_ring then has 3 to 8 rows per lane, and what does not
// depend on the rows -- the ~12 instructions of per-step overhead, the 63*skew columns of pipeline fill --
// is a quarter of the work.  Here a pair occupies half a wave (rows = 32 * R): the two halves run the same
// template (same region, the union of the four reads' windows) in lock step, each with its own ring
// (lane 31 hands out what enters lane 0, lane 63 what enters lane 32), so the overhead is paid once for four
// reads and the pipeline is 32*skew columns deep.
template <int R, bool HAS_N, int DIR, bool COMB, bool QUANTA>          // COMB, QUANTA and the last four arguments as in sweep_ring_body
__device__ __forceinline__ void sweep_ring32_body(const int task, const int lane, int4* ring, int2* racc,
                                                  const NraSweepTask* __restrict__ tasks,
                                                  const NraDevRead* __restrict__ reads,
                                                  const NraDevRegion* __restrict__ regions,
                                                  const uint8_t* __restrict__ pool,
                                                  const uint32_t* __restrict__ q2bit,
                                                  const uint32_t* __restrict__ qnmask,
                                                  NraScoreParams sp,
                                                  const int32_t* __restrict__ kmin_arr,
                                                  const int32_t* __restrict__ kmax_arr,
                                                  const uint32_t* __restrict__ coff,
                                                  int32_t* __restrict__ snap,
                                                  int32_t* __restrict__ read_a,
                                                  int32_t* __restrict__ cand_score,
                                                  uint8_t* __restrict__ cand_flag,
                                                  int32_t* __restrict__ qs, const int s_begin, const int s_end,
                                                  const bool load, const bool store)
{
    static_assert(DIR == 1 || !COMB, "only a forward sweep meets unit boundaries");
    constexpr int SC = 2;
    const int hoff = lane & 32;                           // first lane of this lane's half
    const int hl = lane & 31;                             // lane within the half
    const NraSweepTask tk = tasks[task];
    // the half's own pair (a half without reads repeats pair a/b and stores nothing)
    const bool half_on = hoff == 0 || tk.read_c >= 0;
    const int r0 = (hoff && tk.read_c >= 0) ? tk.read_c : tk.read_a;
    const int r1x = (hoff && tk.read_c >= 0) ? tk.read_d : tk.read_b;
    const bool has_b = r1x >= 0;
    const int ra = r0, rb = has_b ? r1x : r0;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[reads[tk.read_a].region];
    const int m = rg.m1;
    const int flank = DIR ? rg.l1 : rg.l3;
    const uint8_t* __restrict__ piece = pool + (DIR ? rg.p1_off : rg.pr_off);
    const int ncols = DIR ? flank + m * tk.kmax : flank;
    const int jfirst = DIR ? flank + m * tk.kmin - 1 : flank - 1;
    const int skew = DIR ? m : 1;
    const int kmin_a = kmin_arr[ra], kmax_a = kmax_arr[ra];
    const int kmin_b = kmin_arr[rb], kmax_b = kmax_arr[rb];
    const uint32_t coff_a = coff[ra], coff_b = coff[rb];
    int32_t* __restrict__ snap_task = snap + tk.snap_off;

    const int o1 = SC * sp.open1, o2 = SC * sp.open2;
    const int P1 = 0x00010001;
    const int v_floor = (BIAS - o1) * P1;
    const int v_o1 = o1 * P1, v_e1 = SC * sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = SC * sp.ext2 * P1;
    const int NEG1 = NEGB * P1, NEG2 = 2 * NEGB * P1;
    const int s_match = SC * sp.match + o1, s_mis = o1 - SC * sp.mismatch, s_ambi = o1 - SC * sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;

    auto column_table = [&](int col) {
        int t = tbl_mis4;
        if (col >= 0 && col < ncols) {
            const int code = piece[col];
            t = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
            if (DIR == 0 && col == flank - 1) t |= FLAG_SNAPSHOT;
            if (DIR == 1 && col + 1 >= flank) t |= FLAG_INREP;     // origin bit of an alignment starting at the NEXT column
        }
        return t;
    };

    // outputs: lane 31 / 63 produce one value per boundary; each half keeps its last 32 in a lane-indexed
    // register (rotated within the half through the LDS crossbar) and writes them 32 candidates at a time
    int out_a = 0, out_b = 0;
    int n_out = 0, kcur = tk.kmin;                          // wave-uniform
    const int rot_src = (hoff | ((hl + 1) & 31)) << 2;      // ds_bpermute address: the next lane of the half
    auto flush = [&](int n_valid) {
        const int k = kcur - 32 + hl;
        if (hl < 32 - n_valid || !half_on) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && !has_b) break;
            const int lo_k = s2 ? kmin_b : kmin_a, hi_k = s2 ? kmax_b : kmax_a;
            if (k < lo_k || k > hi_k) continue;
            const uint32_t idx = (s2 ? coff_b : coff_a) + (uint32_t)(k - lo_k);
            const int v = s2 ? out_b : out_a;
            cand_score[idx] = v >> 2;
            cand_flag[idx] = (uint8_t)(v & 3);
        }
    };

    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gi = hl * R + i;
        const int ca = sweep_query_sel<HAS_N>(rda, q2bit, qnmask, gi, DIR == 0);
        const int cb = sweep_query_sel<HAS_N>(rdb, q2bit, qnmask, gi, DIR == 0);
        qc[i] = ca | (0x0c << 8) | (cb << 16) | (0x0c << 24);
    }
    int Hbo[COMB ? R : 1], Ebo[COMB ? R : 1], E2bo[COMB ? R : 1];
    if (COMB) {
        const int q1 = SC * (sp.open1 - sp.ext1), q2 = SC * (sp.open2 - sp.ext2);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = hl * R + i;
            int h[2], e[2], e2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int a = (s ? rdb.qlen : rda.qlen) - 2 - r;
                if (a >= 0) {
                    const int al = a / R, ai = a - al * R;      // reverse-sweep row a sits in lane al of this half
                    const int32_t* __restrict__ p = snap_task + (size_t)(hoff + al) * NRA_SNAP_LANE_STRIDE(R) + ai;
                    const int vh = p[0], ve = p[R], ve2 = p[2 * R];
                    h[s] = (s ? half_hi(vh) : half_lo(vh)) + 2 * o1;
                    e[s] = (s ? half_hi(ve) : half_lo(ve)) + q1;
                    e2[s] = (s ? half_hi(ve2) : half_lo(ve2)) + q2;
                } else { h[s] = BIAS + o1 - SC; e[s] = BIAS - SC; e2[s] = BIAS - SC; }
            }
            Hbo[i] = pack2(h[0], h[1]);
            Ebo[i] = pack2(e[0], e[1]);
            E2bo[i] = pack2(e2[0], e2[1]);
        }
    }
    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }

#pragma unroll
    for (int s = 0; s < SWEEP_RING_D; ++s) ring[s * 64 + lane] = make_int4(v_floor, NEG1, NEG1, tbl_mis4);
    racc[lane] = make_int2(NEG2, NEG1);
    if (hl < skew) ring[hl * 64 + hoff] = make_int4(v_floor, NEG1, NEG1, column_table(hl));
    ring_order();

    int Hup_prev = v_floor, M = BIAS * P1;
    int feed = tbl_mis4;
    const int nsteps = ncols + 31 * skew;                   // lanes 31 / 63 finish the last column at step ncols - 1 + 31*skew
    const int wr = hoff | ((hl + 1) & 31);
    const int s0 = QUANTA ? s_begin : 0, s1 = QUANTA ? (s_end < nsteps ? s_end : nsteps) : nsteps;
    if (QUANTA && load) {
        qstate_load<R>(qs, Hq, Hq2, E, E2, Hup_prev, M, ring, racc, out_a, out_b, lane);
        ring_order();
        const SweepCounters c = sweep_counters_at(s0, jfirst, m, tk.kmin, tk.kmax, 31, 32);
        kcur = c.kcur; n_out = c.n_out;
    }
    int slot = QUANTA ? s0 % skew : 0;
    const int a_of_a = COMB ? read_a[ra] : 0, a_of_b = COMB ? read_a[rb] : 0;
    int phase = jfirst % m;
    int pcnt = QUANTA ? s0 % m : 0;
    int bidx = QUANTA ? sweep_counters_at(s0, jfirst, m, tk.kmin, tk.kmax, 31, 32).bidx : 0;
#pragma unroll 1
    for (int step = s0; step < s1; ++step) {
        // both halves sweep the same template: the column tables repeat with period 32 across the wave, and a
        // full-wave rotation keeps them so
        if ((step & 31) == 0) feed = column_table(step + skew + ((hl + 1) & 31));
        const int4 in = ring[slot * 64 + lane];
        const int tt = in.w;
        const int fl = DIR == 1 ? (int)((((unsigned)tt >> 15) & (unsigned)P1) | (unsigned)(BIAS * P1)) : BIAS * P1;   // as in k_sweep_ring
        int F = pmaxi(in.y, fl), F2 = in.z;
        sweep_cell<0, R, R, false, true>(Hq, Hq2, E, E2, qc, Hup_prev, F, F2, M, tt & 0x7f7f7f7f, tbl_hi, fl, v_e1, v_e2, v_o1, v_o2);
        Hup_prev = pmaxi(in.x, fl - v_o1);
        ring[slot * 64 + wr] = make_int4(Hq[R - 1], F, F2, tt);
        if (hl == 31) ring[slot * 64 + hoff] = make_int4(v_floor, NEG1, NEG1, feed);
        ring_order();                                       // the next steps' loads stay behind these stores
        feed = dpp_rol1(feed);
        if (++slot == skew) slot = 0;

        if constexpr (DIR == 0) {
            if (tt & FLAG_SNAPSHOT) sweep_snapshot_lane<R>(Hq, E, E2, snap_task + (size_t)lane * NRA_SNAP_LANE_STRIDE(R));
        } else if constexpr (COMB) {
            if (pcnt == phase && step >= jfirst) {          // every lane is on a unit boundary: wave-uniform
                const int tS = sweep_combine<0, R, R>(Hq, E, E2, Hbo, Ebo, E2bo, NEG2);
                const int2 acc = racc[lane];
                const int accS = pmaxi(acc.x, tS), accB = pmaxi(acc.y, M);
                racc[wr] = make_int2(accS, accB);
                if (hl == 31) racc[hoff] = make_int2(NEG2, NEG1);
                ring_order();
                if (bidx >= 31 && kcur <= tk.kmax) {        // lanes 31 and 63 are on the boundary of k = kcur
                    int va = 0, vb = 0;
                    if (hl == 31) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const int B = (s2 ? half_hi(accB) : half_lo(accB)) - BIAS;
                            const int S = (s2 ? half_hi(accS) : half_lo(accS)) - 2 * BIAS;
                            const int lo = sp.min_score > 1 ? sp.min_score : 1;
                            const int V = imax(imax(S, B), (s2 ? a_of_b : a_of_a) + 1);
                            const int best = V >> 1;
                            int flag = 1;
                            if (V & 1) flag = 0;
                            else if ((B >> 1) >= best) flag = ((S >> 1) >= best) ? 2 : 0;
                            const int v = ((best >= lo ? best : -1) << 2) | flag;
                            if (s2) vb = v; else va = v;
                        }
                    }
                    out_a = __builtin_amdgcn_ds_bpermute(rot_src, out_a);
                    out_b = __builtin_amdgcn_ds_bpermute(rot_src, out_b);
                    if (hl == 31) { out_a = va; out_b = vb; }
                    ++kcur; ++n_out;
                    if (n_out == 32) { flush(32); n_out = 0; }
                }
                ++bidx;
            }
            if (++pcnt == m) pcnt = 0;
        }
    }
    if (QUANTA && store) {
        qstate_store<R>(qs, Hq, Hq2, E, E2, Hup_prev, M, ring, racc, out_a, out_b, lane);
    } else if (DIR == 0) {
        // A = best alignment inside R (doubled) = the maximum over every cell of the half's sweep
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) M = pmaxi(M, __shfl_xor(M, off, 64));
        if (hl == 0 && half_on) {
            read_a[ra] = half_lo(M) - BIAS;
            if (has_b) read_a[rb] = half_hi(M) - BIAS;
        }
    } else if (COMB && n_out > 0) {
        flush(n_out);
    }
}

template <int R, bool HAS_N, int DIR>
__global__ __launch_bounds__(WAVE) void k_sweep_ring32(int n_tasks, const NraSweepTask* __restrict__ tasks,
                                                       const NraDevRead* __restrict__ reads,
                                                       const NraDevRegion* __restrict__ regions,
                                                       const uint8_t* __restrict__ pool,
                                                       const uint32_t* __restrict__ q2bit,
                                                       const uint32_t* __restrict__ qnmask,
                                                       NraScoreParams sp,
                                                       const int32_t* __restrict__ kmin_arr,
                                                       const int32_t* __restrict__ kmax_arr,
                                                       const uint32_t* __restrict__ coff,
                                                       int32_t* __restrict__ snap,
                                                       int32_t* __restrict__ read_a,
                                                       int32_t* __restrict__ cand_score,
                                                       uint8_t* __restrict__ cand_flag)
{
    __shared__ int4 ring[SWEEP_RING_D * 64];
    __shared__ int2 racc[64];
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    sweep_ring32_body<R, HAS_N, DIR, DIR == 1, false>(task, (int)threadIdx.x, ring, racc, tasks, reads, regions, pool, q2bit, qnmask, sp,
                                                      kmin_arr, kmax_arr, coff, snap, read_a, cand_score, cand_flag, nullptr, 0, 0, false, false);
}

// ------------------------------------------------------------------------------------
// k_sweep_ringq: a bucket's reverse and forward sweeps as ONE launch of quanta taken by ticket.
//
// Two launches per bucket (all reverse sweeps, then all forward sweeps) made kernel-length tasks: 3750 waves of 1000 - 2300
// steps on 1024 SIMDs ended with a tail of SIMDs holding 3 or 4 of them (12 - 14 % of a config-2 step), and no forward
// sweep could start before the last reverse sweep of its bucket was over.  Here a sweep is cut every `qsteps` steps (a
// multiple of 64; the host picks it; a forward sweep also at the last multiple of 64 at or before its first boundary
// step) into parts -- quanta of a few hundred steps -- and a wave takes ONE by ticket (an
// atomic add when it starts) from a list the host orders [reverse parts 0 of every task | reverse parts 1 | ... | forward
// parts 0 | forward parts 1 | ...]:
//   * part p of a sweep waits for part p - 1 (it resumes from the state that one left), and a forward part that reaches
//     the first boundary step also for the task's whole reverse sweep (the junction rows, A): two counters per task count
//     the finished parts of either direction; bounded spin with s_sleep, a launch-wide give-up word like k_sweep_ringmt's.
//     Every producer holds a smaller ticket, so it was taken by a wave that has started: running or done, whatever order
//     the workgroups are dispatched in -- no deadlock as long as started waves stay resident;
//   * producers publish with an agent-scope release fence before the arrival, the consumer acquires after the poll
//     (cdna_hip_programming.md Guideline 16: plain payload, atomic flag, fences on both sides);
//   * a cut costs no step: the dumped state is the skewed wave state, the pipeline is not drained.
// A forward part that ends before the first boundary step (the columns of L) runs the plain body: no junction input, no combine.
// (as real function calls -- noinline -- the bodies cost the calling convention's register reserve: 248 at R = 15)
template <int R, bool HAS_N, bool HALF, int DIR, bool COMB>
__device__ __forceinline__ void sweep_quantum(const int task, const int lane, int4* ring, int2* racc,
                                                        const NraSweepTask* __restrict__ tasks,
                                                        const NraDevRead* __restrict__ reads,
                                                        const NraDevRegion* __restrict__ regions,
                                                        const uint8_t* __restrict__ pool,
                                                        const uint32_t* __restrict__ q2bit,
                                                        const uint32_t* __restrict__ qnmask,
                                                        NraScoreParams sp,
                                                        const int32_t* __restrict__ kmin_arr,
                                                        const int32_t* __restrict__ kmax_arr,
                                                        const uint32_t* __restrict__ coff,
                                                        int32_t* __restrict__ snap,
                                                        int32_t* __restrict__ read_a,
                                                        int32_t* __restrict__ cand_score,
                                                        uint8_t* __restrict__ cand_flag,
                                                        int32_t* __restrict__ qs, const int s_begin, const int s_end,
                                                        const bool load, const bool store)
{
    if (HALF) sweep_ring32_body<R, HAS_N, DIR, COMB, true>(task, lane, ring, racc, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin_arr,
                                                           kmax_arr, coff, snap, read_a, cand_score, cand_flag, qs, s_begin, s_end, load, store);
    else sweep_ring_body<R, HAS_N, DIR, COMB, true>(task, lane, ring, racc, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin_arr, kmax_arr,
                                                    coff, snap, read_a, cand_score, cand_flag, qs, s_begin, s_end, load, store);
}

// (waves per SIMD as the forward sweep alone: without the hint the merged body takes 226 registers at R = 15 where
// k_sweep_ring's forward sweep takes 156; with it 168 and a few spilled values in the prologues -- the R side's
// constants while the state is loaded --, none in a step loop)
constexpr int ringq_waves(int R)
{
    const int w = 512 / (9 * R + 30);
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

template <int R, bool HAS_N, bool HALF>
__global__ __launch_bounds__(WAVE, ringq_waves(R)) void k_sweep_ringq(int n_quanta, const uint32_t* __restrict__ qlist, int qsteps, int n_tasks,
                                                      int32_t* ticket, int32_t* arrivals, int32_t* giveup,
                                                      int32_t* __restrict__ qstate,
                                                      const NraSweepTask* __restrict__ tasks,
                                                      const NraDevRead* __restrict__ reads,
                                                      const NraDevRegion* __restrict__ regions,
                                                      const uint8_t* __restrict__ pool,
                                                      const uint32_t* __restrict__ q2bit,
                                                      const uint32_t* __restrict__ qnmask,
                                                      NraScoreParams sp,
                                                      const int32_t* __restrict__ kmin_arr,
                                                      const int32_t* __restrict__ kmax_arr,
                                                      const uint32_t* __restrict__ coff,
                                                      int32_t* __restrict__ snap,
                                                      int32_t* __restrict__ read_a,
                                                      int32_t* __restrict__ cand_score,
                                                      uint8_t* __restrict__ cand_flag)
{
    __shared__ int4 ring[SWEEP_RING_D * 64];
    __shared__ int2 racc[64];
    const int lane = threadIdx.x;
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= n_quanta) return;
    // a quantum: direction << 31 | part << NRA_Q_PART_SHIFT | task (NRA_Q_* in nra_internal.h, with the step counts the host
    // lists the parts by)
    const uint32_t q = qlist[t];
    const int dir = (int)(q >> 31), part = (int)((q >> NRA_Q_PART_SHIFT) & NRA_Q_PART_MASK), task = (int)(q & NRA_Q_TASK_MASK);
    const NraSweepTask tk = tasks[task];
    const NraDevRegion rg = regions[reads[tk.read_a].region];
    const int nsteps_rev = NRA_Q_STEPS_REV(rg.l3, HALF), nsteps_fwd = NRA_Q_STEPS_FWD(rg.l1, rg.m1, tk.kmax, HALF);
    const int jfirst = rg.l1 + rg.m1 * tk.kmin - 1;                 // the first boundary step of the forward sweep
    // a forward sweep is cut at NRA_Q_CUT(jfirst) first -- the last multiple of 64 steps at or before the first boundary
    // step: the parts before it are plain -- and every `qsteps` steps on either side of that cut; a reverse sweep every `qsteps`
    const int s_cut = dir ? NRA_Q_CUT(jfirst) : 0;
    const int n_plain = (s_cut + qsteps - 1) / qsteps;
    const bool comb = dir && part >= n_plain;
    const int s_begin = comb ? s_cut + (part - n_plain) * qsteps : part * qsteps;
    const int s_end = (!comb && dir && s_begin + qsteps > s_cut) ? s_cut : s_begin + qsteps;
    const bool load = part > 0, store = s_end < (dir ? nsteps_fwd : nsteps_rev);
    const int need_rev = comb ? (nsteps_rev + qsteps - 1) / qsteps : 0;
    int32_t* mine = arrivals + 2 * (size_t)task + dir;
    const int32_t* rev = arrivals + 2 * (size_t)task;
    if (load || comb) {
        for (unsigned spins = 0;; ++spins) {
            int ok = 0;
            if (lane == 0)
                ok = (!load || __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= part) &&
                     (!comb || __hip_atomic_load(rev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need_rev);
            if (__builtin_amdgcn_readfirstlane(ok)) break;
            __builtin_amdgcn_s_sleep(64);
            if ((spins & 15) == 15) {
                int failed = 0;
                if (lane == 0) failed = __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(failed) != 0) return;
                if (spins >= NRA_Q_SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(giveup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // a slot per forward sweep, then -- where a reverse sweep has more than one part -- one per reverse sweep
    int32_t* __restrict__ qs = qstate + ((size_t)(dir ? 0 : n_tasks) + (size_t)task) * (NRA_QSTATE_INTS(R) * 64);
#define NRA_Q_ARGS task, lane, ring, racc, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin_arr, kmax_arr, coff, snap, read_a, cand_score, cand_flag, qs, s_begin, s_end, load, store
    if (!dir) sweep_quantum<R, HAS_N, HALF, 0, false>(NRA_Q_ARGS);
    else if (!comb) sweep_quantum<R, HAS_N, HALF, 1, false>(NRA_Q_ARGS);
    else sweep_quantum<R, HAS_N, HALF, 1, true>(NRA_Q_ARGS);
#undef NRA_Q_ARGS
    // everything this wave stored -- the wave state at the cut, or the R side's snapshot and A -- before the arrival
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------
// k_sweep_ringchain: the LDS-ring sweep for reads of more than NRA_RING_CHAIN_MIN_ROWS rows.
//
// k_sweep_ring keeps all rows of a read pair in the registers of one wave: from 28 rows per lane on that is one
// wave per SIMD.  Here the rows are swept as consecutive blocks of 64*R rows (R = NRA_RING_CHAIN_R) by the same
// wave, which then fits three to a SIMD: what lane 63 leaves per column -- H of the block's last row, the two
// vertical-gap states and, on boundary columns, the S and B accumulators -- goes to a scratch strip that the
// next block's lane 0 picks up (lane 63 feeds lane 0's ring slot anyway: it hands out the strip's values
// together with the template column).  W = false: two reads per wave in packed int16 (doubled scores up to
// 27000); W = true: one read per wave in int32 cells, any length.  A launch has one wave per strip and walks its
// tasks with a wave-uniform grid stride.
template <int R, bool HAS_N, int DIR, bool W>
__global__ __launch_bounds__(WAVE) void k_sweep_ringchain(int n_tasks, const NraSweepTask* __restrict__ tasks,
                                                          const NraDevRead* __restrict__ reads,
                                                          const NraDevRegion* __restrict__ regions,
                                                          const uint8_t* __restrict__ pool,
                                                          const uint32_t* __restrict__ q2bit,
                                                          const uint32_t* __restrict__ qnmask,
                                                          NraScoreParams sp,
                                                          const int32_t* __restrict__ kmin_arr,
                                                          const int32_t* __restrict__ kmax_arr,
                                                          const uint32_t* __restrict__ coff,
                                                          int32_t* __restrict__ snap,
                                                          int32_t* __restrict__ read_a,
                                                          int32_t* __restrict__ cand_score,
                                                          uint8_t* __restrict__ cand_flag,
                                                          int32_t* chain_buf, int chain_cap)
{
    constexpr int SC = 2;
    constexpr int BIASW = W ? 0 : BIAS;
    __shared__ int4 ring[SWEEP_RING_D * 64];
    __shared__ int2 racc[64];
    const int lane = threadIdx.x;
    const int wr = (lane + 1) & 63;
    volatile int32_t* strip = chain_buf + (size_t)blockIdx.x * 10 * chain_cap;
  for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const NraSweepTask tk = tasks[task];
    const bool has_b = !W && tk.read_b >= 0;
    const int ra = tk.read_a, rb = has_b ? tk.read_b : tk.read_a;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[rda.region];
    const int m = rg.m1;
    const int flank = DIR ? rg.l1 : rg.l3;
    const uint8_t* __restrict__ piece = pool + (DIR ? rg.p1_off : rg.pr_off);
    const int ncols = DIR ? flank + m * tk.kmax : flank;
    const int jfirst = DIR ? flank + m * tk.kmin - 1 : flank - 1;
    const int skew = DIR ? m : 1;
    const int kmin_a = kmin_arr[ra], kmax_a = kmax_arr[ra];
    const int kmin_b = kmin_arr[rb], kmax_b = kmax_arr[rb];
    const uint32_t coff_a = coff[ra], coff_b = coff[rb];
    int32_t* __restrict__ snap_task = snap + tk.snap_off;

    const int o1 = SC * sp.open1, o2 = SC * sp.open2;
    const int P1 = W ? 1 : 0x00010001;
    const int v_floor = (BIASW - o1) * P1;
    const int v_o1 = o1 * P1, v_e1 = SC * sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = SC * sp.ext2 * P1;
    const int NEG1 = W ? -(1 << 28) : NEGB * P1;
    const int NEG2 = W ? -(1 << 28) : 2 * NEGB * P1;
    const int s_match = SC * sp.match + o1, s_mis = o1 - SC * sp.mismatch, s_ambi = o1 - SC * sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;

    auto column_table = [&](int col) {
        int t = tbl_mis4;
        if (col >= 0 && col < ncols) {
            const int code = piece[col];
            t = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
            if (DIR == 0 && col == flank - 1) t |= FLAG_SNAPSHOT;
            if (DIR == 1 && col + 1 >= flank) t |= FLAG_INREP;     // origin bit of an alignment starting at the NEXT column
        }
        return t;
    };

    int out_a = 0, out_b = 0;
    int n_out = 0, kcur = tk.kmin;                          // wave-uniform
    auto flush = [&](int n_valid) {
        const int k = kcur - 64 + lane;
        if (lane < 64 - n_valid) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && !has_b) break;
            const int lo_k = s2 ? kmin_b : kmin_a, hi_k = s2 ? kmax_b : kmax_a;
            if (k < lo_k || k > hi_k) continue;
            const uint32_t idx = (s2 ? coff_b : coff_a) + (uint32_t)(k - lo_k);
            const int v = s2 ? out_b : out_a;
            cand_score[idx] = v >> 2;
            cand_flag[idx] = (uint8_t)(v & 3);
        }
    };
    const int a_of_a = DIR ? read_a[ra] : 0, a_of_b = DIR ? read_a[rb] : 0;
    int a_all = BIASW * P1;                                  // reverse sweep: maximum over the cells of every block

    const int n_blk = (imax(rda.qlen, rdb.qlen) + 64 * R - 1) / (64 * R);
    const int nsteps = ncols + 63 * skew;
  for (int blk = 0; blk < n_blk; ++blk) {
    const int row_base = blk * 64 * R;
    const bool first_blk = blk == 0, last_blk = blk == n_blk - 1;
    volatile int32_t* cin = strip + ((blk + 1) & 1) * 5 * chain_cap;
    volatile int32_t* cout = strip + (blk & 1) * 5 * chain_cap;
    // what enters row 0 of this block at template column `col`: constants for the first block, else the strip
    auto strip_in = [&](int col, int which, int dflt) {
        return (!first_blk && col >= 0 && col < ncols) ? (int)cin[which * chain_cap + col] : dflt;
    };

    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gi = row_base + lane * R + i;
        const int ca = sweep_query_sel<HAS_N>(rda, q2bit, qnmask, gi, DIR == 0);
        const int cb = W ? 0x0c : sweep_query_sel<HAS_N>(rdb, q2bit, qnmask, gi, DIR == 0);
        qc[i] = ca | (0x0c << 8) | (cb << 16) | (0x0c << 24);
    }
    int Hbo[DIR ? R : 1], Ebo[DIR ? R : 1], E2bo[DIR ? R : 1];
    if (DIR) {
        const int q1 = SC * (sp.open1 - sp.ext1), q2 = SC * (sp.open2 - sp.ext2);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = row_base + lane * R + i;
            int h[2], e[2], e2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int a = (s ? rdb.qlen : rda.qlen) - 2 - r;
                if (a >= 0) {
                    const int ablk = a / (64 * R);
                    const int w = a - ablk * 64 * R;
                    const int al = w / R, ai = w - al * R;
                    const int32_t* __restrict__ p = snap_task + ((size_t)ablk * 3 * R + ai) * 64 + al;
                    const int vh = p[0], ve = p[R * 64], ve2 = p[2 * R * 64];
                    h[s] = (W ? vh : (s ? half_hi(vh) : half_lo(vh))) + 2 * o1;
                    e[s] = (W ? ve : (s ? half_hi(ve) : half_lo(ve))) + q1;
                    e2[s] = (W ? ve2 : (s ? half_hi(ve2) : half_lo(ve2))) + q2;
                } else { h[s] = BIASW + o1 - SC; e[s] = BIASW - SC; e2[s] = BIASW - SC; }
            }
            Hbo[i] = W ? h[0] : pack2(h[0], h[1]);
            Ebo[i] = W ? e[0] : pack2(e[0], e[1]);
            E2bo[i] = W ? e2[0] : pack2(e2[0], e2[1]);
        }
    }
    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }

    // ring: padding columns everywhere, then lane 0's first `skew` columns and first boundary
#pragma unroll
    for (int s = 0; s < SWEEP_RING_D; ++s) ring[s * 64 + lane] = make_int4(v_floor, NEG1, NEG1, tbl_mis4);
    racc[lane] = make_int2(NEG2, NEG1);
    if (lane < skew)
        ring[lane * 64] = make_int4(strip_in(lane, 0, v_floor), strip_in(lane, 1, NEG1), strip_in(lane, 2, NEG1), column_table(lane));
    if (DIR == 1 && lane == 0) racc[0] = make_int2(strip_in(jfirst, 3, NEG2), strip_in(jfirst, 4, NEG1));
    ring_order();

    int Hup_prev = v_floor, M = BIASW * P1;
    int feed = tbl_mis4, sH = v_floor, sF = NEG1, sF2 = NEG1, sS = NEG2, sB = NEG1;   // what lane 63 hands to lane 0
    int slot = 0;
    int phase = jfirst % m, pcnt = 0, bidx = 0;
    n_out = 0; kcur = tk.kmin;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        if ((step & 63) == 0) {                                            // columns step + skew + (0..63)
            const int col = step + skew + wr;
            feed = column_table(col);
            sH = strip_in(col, 0, v_floor); sF = strip_in(col, 1, NEG1); sF2 = strip_in(col, 2, NEG1);
            if (DIR == 1) { sS = strip_in(col, 3, NEG2); sB = strip_in(col, 4, NEG1); }
        }
        const int4 in = ring[slot * 64 + lane];
        const int tt = in.w;
        const int fl = DIR == 1 ? (int)((((unsigned)tt >> 15) & (unsigned)P1) | (unsigned)(BIASW * P1)) : BIASW * P1;   // as in k_sweep_ring
        int F = mx2<W>(in.y, fl), F2 = in.z;
        sweep_cell<0, R, R, W, true>(Hq, Hq2, E, E2, qc, Hup_prev, F, F2, M, tt & 0x7f7f7f7f, tbl_hi, fl, v_e1, v_e2, v_o1, v_o2);
        Hup_prev = mx2<W>(in.x, fl - v_o1);
        ring[slot * 64 + wr] = make_int4(Hq[R - 1], F, F2, tt);
        if (lane == 63) ring[slot * 64] = make_int4(sH, sF, sF2, feed);
        ring_order();                                       // the next steps' loads stay behind these stores
        const int c63 = step - 63 * skew;                                  // the column lane 63 has just finished
        if (lane == 63 && !last_blk && c63 >= 0) {
            cout[c63] = Hq[R - 1]; cout[chain_cap + c63] = F; cout[2 * chain_cap + c63] = F2;
        }
        feed = dpp_rol1(feed); sH = dpp_rol1(sH); sF = dpp_rol1(sF); sF2 = dpp_rol1(sF2);
        if (++slot == skew) slot = 0;

        if constexpr (DIR == 0) {
            if (tt & FLAG_SNAPSHOT) sweep_snapshot<0, R, R>(Hq, E, E2, snap_task + (size_t)blk * 3 * R * 64, lane);
        } else {
            const bool boundary = pcnt == phase && step >= jfirst;         // wave-uniform: every lane on a unit boundary
            if (boundary) {
                const int tS = sweep_combine<0, R, R, W>(Hq, E, E2, Hbo, Ebo, E2bo, NEG2);
                const int2 acc = racc[lane];
                const int accS = mx2<W>(acc.x, tS), accB = mx2<W>(acc.y, M);
                racc[wr] = make_int2(accS, accB);
                if (lane == 63) {
                    racc[0] = make_int2(sS, sB);                           // of column step + skew: lane 0's next boundary
                    if (!last_blk && c63 >= 0) { cout[3 * chain_cap + c63] = accS; cout[4 * chain_cap + c63] = accB; }
                }
                ring_order();
                if (last_blk && bidx >= 63 && kcur <= tk.kmax) {           // lane 63 is on the boundary of k = kcur
                    int va = 0, vb = 0;
                    if (lane == 63) {
#pragma unroll
                        for (int s2 = 0; s2 < (W ? 1 : 2); ++s2) {
                            const int B = (W ? accB : (s2 ? half_hi(accB) : half_lo(accB))) - BIASW;
                            const int S = (W ? accS : (s2 ? half_hi(accS) : half_lo(accS))) - 2 * BIASW;
                            const int lo = sp.min_score > 1 ? sp.min_score : 1;
                            const int V = imax(imax(S, B), (s2 ? a_of_b : a_of_a) + 1);
                            const int best = V >> 1;
                            int flag = 1;
                            if (V & 1) flag = 0;
                            else if ((B >> 1) >= best) flag = ((S >> 1) >= best) ? 2 : 0;
                            const int v = ((best >= lo ? best : -1) << 2) | flag;
                            if (s2) vb = v; else va = v;
                        }
                    }
                    out_a = dpp_rol1(out_a); out_b = dpp_rol1(out_b);
                    if (lane == 63) { out_a = va; out_b = vb; }
                    ++kcur; ++n_out;
                    if (n_out == 64) { flush(64); n_out = 0; }
                }
                ++bidx;
            }
            sS = dpp_rol1(sS); sB = dpp_rol1(sB);
            if (++pcnt == m) pcnt = 0;
        }
    }
    if (DIR == 0) a_all = mx2<W>(a_all, M);
    else if (last_blk && n_out > 0) flush(n_out);
  }   // row blocks
    if (DIR == 0) {
        // A = best alignment inside R (doubled) = the maximum over every cell of the sweep
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a_all = mx2<W>(a_all, __shfl_xor(a_all, off, 64));
        if (lane == 0) {
            read_a[ra] = (W ? a_all : half_lo(a_all)) - BIASW;
            if (has_b) read_a[rb] = half_hi(a_all) - BIASW;
        }
    }
  }   // tasks
}

// ------------------------------------------------------------------------------------
// k_sweep_ringmt: the chained LDS-ring sweep with the row blocks of a read as CONCURRENT waves.
//
// k_sweep_ringchain runs the blocks of a task one after the other in one wave: a batch of few long reads (2000
// cores of 3.7 kb = 1000 tasks of three blocks) then has one wave per SIMD.  Here a wave is one (task, row block):
// block b+1 starts as soon as block b has published the columns it needs and follows 64 * (skew + 1) steps behind,
// so the blocks of a read overlap and the launch has (blocks per read) times the waves.
//
// Hand-off (cdna_hip_programming.md Guideline 16, R2: the data is the flag): what lane 63 of block b leaves per
// column -- H of the block's last row, the two vertical-gap states and, on boundary columns, the S and B
// accumulators -- is stored as 8-byte granules {epoch, value}, ONE agent-scope (sc1, write-through) store each, into
// the strip between the two blocks; block b+1 loads the granules of the 64 columns it is about to hand to its
// lane 0 with agent-scope (sc1) loads and polls until every tag carries this launch's epoch.  No fence, no flag;
// the strips are zeroed once when the batch is created and every launch has an epoch of its own (never 0), so a
// granule of an earlier launch or run never passes.
//
// No deadlock, whatever the dispatch order: a wave takes its (task, block) by TICKET -- one atomic add when it
// starts -- and the block list is ordered so that a producer precedes its consumer (all blocks 0, then all blocks 1,
// ...).  Whoever holds a ticket runs; every smaller ticket was taken by a wave that has started, so the producer a
// wave waits for is running or done and never waits for a larger ticket.  Spins sleep (s_sleep), are bounded, and
// watch a launch-wide error word: a wave that gives up sets it, the others leave, the host reports NRA_E_DEVICE.
typedef unsigned long long nra_u64;
typedef __attribute__((address_space(1))) nra_u64 nra_gu64;
typedef __attribute__((address_space(1))) int nra_gi32;
// Polls (2 - 3 us each: an s_sleep and 3 - 5 agent-scope loads) before a waiting wave gives up: ~2.5 s, three orders above
// the longest legitimate wait (64 * (skew + 1) steps of the block above, a millisecond or two).  A hang guard, not a
// path: the argument that no wave waits for ever (tickets: a producer holds a smaller ticket than its consumer, so it
// is running or done) assumes that a wave, once started, stays resident until it ends -- true as long as nothing
// preempts the queue (no CWSR / time-slicing against another process on the device); if that ever fails the guard
// turns a hang into NRA_E_DEVICE from nra_batch_sync / nra_batch1d_fetch.
#define NRA_MT_SPIN_LIMIT (1u << 20)

template <int R, bool HAS_N, int DIR, bool W>
__global__ __launch_bounds__(WAVE, 3) void k_sweep_ringmt(int n_blocks, const NraChainBlock* __restrict__ blocks,
                                                       int32_t* ticket, const NraSweepTask* __restrict__ tasks,
                                                       const NraDevRead* __restrict__ reads,
                                                       const NraDevRegion* __restrict__ regions,
                                                       const uint8_t* __restrict__ pool,
                                                       const uint32_t* __restrict__ q2bit,
                                                       const uint32_t* __restrict__ qnmask,
                                                       NraScoreParams sp,
                                                       const int32_t* __restrict__ kmin_arr,
                                                       const int32_t* __restrict__ kmax_arr,
                                                       const uint32_t* __restrict__ coff,
                                                       int32_t* __restrict__ snap,
                                                       int32_t* read_a,
                                                       int32_t* __restrict__ cand_score,
                                                       uint8_t* __restrict__ cand_flag,
                                                       nra_u64* strips, int chain_cap, uint32_t epoch, int32_t* error)
{
    constexpr int SC = 2;
    constexpr int BIASW = W ? 0 : BIAS;
    __shared__ int4 ring[SWEEP_RING_D * 64];
    __shared__ int2 racc[64];
    const int lane = threadIdx.x;
    const int wr = (lane + 1) & 63;
    int my = 0;
    if (lane == 0) my = atomicAdd(ticket, 1);
    my = __builtin_amdgcn_readfirstlane(my);
    if (my >= n_blocks) return;
    const NraChainBlock cb = blocks[my];
    const NraSweepTask tk = tasks[cb.task];
    const int blk = cb.blk;
    const bool first_blk = blk == 0, last_blk = blk == cb.nblk - 1;
    nra_gu64* cin = (nra_gu64*)(strips + (size_t)(first_blk ? 0 : cb.strip_in) * 5 * (size_t)chain_cap);
    nra_gu64* cout = (nra_gu64*)(strips + (size_t)(last_blk ? 0 : cb.strip_out) * 5 * (size_t)chain_cap);
    nra_gi32* err = (nra_gi32*)error;
    const nra_u64 tag = (nra_u64)epoch << 32;

    const bool has_b = !W && tk.read_b >= 0;
    const int ra = tk.read_a, rb = has_b ? tk.read_b : tk.read_a;
    const NraDevRead rda = reads[ra], rdb = reads[rb];
    const NraDevRegion rg = regions[rda.region];
    const int m = rg.m1;
    const int flank = DIR ? rg.l1 : rg.l3;
    const uint8_t* __restrict__ piece = pool + (DIR ? rg.p1_off : rg.pr_off);
    const int ncols = DIR ? flank + m * tk.kmax : flank;
    const int jfirst = DIR ? flank + m * tk.kmin - 1 : flank - 1;
    const int skew = DIR ? m : 1;
    const int kmin_a = kmin_arr[ra], kmax_a = kmax_arr[ra];
    const int kmin_b = kmin_arr[rb], kmax_b = kmax_arr[rb];
    const uint32_t coff_a = coff[ra], coff_b = coff[rb];
    int32_t* __restrict__ snap_task = snap + tk.snap_off;

    const int o1 = SC * sp.open1, o2 = SC * sp.open2;
    const int P1 = W ? 1 : 0x00010001;
    const int v_floor = (BIASW - o1) * P1;
    const int v_o1 = o1 * P1, v_e1 = SC * sp.ext1 * P1, v_o2 = o2 * P1, v_e2 = SC * sp.ext2 * P1;
    const int NEG1 = W ? -(1 << 28) : NEGB * P1;
    const int NEG2 = W ? -(1 << 28) : 2 * NEGB * P1;
    const int s_match = SC * sp.match + o1, s_mis = o1 - SC * sp.mismatch, s_ambi = o1 - SC * sp.ambi;
    const int tbl_hi = s_mis | (s_ambi << 8);
    const int tbl_mis4 = s_mis * 0x01010101, tbl_ambi4 = s_ambi * 0x01010101;

    auto column_table = [&](int col) {
        int t = tbl_mis4;
        if (col >= 0 && col < ncols) {
            const int code = piece[col];
            t = code < 4 ? tbl_mis4 + ((s_match - s_mis) << (8 * code)) : tbl_ambi4;
            if (DIR == 0 && col == flank - 1) t |= FLAG_SNAPSHOT;
            if (DIR == 1 && col + 1 >= flank) t |= FLAG_INREP;     // origin bit of an alignment starting at the NEXT column
        }
        return t;
    };

    // What enters row 0 of this block at template column `col` (each lane its own column): constants for the
    // first block and outside the template, else the granules the block above has published -- polled until they
    // carry this launch's epoch.  Wave-uniform control flow; returns false when the launch has failed.
    auto fetch = [&](int col, int& h, int& f, int& f2, int& s2, int& b2) -> bool {
        h = v_floor; f = NEG1; f2 = NEG1; s2 = NEG2; b2 = NEG1;
        const bool mine = !first_blk && col >= 0 && col < ncols;
        const bool at_boundary = DIR == 1 && mine && col >= jfirst && (col - jfirst) % m == 0;
        if (first_blk) return true;
        const nra_gu64* g = cin + (mine ? col : 0);
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
            if (mine) {
                const nra_u64 x0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const nra_u64 x1 = __hip_atomic_load(g + chain_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const nra_u64 x2 = __hip_atomic_load(g + 2 * (size_t)chain_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (x0 >> 32) == epoch && (x1 >> 32) == epoch && (x2 >> 32) == epoch;
                h = (int)(unsigned)x0; f = (int)(unsigned)x1; f2 = (int)(unsigned)x2;
                if (at_boundary) {
                    const nra_u64 x3 = __hip_atomic_load(g + 3 * (size_t)chain_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const nra_u64 x4 = __hip_atomic_load(g + 4 * (size_t)chain_cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (x3 >> 32) == epoch && (x4 >> 32) == epoch;
                    s2 = (int)(unsigned)x3; b2 = (int)(unsigned)x4;
                }
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;
            __builtin_amdgcn_s_sleep(32);
            if ((spins & 15) == 15) {
                int failed = 0;
                if (lane == 0) failed = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane(failed) != 0) return false;
                if (spins >= NRA_MT_SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
    };
    auto publish = [&](int plane, int col, int v) {
        __hip_atomic_store(cout + (size_t)plane * chain_cap + col, tag | (nra_u64)(unsigned)v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    };

    int out_a = 0, out_b = 0;
    int n_out = 0, kcur = tk.kmin;                          // wave-uniform
    auto flush = [&](int n_valid) {
        const int k = kcur - 64 + lane;
        if (lane < 64 - n_valid) return;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 == 1 && !has_b) break;
            const int lo_k = s2 ? kmin_b : kmin_a, hi_k = s2 ? kmax_b : kmax_a;
            if (k < lo_k || k > hi_k) continue;
            const uint32_t idx = (s2 ? coff_b : coff_a) + (uint32_t)(k - lo_k);
            const int v = s2 ? out_b : out_a;
            cand_score[idx] = v >> 2;
            cand_flag[idx] = (uint8_t)(v & 3);
        }
    };
    const int a_of_a = DIR ? read_a[ra] : 0, a_of_b = DIR ? read_a[rb] : 0;

    const int nsteps = ncols + 63 * skew;
    const int row_base = blk * 64 * R;

    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int gi = row_base + lane * R + i;
        const int ca = sweep_query_sel<HAS_N>(rda, q2bit, qnmask, gi, DIR == 0);
        const int cb2 = W ? 0x0c : sweep_query_sel<HAS_N>(rdb, q2bit, qnmask, gi, DIR == 0);
        qc[i] = ca | (0x0c << 8) | (cb2 << 16) | (0x0c << 24);
    }
    int Hbo[DIR ? R : 1], Ebo[DIR ? R : 1], E2bo[DIR ? R : 1];
    if (DIR) {
        const int q1 = SC * (sp.open1 - sp.ext1), q2 = SC * (sp.open2 - sp.ext2);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = row_base + lane * R + i;
            int h[2], e[2], e2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int a = (s ? rdb.qlen : rda.qlen) - 2 - r;
                if (a >= 0) {
                    const int ablk = a / (64 * R);
                    const int w = a - ablk * 64 * R;
                    const int al = w / R, ai = w - al * R;
                    const int32_t* __restrict__ p = snap_task + ((size_t)ablk * 3 * R + ai) * 64 + al;
                    const int vh = p[0], ve = p[R * 64], ve2 = p[2 * R * 64];
                    h[s] = (W ? vh : (s ? half_hi(vh) : half_lo(vh))) + 2 * o1;
                    e[s] = (W ? ve : (s ? half_hi(ve) : half_lo(ve))) + q1;
                    e2[s] = (W ? ve2 : (s ? half_hi(ve2) : half_lo(ve2))) + q2;
                } else { h[s] = BIASW + o1 - SC; e[s] = BIASW - SC; e2[s] = BIASW - SC; }
            }
            Hbo[i] = W ? h[0] : pack2(h[0], h[1]);
            Ebo[i] = W ? e[0] : pack2(e[0], e[1]);
            E2bo[i] = W ? e2[0] : pack2(e2[0], e2[1]);
        }
    }
    int Hq[R], Hq2[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hq[i] = v_floor; Hq2[i] = NEG1; E[i] = NEG1; E2[i] = NEG1; }

    // ring: padding columns everywhere, then lane 0's first `skew` columns (and, when the first boundary is one
    // of them, its accumulators; a later first boundary gets them from lane 63 one boundary phase ahead, below)
#pragma unroll
    for (int s = 0; s < SWEEP_RING_D; ++s) ring[s * 64 + lane] = make_int4(v_floor, NEG1, NEG1, tbl_mis4);
    racc[lane] = make_int2(NEG2, NEG1);
    int feed = tbl_mis4, sH, sF, sF2, sS, sB;   // what lane 63 hands to lane 0
    if (!fetch(lane < skew ? lane : -1, sH, sF, sF2, sS, sB)) return;
    if (lane < skew) ring[lane * 64] = make_int4(sH, sF, sF2, column_table(lane));
    if (DIR == 1 && lane == jfirst && jfirst < skew) racc[0] = make_int2(sS, sB);
    ring_order();

    int Hup_prev = v_floor, M = BIASW * P1;
    int slot = 0;
    int phase = jfirst % m, pcnt = 0, bidx = 0;
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        if ((step & 63) == 0) {                                            // columns step + skew + (0..63)
            const int col = step + skew + wr;
            feed = column_table(col);
            if (!fetch(col, sH, sF, sF2, sS, sB)) return;
        }
        const int4 in = ring[slot * 64 + lane];
        const int tt = in.w;
        const int fl = DIR == 1 ? (int)((((unsigned)tt >> 15) & (unsigned)P1) | (unsigned)(BIASW * P1)) : BIASW * P1;   // as in k_sweep_ring
        int F = mx2<W>(in.y, fl), F2 = in.z;
        sweep_cell<0, R, R, W, true>(Hq, Hq2, E, E2, qc, Hup_prev, F, F2, M, tt & 0x7f7f7f7f, tbl_hi, fl, v_e1, v_e2, v_o1, v_o2);
        Hup_prev = mx2<W>(in.x, fl - v_o1);
        ring[slot * 64 + wr] = make_int4(Hq[R - 1], F, F2, tt);
        if (lane == 63) ring[slot * 64] = make_int4(sH, sF, sF2, feed);
        ring_order();                                       // the next steps' loads stay behind these stores
        const int c63 = step - 63 * skew;                                  // the column lane 63 has just finished
        if (lane == 63 && !last_blk && c63 >= 0) { publish(0, c63, Hq[R - 1]); publish(1, c63, F); publish(2, c63, F2); }
        feed = dpp_rol1(feed); sH = dpp_rol1(sH); sF = dpp_rol1(sF); sF2 = dpp_rol1(sF2);
        if (++slot == skew) slot = 0;

        if constexpr (DIR == 0) {
            if (tt & FLAG_SNAPSHOT) sweep_snapshot<0, R, R>(Hq, E, E2, snap_task + (size_t)blk * 3 * R * 64, lane);
        } else {
            if (pcnt == phase) {                                           // wave-uniform
                if (step >= jfirst) {                                      // every lane on a unit boundary
                    const int tS = sweep_combine<0, R, R, W>(Hq, E, E2, Hbo, Ebo, E2bo, NEG2);
                    const int2 acc = racc[lane];
                    const int accS = mx2<W>(acc.x, tS), accB = mx2<W>(acc.y, M);
                    racc[wr] = make_int2(accS, accB);
                    if (lane == 63) {
                        racc[0] = make_int2(sS, sB);                       // of column step + skew: lane 0's next boundary
                        if (!last_blk && c63 >= 0) { publish(3, c63, accS); publish(4, c63, accB); }
                    }
                    ring_order();
                    if (last_blk && bidx >= 63 && kcur <= tk.kmax) {       // lane 63 is on the boundary of k = kcur
                        int va = 0, vb = 0;
                        if (lane == 63) {
#pragma unroll
                            for (int s2 = 0; s2 < (W ? 1 : 2); ++s2) {
                                const int B = (W ? accB : (s2 ? half_hi(accB) : half_lo(accB))) - BIASW;
                                const int S = (W ? accS : (s2 ? half_hi(accS) : half_lo(accS))) - 2 * BIASW;
                                const int lo = sp.min_score > 1 ? sp.min_score : 1;
                                const int V = imax(imax(S, B), (s2 ? a_of_b : a_of_a) + 1);
                                const int best = V >> 1;
                                int flag = 1;
                                if (V & 1) flag = 0;
                                else if ((B >> 1) >= best) flag = ((S >> 1) >= best) ? 2 : 0;
                                const int v = ((best >= lo ? best : -1) << 2) | flag;
                                if (s2) vb = v; else va = v;
                            }
                        }
                        out_a = dpp_rol1(out_a); out_b = dpp_rol1(out_b);
                        if (lane == 63) { out_a = va; out_b = vb; }
                        ++kcur; ++n_out;
                        if (n_out == 64) { flush(64); n_out = 0; }
                    }
                    ++bidx;
                } else if (step + m >= jfirst) {                           // one boundary phase before lane 0's first boundary
                    if (lane == 63) racc[0] = make_int2(sS, sB);           // of column step + skew = jfirst
                    ring_order();
                }
            }
            sS = dpp_rol1(sS); sB = dpp_rol1(sB);
            if (++pcnt == m) pcnt = 0;
        }
    }
    if (DIR == 0) {
        // A = best alignment inside R (doubled) = the maximum over every cell of every block: the blocks are waves
        // of their own, so each adds its maximum to the read's word (zeroed before the launch; A >= 0)
        int a_all = M;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a_all = mx2<W>(a_all, __shfl_xor(a_all, off, 64));
        if (lane == 0) {
            atomicMax(&read_a[ra], (W ? a_all : half_lo(a_all)) - BIASW);
            if (has_b) atomicMax(&read_a[rb], half_hi(a_all) - BIASW);
        }
    } else if (last_blk && n_out > 0) {
        flush(n_out);
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
template <int DIR>
static int launch_sweep(int R, int has_n, int chain, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                        const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                        const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                        const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                        int32_t* snap, int32_t* read_a,
                        int32_t* cand_score, uint8_t* cand_flag, int32_t* chain_buf, int chain_cap, int n_strips)
{
    if (n_tasks <= 0) return 0;
#define ARGS n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag, chain_buf, chain_cap
    if (chain) {      // row-block chaining: only the instantiations long reads (and the tests) use
        if (n_strips <= 0) return (int)hipErrorInvalidValue;
        const int n_waves = n_tasks < n_strips ? n_tasks : n_strips;
        if (R == NRA_CHAIN_R) {
            if (has_n) k_sweep_pk16<NRA_CHAIN_R, true, DIR, true><<<n_waves, WAVE, 0, st>>>(ARGS);
            else k_sweep_pk16<NRA_CHAIN_R, false, DIR, true><<<n_waves, WAVE, 0, st>>>(ARGS);
        } else if (R == NRA_CHAIN_R_TEST) {
            if (has_n) k_sweep_pk16<NRA_CHAIN_R_TEST, true, DIR, true><<<n_waves, WAVE, 0, st>>>(ARGS);
            else k_sweep_pk16<NRA_CHAIN_R_TEST, false, DIR, true><<<n_waves, WAVE, 0, st>>>(ARGS);
        } else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_sweep_pk16<r, true, DIR, false><<<n_tasks, WAVE, 0, st>>>(ARGS); \
        else k_sweep_pk16<r, false, DIR, false><<<n_tasks, WAVE, 0, st>>>(ARGS);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
#undef ARGS
    return (int)hipGetLastError();
}

template <int DIR>
static int launch_sweep_ring(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                             const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                             const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                             const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                             int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    if (n_tasks <= 0) return 0;
#define ARGS n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag
#define CASE(r)                                                                          \
    case r:                                                                              \
        if (has_n) k_sweep_ring<r, true, DIR><<<(n_tasks + SWEEP_RING_WPB - 1) / SWEEP_RING_WPB, WAVE * SWEEP_RING_WPB, 0, st>>>(ARGS); \
        else k_sweep_ring<r, false, DIR><<<(n_tasks + SWEEP_RING_WPB - 1) / SWEEP_RING_WPB, WAVE * SWEEP_RING_WPB, 0, st>>>(ARGS);      \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
#undef ARGS
    return (int)hipGetLastError();
}

#if NRA_HAS_PART(11)
extern "C" int nra_launch_sweep_ring_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                         const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                         int32_t* snap, int32_t* read_a)
{
    return launch_sweep_ring<0>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                                coff, snap, read_a, nullptr, nullptr);
}
#endif
#if NRA_HAS_PART(12)
extern "C" int nra_launch_sweep_ring_fwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                         const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                         int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    return launch_sweep_ring<1>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                                coff, snap, read_a, cand_score, cand_flag);
}
#endif

template <int DIR>
static int launch_sweep_ringchain(int R, int has_n, int wide, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                  const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                  const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                  const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                  int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag,
                                  int32_t* chain_buf, int chain_cap, int n_strips)
{
    if (n_tasks <= 0) return 0;
    const int grid = n_tasks < n_strips ? n_tasks : n_strips;
#define ARGS n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag, chain_buf, chain_cap
#define LAUNCH(r)                                                                                        \
    do {                                                                                                 \
        if (wide) { if (has_n) k_sweep_ringchain<r, true, DIR, true><<<grid, WAVE, 0, st>>>(ARGS);      \
                    else k_sweep_ringchain<r, false, DIR, true><<<grid, WAVE, 0, st>>>(ARGS); }         \
        else { if (has_n) k_sweep_ringchain<r, true, DIR, false><<<grid, WAVE, 0, st>>>(ARGS);          \
               else k_sweep_ringchain<r, false, DIR, false><<<grid, WAVE, 0, st>>>(ARGS); }             \
    } while (0)
    if (R == NRA_RING_CHAIN_R) LAUNCH(NRA_RING_CHAIN_R);
    else if (R == NRA_CHAIN_R_TEST) LAUNCH(NRA_CHAIN_R_TEST);
    else return (int)hipErrorInvalidValue;
#undef LAUNCH
#undef ARGS
    return (int)hipGetLastError();
}

template <int DIR>
static int launch_sweep_ring32(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                               const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                               const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                               const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                               int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    if (n_tasks <= 0) return 0;
#define ARGS n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag
#define CASE(r)                                                                          \
    case r:                                                                              \
        if (has_n) k_sweep_ring32<r, true, DIR><<<n_tasks, WAVE, 0, st>>>(ARGS);         \
        else k_sweep_ring32<r, false, DIR><<<n_tasks, WAVE, 0, st>>>(ARGS);              \
        break;
    switch (R) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16) CASE(18) CASE(20) CASE(22) CASE(24)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
#undef ARGS
    return (int)hipGetLastError();
}

// k_sweep_ringq launchers: `half` = the half-wave kernel's buckets (two read pairs per wave, R <= NRA_RING32_MAX_R)
#define NRA_Q_LAUNCH_ARGS n_quanta, qlist, qsteps, n_tasks, ticket, arrivals, giveup, qstate, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag
#if NRA_HAS_PART(25)
extern "C" int nra_launch_sweep_ringq(int R, int has_n, hipStream_t st, int n_quanta, const uint32_t* qlist, int qsteps, int n_tasks, int32_t* ticket,
                                      int32_t* arrivals, int32_t* giveup, int32_t* qstate, const NraSweepTask* tasks,
                                      const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                      const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                      const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                      int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    if (n_quanta <= 0) return 0;
#define CASE(r)                                                                                          \
    case r:                                                                                              \
        if (has_n) k_sweep_ringq<r, true, false><<<n_quanta, WAVE, 0, st>>>(NRA_Q_LAUNCH_ARGS);          \
        else k_sweep_ringq<r, false, false><<<n_quanta, WAVE, 0, st>>>(NRA_Q_LAUNCH_ARGS);               \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}
#endif
#if NRA_HAS_PART(26)
extern "C" int nra_launch_sweep_ringq32(int R, int has_n, hipStream_t st, int n_quanta, const uint32_t* qlist, int qsteps, int n_tasks, int32_t* ticket,
                                        int32_t* arrivals, int32_t* giveup, int32_t* qstate, const NraSweepTask* tasks,
                                        const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                        const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                        const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                        int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    if (n_quanta <= 0) return 0;
#define CASE(r)                                                                                          \
    case r:                                                                                              \
        if (has_n) k_sweep_ringq<r, true, true><<<n_quanta, WAVE, 0, st>>>(NRA_Q_LAUNCH_ARGS);           \
        else k_sweep_ringq<r, false, true><<<n_quanta, WAVE, 0, st>>>(NRA_Q_LAUNCH_ARGS);                \
        break;
    switch (R) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16) CASE(18) CASE(20) CASE(22) CASE(24)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}
#endif
#undef NRA_Q_LAUNCH_ARGS

#if NRA_HAS_PART(15)
extern "C" int nra_launch_sweep_ring32_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                           const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                           const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                           const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                           int32_t* snap, int32_t* read_a)
{
    return launch_sweep_ring32<0>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                                  coff, snap, read_a, nullptr, nullptr);
}
#endif
#if NRA_HAS_PART(16)
extern "C" int nra_launch_sweep_ring32_fwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                           const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                           const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                           const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                           int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag)
{
    return launch_sweep_ring32<1>(R, has_n, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                                  coff, snap, read_a, cand_score, cand_flag);
}
#endif

#if NRA_HAS_PART(13)
extern "C" int nra_launch_sweep_ringchain_bwd(int R, int has_n, int wide, hipStream_t st, int n_tasks,
                                              const NraSweepTask* tasks, const NraDevRead* reads,
                                              const NraDevRegion* regions, const uint8_t* pool,
                                              const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                              const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                              int32_t* snap, int32_t* read_a, int32_t* chain_buf, int chain_cap,
                                              int n_strips)
{
    return launch_sweep_ringchain<0>(R, has_n, wide, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin,
                                     kmax, coff, snap, read_a, nullptr, nullptr, chain_buf, chain_cap, n_strips);
}
#endif
#if NRA_HAS_PART(14)
extern "C" int nra_launch_sweep_ringchain_fwd(int R, int has_n, int wide, hipStream_t st, int n_tasks,
                                              const NraSweepTask* tasks, const NraDevRead* reads,
                                              const NraDevRegion* regions, const uint8_t* pool,
                                              const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                              const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                              int32_t* snap, int32_t* read_a, int32_t* cand_score,
                                              uint8_t* cand_flag, int32_t* chain_buf, int chain_cap, int n_strips)
{
    return launch_sweep_ringchain<1>(R, has_n, wide, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin,
                                     kmax, coff, snap, read_a, cand_score, cand_flag, chain_buf, chain_cap, n_strips);
}
#endif

template <int DIR>
static int launch_sweep_ringmt(int R, int has_n, int wide, hipStream_t st, int n_blocks, const NraChainBlock* blocks,
                               int32_t* ticket, const NraSweepTask* tasks, const NraDevRead* reads,
                               const NraDevRegion* regions, const uint8_t* pool, const uint32_t* q2bit,
                               const uint32_t* qnmask, NraScoreParams sp, const int32_t* kmin, const int32_t* kmax,
                               const uint32_t* coff, int32_t* snap, int32_t* read_a, int32_t* cand_score,
                               uint8_t* cand_flag, uint64_t* strips, int chain_cap, uint32_t epoch, int32_t* error)
{
    if (n_blocks <= 0) return 0;
    if (epoch == 0) return (int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ticket, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
#define ARGS n_blocks, blocks, ticket, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag, (nra_u64*)strips, chain_cap, epoch, error
#define LAUNCH(r)                                                                                        \
    do {                                                                                                 \
        if (wide) { if (has_n) k_sweep_ringmt<r, true, DIR, true><<<n_blocks, WAVE, 0, st>>>(ARGS);      \
                    else k_sweep_ringmt<r, false, DIR, true><<<n_blocks, WAVE, 0, st>>>(ARGS); }         \
        else { if (has_n) k_sweep_ringmt<r, true, DIR, false><<<n_blocks, WAVE, 0, st>>>(ARGS);          \
               else k_sweep_ringmt<r, false, DIR, false><<<n_blocks, WAVE, 0, st>>>(ARGS); }             \
    } while (0)
    // (the host picks the block height that pads a bucket's reads least: NRA_RING_MT_R_MIN .. NRA_RING_MT_R rows per lane)
    switch (R) {
    case 15: LAUNCH(15); break;
    case 14: LAUNCH(14); break;
    case 13: LAUNCH(13); break;
    case 12: LAUNCH(12); break;
    case NRA_CHAIN_R_TEST: LAUNCH(NRA_CHAIN_R_TEST); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef LAUNCH
#undef ARGS
    return (int)hipGetLastError();
}

#if NRA_HAS_PART(18)
extern "C" int nra_launch_sweep_ringmt_bwd(int R, int has_n, int wide, hipStream_t st, int n_blocks,
                                           const NraChainBlock* blocks, int32_t* ticket, const NraSweepTask* tasks,
                                           const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                           const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                           const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                           int32_t* snap, int32_t* read_a, uint64_t* strips, int chain_cap,
                                           uint32_t epoch, int32_t* error)
{
    return launch_sweep_ringmt<0>(R, has_n, wide, st, n_blocks, blocks, ticket, tasks, reads, regions, pool, q2bit, qnmask,
                                  sp, kmin, kmax, coff, snap, read_a, nullptr, nullptr, strips, chain_cap, epoch, error);
}
#endif
#if NRA_HAS_PART(19)
extern "C" int nra_launch_sweep_ringmt_fwd(int R, int has_n, int wide, hipStream_t st, int n_blocks,
                                           const NraChainBlock* blocks, int32_t* ticket, const NraSweepTask* tasks,
                                           const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                           const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                           const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                           int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag,
                                           uint64_t* strips, int chain_cap, uint32_t epoch, int32_t* error)
{
    return launch_sweep_ringmt<1>(R, has_n, wide, st, n_blocks, blocks, ticket, tasks, reads, regions, pool, q2bit, qnmask,
                                  sp, kmin, kmax, coff, snap, read_a, cand_score, cand_flag, strips, chain_cap, epoch, error);
}
#endif

#if NRA_HAS_PART(5)
extern "C" int nra_launch_sweep_bwd(int R, int has_n, int chain, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                    const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                    const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                    const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                    int32_t* snap,
                                    int32_t* read_a, int32_t* chain_buf, int chain_cap, int n_strips)
{
    return launch_sweep<0>(R, has_n, chain, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                           coff, snap, read_a, nullptr, nullptr, chain_buf, chain_cap, n_strips);
}
#endif
#if NRA_HAS_PART(6)
extern "C" int nra_launch_sweep_fwd(int R, int has_n, int chain, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                    const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                    const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                    const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                    int32_t* snap,
                                    int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag, int32_t* chain_buf,
                                    int chain_cap, int n_strips)
{
    return launch_sweep<1>(R, has_n, chain, st, n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, kmin, kmax,
                           coff, snap, read_a, cand_score, cand_flag, chain_buf, chain_cap, n_strips);
}
#endif
