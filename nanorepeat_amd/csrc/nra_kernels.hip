// nra_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the repeat-size scoring path.
//
// One wavefront (64 lanes) computes one optimal local alignment (two-piece affine gap)
// of a read against an implicit candidate template as a systolic array:
//
//   * lane l owns query rows [l*R, l*R+R) -- their bases, H, E and E2 live in VGPRs
//     (R is a template parameter, the row loop is fully unrolled, nothing spills);
//   * the wave sweeps the template left to right, lane l working on column t-l at step t,
//     so the cells of one step form an anti-diagonal band and are independent;
//   * the only cross-lane traffic per step is three values handed from lane l-1 to lane l
//     (H of its last row, and the two vertical gap states) plus the template base, all
//     moved with DPP wave_shr:1 -- no LDS round trip, no barrier;
//   * template bases are never materialised per candidate: a lane fetches 1 base per 64
//     steps from the pre-expanded pieces (L+unit^kmax | mid+unit2^k2max | R) in the L2-
//     resident pool and the bases circulate through the wave by DPP rotate.
//
// Integer VALU only (no MFMA: DP recurrences are max-plus, not a contraction).
//
//   k_score_pk16   scores TWO candidates of one read per wave in packed int16 halves
//                  (v_pk_add/max/min_i16, v_pk_mad): the dominant kernel.
//   k_payload_i32  scores one candidate in int32 with a 16-bit payload packed under the
//                  score: tstart (ORIGIN, 1D flank test) or the CIGAR-window score
//                  (WINDOW, 2D selector).  max() on the packed word is the
//                  lexicographic (score, payload) max the oracle defines.
//   select kernels implement the reference's selectors.
//
// Semantics: oracle/nr_oracle.c (header comment) is the contract; results are bit-identical.
#include "nra_internal.h"

// NRA_PART splits this file into translation units that build in parallel:
// 0/undefined = everything, 1 = k_score_pk16, 2 = k_payload_i32 ORIGIN, 3 = k_payload_i32 WINDOW,
// 4 = selectors.
#ifndef NRA_PART
#define NRA_PART 0
#endif
#define NRA_HAS_PART(n) (NRA_PART == 0 || NRA_PART == (n))

#include "nra_device.h"

#if NRA_HAS_PART(1)
// ------------------------------------------------------------------------------------
// k_score_pk16: two candidates per wave, packed int16
// ------------------------------------------------------------------------------------
template <int R, bool HAS_N>
__global__ __launch_bounds__(WAVE) void k_score_pk16(int n_tasks, const NraPairTask* __restrict__ tasks,
                                                     const NraDevRead* __restrict__ reads,
                                                     const NraDevRegion* __restrict__ regions,
                                                     const uint8_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ q2bit,
                                                     const uint32_t* __restrict__ qnmask,
                                                     NraScoreParams sp, int32_t* __restrict__ out_score)
{
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    const NraPairTask tk = tasks[task];
    const NraDevRead rd = reads[tk.read];
    const NraDevRegion rg = regions[rd.region];
    const bool has_b = tk.out_b >= 0;
    const Tmpl ta = make_tmpl(rg, pool, tk.k1a, tk.k2a, 0);
    const Tmpl tb = has_b ? make_tmpl(rg, pool, tk.k1b, tk.k2b, tk.flags & 1) : ta;
    const int ncols = imax(ta.tlen, tb.tlen);

    // query rows of this lane, the base replicated into both halves
    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int c = query_code<HAS_N>(rd, q2bit, qnmask, lane * R + i);
        qc[i] = c | (c << 16);
    }

    // Registers hold (value + 8192) in both halves: subtracting a gap constant is then carry-safe
    // as one full-rate 32-bit v_sub_u32 (2 cycles per wave) instead of a half-rate v_pk_sub_i16
    // (4 cycles); maxima and the signed substitution add stay packed.
    constexpr int KB = 8192;
    const s16x2 NEGP = splat(1024);          // biased "minus infinity" (-7168): below any real state
    const s16x2 ZERO = splat(KB);
    const s16x2 v_match = splat(sp.match);
    const s16x2 v_negab = splat(-(sp.match + sp.mismatch));
    const int P1 = 0x00010001;
    const int v_open1 = sp.open1 * P1, v_ext1 = sp.ext1 * P1;
    const int v_open2 = sp.open2 * P1, v_ext2 = sp.ext2 * P1;
    const s16x2 v_negambi = splat(-sp.ambi);
    const s16x2 v_negb = splat(-sp.mismatch);
#define SUB32(a, b) as_s(as_i(a) - (b))

    s16x2 Hprev[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hprev[i] = ZERO; E[i] = NEGP; E2[i] = NEGP; }
    s16x2 Hbot = ZERO, Fout = NEGP, F2out = NEGP, Hup_prev = ZERO, M = ZERO;
    int tt = NRA_PAD_T | (NRA_PAD_T << 16);

    const int nchunks = (ncols + 63 + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int col = c * 64 + lane;
        int feed = tmpl_code(ta, col) | (tmpl_code(tb, col) << 16);
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
            // hand-off from the lane above (rows just before mine), one column behind me
            s16x2 F = as_s(dpp_shr1(as_i(NEGP), as_i(Fout)));
            s16x2 F2 = as_s(dpp_shr1(as_i(NEGP), as_i(F2out)));
            tt = dpp_shr1(feed, tt);          // lane 0 takes the next template column
            feed = dpp_rol1(feed);
            // substitution score: +a where the bases agree, -b otherwise.  x = q ^ t is 0 on a
            // match and 1..0x7f otherwise, so max(a - (a+b)*x, -b) needs no compare.
#define NRA_SUBST(i, out)                                                                          \
            {                                                                                      \
                const s16x2 x_ = as_s(qc[i] ^ tt);                                                 \
                out = pmax(x_ * v_negab + v_match, v_negb);                                        \
                if (HAS_N) {                                                                       \
                    const s16x2 n_ = as_s(((qc[i] | tt) >> 2) & 0x00010001);                       \
                    out = out + n_ * (v_negambi - out);                                            \
                }                                                                                  \
            }
            s16x2 sc;
            NRA_SUBST(0, sc);
            s16x2 d = pmax(Hup_prev, ZERO) + sc;   // diagonal of my first row came in one step ago
            Hup_prev = as_s(dpp_shr1(as_i(ZERO), as_i(Hbot)));
            s16x2 h = ZERO;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                s16x2 d_next = d;
                if (i + 1 < R) {                   // read H(i, j-1) before it is overwritten below
                    NRA_SUBST(i + 1, sc);
                    d_next = pmax(Hprev[i], ZERO) + sc;
                }
                h = pmax(pmax(d, E[i]), pmax(F, pmax(E2[i], F2)));
                M = pmax(M, h);
                Hprev[i] = h;
                const s16x2 hq = SUB32(h, v_open1);
                E[i] = pmax(SUB32(E[i], v_ext1), hq);
                F = pmax(SUB32(F, v_ext1), hq);
                const s16x2 hq2 = SUB32(h, v_open2);
                E2[i] = pmax(SUB32(E2[i], v_ext2), hq2);
                F2 = pmax(SUB32(F2, v_ext2), hq2);
                d = d_next;
            }
#undef NRA_SUBST
            Hbot = h; Fout = F; F2out = F2;
        }
    }
#undef SUB32
    // wave-wide max of both halves
    int m = as_i(M);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = as_i(pmax(as_s(m), as_s(__shfl_xor(m, off, WAVE))));
    if (lane == 0) {
        const int sa = (m & 0xffff) - KB, sb = ((m >> 16) & 0xffff) - KB;
        // flags bit 1: raw scores (the strand probe compares them even below min_dp_score)
        const int lo = (tk.flags & 2) ? 0 : (sp.min_score > 1 ? sp.min_score : 1);
        out_score[tk.out_a] = sa >= lo ? sa : -1;
        if (has_b) out_score[tk.out_b] = sb >= lo ? sb : -1;
    }
}

#endif  // part 1

#if NRA_HAS_PART(2) || NRA_HAS_PART(3)
// ------------------------------------------------------------------------------------
// k_payload: one candidate per wave, cells = (score << SH) | payload with the oracle's lexicographic max.
//   CELL = int32: SH 16 (score < 32768, template <= 65000 columns) -- the common case
//   CELL = int64: SH 32 -- long cores (scores beyond 32767), long targets (a whole ONT read as the
//                 target of an anchor), long joint reads; instantiated for three row counts only
// ------------------------------------------------------------------------------------
template <typename CELL, int R, bool HAS_N, int MODE, bool CHAIN>
__global__ __launch_bounds__(WAVE) void k_payload(const NraTask* __restrict__ tasks,
                                                  const int32_t* __restrict__ count,
                                                  const NraDevRead* __restrict__ reads,
                                                  const NraDevRegion* __restrict__ regions,
                                                  const uint8_t* __restrict__ pool,
                                                  const uint32_t* __restrict__ q2bit,
                                                  const uint32_t* __restrict__ qnmask,
                                                  NraScoreParams sp,
                                                  int32_t* __restrict__ out_score,
                                                  int32_t* __restrict__ out_p,
                                                  int32_t* __restrict__ out_tend,
                                                  void* chain_buf, int chain_cap)
{
    constexpr int SH = sizeof(CELL) == 8 ? 32 : 16;
    constexpr CELL NEGC = -((CELL)1 << (SH + 13));
    constexpr CELL PMASK = ((CELL)1 << SH) - 1;
    constexpr CELL WB = (CELL)1 << (SH - 1);             // bias of the window-score payload
    const int lane = threadIdx.x;
    const int n_tasks = *count;       // written by an earlier kernel on the same stream (or the host)
    // grid-stride over the queue: the trip count is wave-uniform and bounded by n_tasks, so
    // every wave drains; tasks of one queue cost about the same (same R, similar template length)
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const NraTask tk = tasks[task];
        const NraDevRead rd = reads[tk.read];
        // k2 < 0 marks a free (query, target) pair of nra_align_pairs: k1 is the target's region
        const bool pair = tk.k2 < 0;
        const NraDevRegion rg = regions[pair ? tk.k1 : rd.region];
        const Tmpl tm = make_tmpl(rg, pool, pair ? 0 : tk.k1, pair ? 0 : tk.k2, 0);
        const int ncols = tm.tlen;
        // window of nanoRepeat_joint.py:445-448 (WINDOW mode only)
        const int wa = imax(0, rg.l1 - 10);
        const int wb = imin(tm.tlen, tm.len1 + tm.len2 + 10);

        CELL best = PMASK;                 // (0, max payload): only cells with score >= 1 can beat it
        int bestj = -1;
        // CHAIN: a read longer than 64*R rows is swept in row blocks, one after the other in this wave;
        // lane 63 leaves its per-column hand-off (H, F, F2) in a wave-private strip for the next block
        const int n_blk = CHAIN ? (rd.qlen + 64 * R - 1) / (64 * R) : 1;
        volatile CELL* strip = CHAIN ? (CELL*)chain_buf + (size_t)blockIdx.x * 6 * chain_cap : nullptr;
      for (int blk = 0; blk < n_blk; ++blk) {
        const int row_base = blk * 64 * R;
        const bool first_blk = blk == 0, last_blk = blk == n_blk - 1;
        volatile CELL* cin = CHAIN ? strip + ((blk + 1) & 1) * 3 * chain_cap : nullptr;
        volatile CELL* cout = CHAIN ? strip + (blk & 1) * 3 * chain_cap : nullptr;

        int qc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) qc[i] = query_code<HAS_N>(rd, q2bit, qnmask, row_base + lane * R + i);

        CELL Hprev[R], E[R], E2[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { Hprev[i] = NEGC; E[i] = NEGC; E2[i] = NEGC; }
        CELL Hbot = NEGC, Fout = NEGC, F2out = NEGC, Hup_prev = NEGC;
        int tt = NRA_PAD_T;
        int j = -lane;

        const CELL sA = (CELL)sp.match << SH, sB = -((CELL)sp.mismatch << SH), sN = -((CELL)sp.ambi << SH);
        const CELL o1 = -((CELL)sp.open1 << SH), x1 = -((CELL)sp.ext1 << SH);
        const CELL o2 = -((CELL)sp.open2 << SH), x2 = -((CELL)sp.ext2 << SH);

        const int nchunks = (ncols + 63 + 63) >> 6;
        for (int c = 0; c < nchunks; ++c) {
            int feed = tmpl_code(tm, c * 64 + lane);
            CELL inH = NEGC, inF = NEGC, inF2 = NEGC;      // what enters lane 0 at each column
            if (CHAIN) {
                const int col = c * 64 + lane;
                if (!first_blk && col < ncols) { inH = cin[col]; inF = cin[chain_cap + col]; inF2 = cin[2 * chain_cap + col]; }
            }
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                CELL F = dpp_shr1(CHAIN ? inF : NEGC, Fout);
                CELL F2 = dpp_shr1(CHAIN ? inF2 : NEGC, F2out);
                tt = dpp_shr1(feed, tt);
                feed = dpp_rol1(feed);

                CELL fresh, s_eq, s_ne, n_eq, n_ne, eo1, ex1, eo2, ex2, fo1, fx1, fo2, fx2;
                if (MODE == 0) {
                    fresh = (CELL)j;                         // (0, origin = this column)
                    s_eq = sA; s_ne = sB; n_eq = sN; n_ne = sN;
                    eo1 = fo1 = o1; ex1 = fx1 = x1; eo2 = fo2 = o2; ex2 = fx2 = x2;
                } else {
                    fresh = WB;
                    const bool inw = (j >= wa) && (j < wb);
                    const int pe = inw ? 2 : 0, pn = inw ? -4 : 0;          // tk.py:464-475
                    s_eq = sA + pe; s_ne = sB + pn; n_eq = sN + pe; n_ne = sN + pn;
                    const int jn = j + 1;                                    // E written below is column j+1
                    const bool inwn = (jn >= wa) && (jn < wb);
                    const int po = inwn ? -4 : 0;                            // tk.py:480-485
                    const int px = inwn ? (jn == wa ? -4 : -2) : 0;
                    eo1 = o1 + po; ex1 = x1 + px; eo2 = o2 + po; ex2 = x2 + px;
                    const bool fin = (jn > wa) && (jn < wb - 1);             // tk.py:476-479, ref_pos = j+1
                    const int qo = fin ? -4 : 0, qx = fin ? -2 : 0;
                    fo1 = o1 + qo; fx1 = x1 + qx; fo2 = o2 + qo; fx2 = x2 + qx;
                }
#define NRA_SUBST(i, out)                                                                          \
                {                                                                                  \
                    const bool eq_ = qc[i] == tt;                                                  \
                    out = eq_ ? s_eq : s_ne;                                                       \
                    if (HAS_N) {                                                                   \
                        if ((qc[i] | tt) & 4) out = eq_ ? n_eq : n_ne;                             \
                    }                                                                              \
                }
                CELL sc;
                NRA_SUBST(0, sc);
                CELL d = imax(Hup_prev, fresh) + sc;   // diagonal of my first row came in one step ago
                Hup_prev = dpp_shr1(CHAIN ? inH : NEGC, Hbot);
                if (CHAIN) { inH = dpp_rol1(inH); inF = dpp_rol1(inF); inF2 = dpp_rol1(inF2); }
                CELL colmax = NEGC, h = NEGC;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    CELL d_next = d;
                    if (i + 1 < R) {                  // read H(i, j-1) before it is overwritten below
                        NRA_SUBST(i + 1, sc);
                        d_next = imax(Hprev[i], fresh) + sc;
                    }
                    h = imax(imax(d, E[i]), F);
                    h = imax(imax(h, E2[i]), F2);
                    colmax = imax(colmax, h);
                    Hprev[i] = h;
                    E[i] = imax(E[i] + ex1, h + eo1);
                    F = imax(F + fx1, h + fo1);
                    E2[i] = imax(E2[i] + ex2, h + eo2);
                    F2 = imax(F2 + fx2, h + fo2);
                    d = d_next;
                }
#undef NRA_SUBST
                if (colmax > best) { best = colmax; bestj = j; }
                Hbot = h; Fout = F; F2out = F2;
                if (CHAIN) {
                    if (lane == 63 && !last_blk && j >= 0 && j < ncols) {
                        cout[j] = Hbot; cout[chain_cap + j] = Fout; cout[2 * chain_cap + j] = F2out;
                    }
                }
                ++j;
            }
        }
      }   // row blocks
        // wave reduce: max packed value, then the smallest column holding it
        CELL vmax = best;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = imax(vmax, (CELL)__shfl_xor(vmax, off, WAVE));
        int jm = (best == vmax) ? bestj : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) jm = imin(jm, __shfl_xor(jm, off, WAVE));
        if (lane == 0) {
            const long long sc = (long long)(vmax >> SH);
            const int lo = sp.min_score > 1 ? sp.min_score : 1;
            if (sc >= lo && jm >= 0 && jm != 0x7fffffff) {
                out_score[tk.out] = (int32_t)sc;
                out_p[tk.out] = (MODE == 0) ? (int32_t)(vmax & PMASK) : (int32_t)((vmax & PMASK) - WB);
                if (out_tend) out_tend[tk.out] = jm + 1;
            } else {
                out_score[tk.out] = -1;
                out_p[tk.out] = (MODE == 0) ? -1 : 0;
                if (out_tend) out_tend[tk.out] = -1;
            }
        }
    }
}

#endif  // parts 2, 3

#if NRA_HAS_PART(4)
// ------------------------------------------------------------------------------------
// selectors
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = imax(v, __shfl_xor(v, off, WAVE));
    return v;
}
__device__ __forceinline__ long long wave_sum64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// max AS per read; the records tied at it become tasks of the extents kernel
// (nanoRepeat_bam.py:423-425: only records with AS == top AS are ever examined)
__global__ __launch_bounds__(WAVE) void k_select_best_1d(int n_reads, const int32_t* __restrict__ kmin,
                                                        const int32_t* __restrict__ kmax,
                                                        const uint32_t* __restrict__ coff,
                                                        const int32_t* __restrict__ cand_score,
                                                        const uint8_t* __restrict__ cand_flag,
                                                        const int32_t* __restrict__ read_bucket,
                                                        const uint32_t* __restrict__ bucket_task_base,
                                                        int append_mode, NraTask* __restrict__ ext_tasks,
                                                        int32_t* __restrict__ ext_count,
                                                        int32_t* __restrict__ best_score)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_reads) return;
    const int k0 = kmin[r], K = kmax[r] - k0 + 1;
    if (K <= 0) { if (lane == 0) best_score[r] = -1; return; }
    const uint32_t base = coff[r];
    int best = -1;
    for (int c = lane; c < K; c += WAVE) best = imax(best, cand_score[base + c]);
    best = wave_max(best);
    if (lane == 0) best_score[r] = best;
    if (!append_mode || best < 0) return;
    const int b = read_bucket[r];
    for (int c = lane; c < K; c += WAVE) {
        // the explicit extents DP runs for every tie (mode 2) or only where the three-score
        // flank verdict of the sweep is ambiguous (mode 1, flag 2)
        if (cand_score[base + c] == best && (append_mode == 2 || cand_flag[base + c] == 2)) {
            const int slot = atomicAdd(&ext_count[b], 1);
            NraTask t; t.read = r; t.k1 = k0 + c; t.k2 = 0; t.out = (int)(base + c);
            ext_tasks[bucket_task_base[b] + slot] = t;
        }
    }
}

// flank test + tie sum (nanoRepeat_bam.py:426-433)
__global__ __launch_bounds__(WAVE) void k_select_final_1d(int n_reads, const int32_t* __restrict__ kmin,
                                                         const int32_t* __restrict__ kmax,
                                                         const uint32_t* __restrict__ coff,
                                                         const NraDevRead* __restrict__ reads,
                                                         const NraDevRegion* __restrict__ regions,
                                                         const int32_t* __restrict__ cand_score,
                                                         const uint8_t* __restrict__ cand_flag,
                                                         const int32_t* __restrict__ cand_tstart,
                                                         const int32_t* __restrict__ cand_tend,
                                                         const int32_t* __restrict__ best_score,
                                                         int64_t* __restrict__ sum_k,
                                                         int32_t* __restrict__ n_ties,
                                                         uint8_t* __restrict__ status)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_reads) return;
    const int k0 = kmin[r], K = kmax[r] - k0 + 1;
    if (K <= 0) { if (lane == 0) { sum_k[r] = 0; n_ties[r] = 0; status[r] = 3; } return; }
    const int best = best_score[r];
    if (best < 0) { if (lane == 0) { sum_k[r] = 0; n_ties[r] = 0; status[r] = 2; } return; }
    const NraDevRegion rg = regions[reads[r].region];
    const uint32_t base = coff[r];
    long long sk = 0; int nt = 0;
    for (int c = lane; c < K; c += WAVE) {
        if (cand_score[base + c] == best) {
            const int k = k0 + c;
            const int tlen = rg.l1 + rg.m1 * k + rg.l3;
            const int ts = cand_tstart[base + c], te = cand_tend[base + c];
            // explicit extents when the extents kernel ran for this record, else the sweep's verdict
            const bool pass = ts >= 0 ? (ts < rg.l1 && tlen - te < rg.l3) : (cand_flag[base + c] == 1);
            if (pass) { sk += k; ++nt; }
        }
    }
    sk = wave_sum64(sk);
    nt = (int)wave_sum64(nt);
    if (lane == 0) { sum_k[r] = sk; n_ties[r] = nt; status[r] = nt > 0 ? 0 : 1; }
}

// strand = the orientation with the higher DP score against the read's first cell
__global__ void k_pick_strand(int n_reads, const int32_t* __restrict__ probe_score,
                              const int8_t* __restrict__ strand_in, int8_t* __restrict__ strand_out,
                              NraDevRead* __restrict__ reads)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    int s = strand_in ? strand_in[r] : 0;
    if (s == 0) s = (probe_score[2 * r + 1] > probe_score[2 * r]) ? -1 : 1;
    strand_out[r] = (int8_t)s;
    reads[r].rc = s < 0 ? 1 : 0;
}

// max window score per read, ties -> sums of k1 / k2 (nanoRepeat_joint.py:458-476)
__global__ __launch_bounds__(WAVE) void k_select_2d(int n_reads, const uint32_t* __restrict__ cell_first,
                                                   const uint32_t* __restrict__ cell_cnt,
                                                   const int32_t* __restrict__ cell_k1,
                                                   const int32_t* __restrict__ cell_k2,
                                                   const NraGridRow* __restrict__ grid_rows, int step1, int step2,
                                                   const int32_t* __restrict__ cell_score,
                                                   const int32_t* __restrict__ cell_wscore,
                                                   int32_t* __restrict__ best_w, int64_t* __restrict__ sum_k1,
                                                   int64_t* __restrict__ sum_k2, int32_t* __restrict__ n_ties,
                                                   uint8_t* __restrict__ status)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t f = cell_first[r], n = cell_cnt[r];
    const int NONE = -0x7fffffff;
    int w = NONE;
    for (uint32_t c = lane; c < n; c += WAVE)
        if (cell_score[f + c] >= 0) w = imax(w, cell_wscore[f + c]);
    w = wave_max(w);
    if (w == NONE) {
        if (lane == 0) { best_w[r] = 0; sum_k1[r] = 0; sum_k2[r] = 0; n_ties[r] = 0; status[r] = 2; }
        return;
    }
    long long s1 = 0, s2 = 0; int nt = 0;
    if (grid_rows) {                          // a routed grid: the cell's k1 / k2 from the read's row (k1-major)
        const NraGridRow g = grid_rows[r];
        for (uint32_t c = lane; c < n; c += WAVE)
            if (cell_score[f + c] >= 0 && cell_wscore[f + c] == w) {
                const uint32_t i = c / (uint32_t)g.n2, j = c - i * (uint32_t)g.n2;
                s1 += g.k1lo + (int)i * step1; s2 += g.k2lo + (int)j * step2; ++nt;
            }
    } else
    for (uint32_t c = lane; c < n; c += WAVE)
        if (cell_score[f + c] >= 0 && cell_wscore[f + c] == w) { s1 += cell_k1[f + c]; s2 += cell_k2[f + c]; ++nt; }
    s1 = wave_sum64(s1); s2 = wave_sum64(s2); nt = (int)wave_sum64(nt);
    if (lane == 0) { best_w[r] = w; sum_k1[r] = s1; sum_k2[r] = s2; n_ties[r] = nt; status[r] = 0; }
}

#endif  // part 4

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
#if NRA_HAS_PART(1)
extern "C" int nra_launch_score_pk16(int R, int has_n, hipStream_t st, int n_tasks,
                                     const NraPairTask* tasks, const NraDevRead* reads,
                                     const NraDevRegion* regions, const uint8_t* pool,
                                     const uint32_t* q2bit, const uint32_t* qnmask,
                                     NraScoreParams sp, int32_t* out_score)
{
    if (n_tasks <= 0) return 0;
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_score_pk16<r, true><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, out_score); \
        else k_score_pk16<r, false><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, out_score);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}

#endif  // part 1

#if NRA_HAS_PART(2) || NRA_HAS_PART(3)
// wide = int64 cells: instantiated for NRA_WIDE_R_SMALL / NRA_WIDE_R_LARGE rows per lane (unchained) and
// NRA_CHAIN_R / NRA_CHAIN_R_TEST (chained); int32 cells: every R unchained, chained as above
template <int MODE>
static int launch_payload(int R, int has_n, int wide, hipStream_t st, int n_waves, const NraTask* tasks,
                          const int32_t* count, const NraDevRead* reads,
                          const NraDevRegion* regions, const uint8_t* pool, const uint32_t* q2bit,
                          const uint32_t* qnmask, NraScoreParams sp, int32_t* out_score,
                          int32_t* out_p, int32_t* out_tend, void* chain_buf = nullptr, int chain_cap = 0)
{
#define PARGS tasks, count, reads, regions, pool, q2bit, qnmask, sp, out_score, out_p, out_tend, chain_buf, chain_cap
#define LAUNCH(CELL, r, chain)                                                                      \
    do {                                                                                            \
        if (has_n) k_payload<CELL, r, true, MODE, chain><<<n_waves, WAVE, 0, st>>>(PARGS);          \
        else k_payload<CELL, r, false, MODE, chain><<<n_waves, WAVE, 0, st>>>(PARGS);               \
    } while (0)
    if (chain_buf) {   // row-block chaining: the instantiations long reads and the tests use
        if (R == NRA_CHAIN_R) { if (wide) LAUNCH(long long, NRA_CHAIN_R, true); else LAUNCH(int, NRA_CHAIN_R, true); }
        else if (R == NRA_CHAIN_R_TEST) { if (wide) LAUNCH(long long, NRA_CHAIN_R_TEST, true); else LAUNCH(int, NRA_CHAIN_R_TEST, true); }
        else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
    if (wide) {
        if (R == NRA_WIDE_R_SMALL) LAUNCH(long long, NRA_WIDE_R_SMALL, false);
        else if (R == NRA_WIDE_R_LARGE) LAUNCH(long long, NRA_WIDE_R_LARGE, false);
        else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
#define CASE(r) case r: LAUNCH(int, r, false); break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
#undef LAUNCH
#undef PARGS
    return (int)hipGetLastError();
}

#if NRA_HAS_PART(2)
extern "C" int nra_launch_payload_origin(int R, int has_n, hipStream_t st, int n_waves,
                                         const NraTask* tasks, const int32_t* count,
                                         const NraDevRead* reads, const NraDevRegion* regions,
                                         const uint8_t* pool, const uint32_t* q2bit, const uint32_t* qnmask,
                                         NraScoreParams sp, int32_t* out_score, int32_t* out_p,
                                         int32_t* out_tend, void* chain_buf, int chain_cap, int wide)
{
    if (n_waves <= 0) return 0;
    return launch_payload<0>(R, has_n, wide, st, n_waves, tasks, count, reads, regions, pool, q2bit, qnmask, sp, out_score, out_p, out_tend, chain_buf, chain_cap);
}
#endif
#if NRA_HAS_PART(3)
extern "C" int nra_launch_payload_window(int R, int has_n, hipStream_t st, int n_waves,
                                         const NraTask* tasks, const int32_t* count,
                                         const NraDevRead* reads, const NraDevRegion* regions,
                                         const uint8_t* pool, const uint32_t* q2bit, const uint32_t* qnmask,
                                         NraScoreParams sp, int32_t* out_score, int32_t* out_p,
                                         int32_t* out_tend, void* chain_buf, int chain_cap, int wide)
{
    if (n_waves <= 0) return 0;
    return launch_payload<1>(R, has_n, wide, st, n_waves, tasks, count, reads, regions, pool, q2bit, qnmask, sp, out_score, out_p, out_tend, chain_buf, chain_cap);
}
#endif
#endif  // parts 2, 3

#if NRA_HAS_PART(4)
extern "C" int nra_launch_select_best_1d(hipStream_t st, int n_reads, const int32_t* kmin, const int32_t* kmax,
                                         const uint32_t* coff, const int32_t* cand_score,
                                         const uint8_t* cand_flag,
                                         const int32_t* read_bucket, const uint32_t* bucket_task_base,
                                         int append_mode, NraTask* ext_tasks, int32_t* ext_count,
                                         int32_t* best_score)
{
    if (n_reads <= 0) return 0;
    k_select_best_1d<<<n_reads, WAVE, 0, st>>>(n_reads, kmin, kmax, coff, cand_score, cand_flag, read_bucket,
                                               bucket_task_base, append_mode, ext_tasks, ext_count, best_score);
    return (int)hipGetLastError();
}

extern "C" int nra_launch_select_final_1d(hipStream_t st, int n_reads, const int32_t* kmin, const int32_t* kmax,
                                          const uint32_t* coff, const NraDevRead* reads,
                                          const NraDevRegion* regions, const int32_t* cand_score,
                                          const uint8_t* cand_flag,
                                          const int32_t* cand_tstart, const int32_t* cand_tend,
                                          const int32_t* best_score, int64_t* sum_k, int32_t* n_ties,
                                          uint8_t* status)
{
    if (n_reads <= 0) return 0;
    k_select_final_1d<<<n_reads, WAVE, 0, st>>>(n_reads, kmin, kmax, coff, reads, regions, cand_score, cand_flag,
                                                cand_tstart, cand_tend, best_score, sum_k, n_ties, status);
    return (int)hipGetLastError();
}

extern "C" int nra_launch_pick_strand(hipStream_t st, int n_reads, const int32_t* probe_score,
                                      const int8_t* strand_in, int8_t* strand_out, NraDevRead* reads)
{
    if (n_reads <= 0) return 0;
    k_pick_strand<<<(n_reads + 255) / 256, 256, 0, st>>>(n_reads, probe_score, strand_in, strand_out, reads);
    return (int)hipGetLastError();
}

extern "C" int nra_launch_select_2d(hipStream_t st, int n_reads, const uint32_t* cell_first,
                                    const uint32_t* cell_cnt, const int32_t* cell_k1, const int32_t* cell_k2,
                                    const NraGridRow* grid_rows, int grid_step1, int grid_step2,
                                    const int32_t* cell_score, const int32_t* cell_wscore,
                                    int32_t* best_w, int64_t* sum_k1, int64_t* sum_k2, int32_t* n_ties,
                                    uint8_t* status)
{
    if (n_reads <= 0) return 0;
    k_select_2d<<<n_reads, WAVE, 0, st>>>(n_reads, cell_first, cell_cnt, cell_k1, cell_k2, grid_rows, grid_step1, grid_step2, cell_score,
                                          cell_wscore, best_w, sum_k1, sum_k2, n_ties, status);
    return (int)hipGetLastError();
}
#endif  // part 4
