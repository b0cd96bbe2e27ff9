/*
 * nr_decomp.c -- the junction decomposition of the 1D candidate bank on the CPU (scalar C, OpenMP over reads).
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY, like everything under oracle/.  It is NOT the oracle: nr_oracle.c scores K
 * full, independent DPs per read -- what the reference hands its aligner (nanoRepeat_bam.py:478-497) -- and stays the
 * independent checker.  This file is the SAME ALGORITHM the HIP sweeps run (nanorepeat_amd/csrc/nra_sweep.hip), so that
 * bench.py can report a GPU / CPU ratio that separates the hardware from the algorithm (`gpu_over_cpu_same_algorithm`),
 * and it is a third implementation the tests hold against the oracle (tests/test_oracle_golden.py).
 *
 * The K candidates L + unit^k + R of a read share L + unit^k as a prefix and R as a suffix; an optimal local alignment
 * against candidate k lies in L + unit^k (best score B_k), or in R (A), or consumes R[0] (S_k):
 *   reverse sweep : reversed read vs rev(R); at its last column (R[0]) every row keeps H, E_in, E2_in; A = max over cells
 *   forward sweep : read vs L + unit^kmax; at the last column of every L + unit^k each row r pairs with reverse row
 *                   a = q - 2 - r:  H_f + H_b,  E_f + E_b + q,  E2_f + E2_b + q2  (a gap across the junction is refunded
 *                   one open) -> S_k; the running maximum over all cells so far is B_k
 *   Score(k) = max(S_k, B_k, A).
 * The reference's flank test (nanoRepeat_bam.py:426-428) needs extents.  The left one is one bit carried with every
 * forward score (2 * score + bit, max = lexicographic): "the best path into this state starts at a column >= |L|"
 * (an alignment starting at column j enters with 0 | (j >= |L|)); an alignment inside R starts beyond L (2A + 1).
 *   tstart < |L|      <=>  the bit of V = max(S_k, B_k, 2A + 1) is 0   (the oracle reports the largest tstart)
 *   tend > |L| + m k  holds if B_k < Score, fails if B_k = Score > S_k; B_k = S_k = Score is decided by ONE explicit
 *                     alignment of that candidate (nro_align), as the HIP path does with its extents kernel.
 */
#include "nr_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG (INT_MIN / 4)
static inline int imax2(int a, int b) { return a > b ? a : b; }

static inline int subst2(uint8_t qc, uint8_t tc, int a2, int b2, int amb2)      /* doubled substitution score */
{
    if (qc >= 4 || tc >= 4) return -amb2;
    return qc == tc ? a2 : -b2;
}

/* Per-read scratch: 3 rows-arrays for the running sweep + 3 for the R side of the junction. */
typedef struct { int *H, *E, *E2, *Hb, *Eb, *E2b; } scratch_t;

int nrd_round3_1d(const nro_region_t* regions, int32_t n_regions,
                  int32_t n_reads, const char* seqs, const int64_t* seq_off,
                  const int32_t* read_region, const int32_t* kmin, const int32_t* kmax,
                  const nro_scoring_t* sc,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int64_t* executed_cells)
{
    if (!regions || n_regions <= 0 || n_reads < 0 || !sc) return -1;
    if (n_reads > 0 && (!seqs || !seq_off || !kmin || !kmax || !best_score || !sum_k || !n_ties || !status)) return -1;
    if (n_regions > 1 && !read_region && n_reads > 0) return -1;
    for (int32_t r = 0; r < n_reads; ++r) {
        if (kmin[r] > kmax[r]) continue;
        const int32_t g = read_region ? read_region[r] : 0;
        if (g < 0 || g >= n_regions || kmin[r] < 0 || seq_off[r + 1] < seq_off[r]) return -1;
        if (regions[g].left_len < 1 || regions[g].right_len < 1 || regions[g].unit_len < 1) return -1;   /* the junction needs a base on either side */
    }
    const int a2 = 2 * sc->match, b2 = 2 * sc->mismatch, amb2 = 2 * sc->sc_ambi;
    const int ext1 = 2 * sc->gap_ext1, opn1 = 2 * (sc->gap_open1 + sc->gap_ext1);
    const int ext2 = 2 * sc->gap_ext2, opn2 = 2 * (sc->gap_open2 + sc->gap_ext2);
    const int q1 = 2 * sc->gap_open1, q2 = 2 * sc->gap_open2;          /* the open refunded to a gap that spans the junction */
    const int lo = sc->min_dp_score > 1 ? sc->min_dp_score : 1;
    int64_t cells_total = 0;
    int bad = 0;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nro_get_threads()) reduction(+ : cells_total)
#endif
    for (int32_t r = 0; r < n_reads; ++r) {
        best_score[r] = 0; sum_k[r] = 0; n_ties[r] = 0;
        if (kmin[r] > kmax[r]) { status[r] = 3; continue; }
        const int32_t g = read_region ? read_region[r] : 0;
        const int32_t ll = regions[g].left_len, m = regions[g].unit_len, rl = regions[g].right_len;
        const int32_t q = (int32_t)(seq_off[r + 1] - seq_off[r]);
        const int32_t k0 = kmin[r], k1 = kmax[r], K = k1 - k0 + 1;
        if (q == 0) { status[r] = 2; continue; }
        uint8_t* qc = (uint8_t*)malloc((size_t)q + (size_t)ll + (size_t)m + (size_t)rl + 4);
        uint8_t* Lc = qc + q; uint8_t* Uc = Lc + ll; uint8_t* Rc = Uc + m;
        nro_encode(seqs + seq_off[r], q, qc);
        nro_encode(regions[g].left, ll, Lc); nro_encode(regions[g].unit, m, Uc); nro_encode(regions[g].right, rl, Rc);
        int* mem = (int*)malloc(sizeof(int) * 6 * (size_t)q);
        scratch_t s = {mem, mem + q, mem + 2 * (size_t)q, mem + 3 * (size_t)q, mem + 4 * (size_t)q, mem + 5 * (size_t)q};
        int32_t* Vk = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)K);      /* per candidate: S_k, B_k (doubled + bit) */
        int32_t* Sk = Vk; int32_t* Bk = Vk + K;

        /* ---- reverse sweep: reversed read (row a = read base q-1-a) vs rev(R) (column b = R[rl-1-b]) */
        for (int32_t i = 0; i < q; ++i) { s.H[i] = NEG; s.E[i] = NEG; s.E2[i] = NEG; }
        int A = 0;
        for (int32_t b = 0; b < rl; ++b) {
            const uint8_t tc = Rc[rl - 1 - b];
            const int last = b == rl - 1;
            int diag = NEG, F = NEG, F2 = NEG;
            for (int32_t a = 0; a < q; ++a) {
                const int d = imax2(diag, 0) + subst2(qc[q - 1 - a], tc, a2, b2, amb2);
                const int ein = s.E[a], e2in = s.E2[a];                   /* E(a, b): from column b - 1 */
                int h = imax2(imax2(d, ein), imax2(F, imax2(e2in, F2)));
                if (h > A) A = h;
                if (last) { s.Hb[a] = h; s.Eb[a] = ein; s.E2b[a] = e2in; }
                diag = s.H[a]; s.H[a] = h;
                s.E[a] = imax2(ein - ext1, h - opn1);
                s.E2[a] = imax2(e2in - ext2, h - opn2);
                F = imax2(F - ext1, h - opn1);
                F2 = imax2(F2 - ext2, h - opn2);
            }
        }
        cells_total += (int64_t)q * rl;

        /* ---- forward sweep: read vs L + unit^k1, values 2 * score + origin bit */
        for (int32_t i = 0; i < q; ++i) { s.H[i] = NEG; s.E[i] = NEG; s.E2[i] = NEG; }
        const int32_t ncols = ll + m * k1;
        int run = 0;                                                      /* running maximum over all cells so far */
        int32_t knext = k0;                                               /* next boundary: last column of L + unit^knext */
        for (int32_t j = 0; j < ncols; ++j) {
            const uint8_t tc = j < ll ? Lc[j] : Uc[(j - ll) % m];
            const int fresh = j >= ll ? 1 : 0;                            /* 0 | origin bit of an alignment that starts here */
            const int boundary = knext <= k1 && j == ll + m * knext - 1;  /* (k = 0 with an empty L cannot occur: ll >= 1) */
            int diag = NEG, F = NEG, F2 = NEG, S = NEG;
            for (int32_t i = 0; i < q; ++i) {
                const int d = imax2(diag, fresh) + subst2(qc[i], tc, a2, b2, amb2);
                const int ein = s.E[i], e2in = s.E2[i];
                int h = imax2(imax2(d, ein), imax2(F, imax2(e2in, F2)));
                if (h > run) run = h;
                if (boundary && i <= q - 2) {
                    const int32_t a = q - 2 - i;                           /* the reverse row of read base i + 1 */
                    const int t1 = h + s.Hb[a];
                    const int t2 = ein > NEG / 2 && s.Eb[a] > NEG / 2 ? ein + s.Eb[a] + q1 : NEG;
                    const int t3 = e2in > NEG / 2 && s.E2b[a] > NEG / 2 ? e2in + s.E2b[a] + q2 : NEG;
                    S = imax2(S, imax2(t1, imax2(t2, t3)));
                }
                diag = s.H[i]; s.H[i] = h;
                s.E[i] = imax2(ein - ext1, h - opn1);
                s.E2[i] = imax2(e2in - ext2, h - opn2);
                F = imax2(F - ext1, h - opn1);
                F2 = imax2(F2 - ext2, h - opn2);
            }
            if (boundary) { Sk[knext - k0] = S; Bk[knext - k0] = run; ++knext; }
        }
        /* (k = 0: the "last column of L" boundary; handled above since ll + m * 0 - 1 = ll - 1 >= 0) */
        cells_total += (int64_t)q * ncols;

        /* ---- Score(k), the flank verdict, the selector (nanoRepeat_bam.py:408-434) */
        int32_t smax = -1;
        for (int32_t c = 0; c < K; ++c) {
            const int V = imax2(imax2(Sk[c], Bk[c]), A + 1);
            const int best = V >> 1;
            if (best >= lo && best > smax) smax = best;
        }
        int64_t sk = 0; int32_t nt = 0;
        if (smax >= 0) {
            uint8_t* tgt = NULL;
            for (int32_t c = 0; c < K; ++c) {
                const int V = imax2(imax2(Sk[c], Bk[c]), A + 1);
                if ((V >> 1) != smax) continue;
                int pass = 1;
                if (V & 1) pass = 0;                                      /* an optimal alignment starts at a column >= |L| */
                else if ((Bk[c] >> 1) >= smax) {
                    if ((Sk[c] >> 1) >= smax) {
                        /* ambiguous: one explicit alignment of this candidate decides, as the oracle would */
                        const int32_t k = k0 + c, tl = ll + m * k + rl;
                        if (!tgt) tgt = (uint8_t*)malloc((size_t)ll + (size_t)m * k1 + rl + 1);
                        memcpy(tgt, Lc, (size_t)ll);
                        for (int32_t i = 0; i < k; ++i) memcpy(tgt + ll + (size_t)i * m, Uc, (size_t)m);
                        memcpy(tgt + ll + (size_t)m * k, Rc, (size_t)rl);
                        int32_t ts = 0, te = 0;
                        nro_align(qc, q, tgt, tl, sc, NRO_MODE_ORIGIN, 0, 0, &ts, &te);
                        pass = ts < ll && tl - te < rl;
                        cells_total += (int64_t)q * tl;
                    } else pass = 0;                                      /* every optimal alignment ends inside L + unit^k */
                }
                if (pass) { sk += k0 + c; ++nt; }
            }
            free(tgt);
        }
        if (smax < 0) status[r] = 2;
        else { best_score[r] = smax; sum_k[r] = sk; n_ties[r] = nt; status[r] = nt > 0 ? 0 : 1; }
        free(Vk); free(mem); free(qc);
    }
    if (executed_cells) *executed_cells = cells_total;
    return bad ? -1 : 0;
}
