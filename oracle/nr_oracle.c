/*
 * nr_oracle.c -- CPU restatement (oracle) of NanoRepeat's repeat-size scoring path.
 *
 * TEST INFRASTRUCTURE ONLY (see nr_oracle.h).  Citations are relative to the reference
 * tree (WGLab/NanoRepeat 1.8.3, src/NanoRepeat/).
 *
 * What is restated
 * ----------------
 *  - candidate bank  L + unit*k + R for k in [kmin,kmax]      nanoRepeat_bam.py:478-481
 *  - the aligner call `-x map-ont -f 0.0 -N 100 -c --eqx`     nanoRepeat_bam.py:495-497
 *    as an OPTIMAL local alignment under minimap2's map-ont objective
 *    (a=2 b=4 q=4 e=2 q2=24 e2=1 sc_ambi=1; ksw2 two-piece recurrences).
 *    minimap2 itself is seed-chain-extend with banding/z-drop; its scores are <= these.
 *    "aligner parity unpinned": pyminimap2 is absent offline.
 *  - 1D selector: max AS, ties, flank test, mean of k           nanoRepeat_bam.py:408-434
 *  - 2D template L + u1*k1 + mid + u2*k2 + R                    nanoRepeat_joint.py:499-505
 *  - 2D window score from the alignment path                    tk.py:435-500
 *    window [max(0,L-10), min(tlen, L+m1k1+mid+m2k2+10))        nanoRepeat_joint.py:445-448
 *  - 2D selector: max window score, ties, mean k1 / mean k2     nanoRepeat_joint.py:458-476
 *
 * Alignment definition (the contract the HIP kernels match bit for bit)
 * ---------------------------------------------------------------------
 * Every DP state carries a pair V = (S, P): S the alignment score, P a payload.  Pairs
 * are ordered lexicographically and every "max" below is that lexicographic max, so
 * among equal-score alternatives the one with the larger payload wins -- an
 * order-independent tie-break (max over integers is associative and commutative).
 *
 *   fresh(j) = (0, P0(j))                        an empty alignment about to start at column j
 *   d        = max(H(i-1,j-1), fresh(j)) + (s(i,j), pd(i,j))
 *   H(i,j)   = max(d, E(i,j), F(i,j), E2(i,j), F2(i,j))
 *   E(i,j+1) = max(E(i,j)  + (-e,  xe(j+1)),  H(i,j) + (-(q+e),   oe(j+1)))   gap in query (D)
 *   F(i+1,j) = max(F(i,j)  + (-e,  xf(j)),    H(i,j) + (-(q+e),   of(j)))     gap in target (I)
 *   E2/F2 likewise with (q2, e2).     Boundaries: H(-1,.) = H(.,-1) = E(.,0) = F(0,.) = -inf.
 *   best = max over all cells of H(i,j);  score = S(best);  tend = 1 + the smallest j
 *   whose column contains a cell equal to best.
 *
 * ORIGIN mode (1D): P0(j) = j and every payload increment is 0, so P(best) = tstart, and
 *   among co-optimal alignments the largest tstart is reported.
 * WINDOW mode (2D): P0(j) = 0 and the increments replay tk.py:435-500 on the path:
 *   pd = +2 / -4 for '=' / 'X' when wa <= j < wb                        (tk.py:464-475)
 *   D step onto target base j with wa <= j < wb: -4 if it opens the gap or j == wa
 *     (first overlapped base of the run), else -2                        (tk.py:480-485)
 *   I step at ref_pos = j+1 with wa < ref_pos < wb-1: -4 opening, -2 extending (tk.py:476-479)
 *   so P(best) is the window score of the co-optimal path with the highest window score.
 */
#include "nr_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PBIAS ((int64_t)1 << 30)
#define PACK(s, p) (((int64_t)(s) << 32) + (int64_t)(p))
#define NEGV (-((int64_t)1 << 60))

static int g_threads = 0;

void nro_set_threads(int n) { g_threads = n; }
int nro_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

void nro_default_scoring(nro_scoring_t* sc)
{
    sc->match = 2; sc->mismatch = 4;
    sc->gap_open1 = 4; sc->gap_ext1 = 2;
    sc->gap_open2 = 24; sc->gap_ext2 = 1;
    sc->sc_ambi = 1; sc->min_dp_score = 80;
}

void nro_encode(const char* s, int64_t n, uint8_t* out)
{
    for (int64_t i = 0; i < n; ++i) {
        switch (s[i]) {
        case 'A': case 'a': out[i] = 0; break;
        case 'C': case 'c': out[i] = 1; break;
        case 'G': case 'g': out[i] = 2; break;
        case 'T': case 't': case 'U': case 'u': out[i] = 3; break;
        default: out[i] = 4;
        }
    }
}

void nro_revcomp_codes(const uint8_t* in, int64_t n, uint8_t* out)
{
    for (int64_t i = 0; i < n; ++i) {
        uint8_t c = in[n - 1 - i];
        out[i] = c < 4 ? (uint8_t)(3 - c) : c;
    }
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* trace matrices (optional): value of each state AT cell (i,j), row-major [i*tlen + j] */
typedef struct {
    int64_t *H, *E, *F, *E2, *F2;
} trace_t;

typedef struct {
    int64_t best;   /* packed (S, P) */
    int32_t best_j; /* smallest column holding best */
    int32_t best_i; /* smallest row in that column */
} dp_result_t;

static void dp_core(const uint8_t* q, int32_t ql, const uint8_t* t, int32_t tl,
                    const nro_scoring_t* sc, int mode, int32_t wa, int32_t wb,
                    dp_result_t* res, trace_t* tr)
{
    res->best = PACK(0, 0); res->best_j = -1; res->best_i = -1;
    if (ql <= 0 || tl <= 0) return;

    int64_t* Hp = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)ql);
    int64_t* E = Hp + ql;
    int64_t* E2 = E + ql;
    for (int32_t i = 0; i < ql; ++i) { Hp[i] = NEGV; E[i] = NEGV; E2[i] = NEGV; }

    const int64_t a = sc->match, b = sc->mismatch, amb = sc->sc_ambi;
    const int64_t ext1 = sc->gap_ext1, opn1 = sc->gap_open1 + sc->gap_ext1;
    const int64_t ext2 = sc->gap_ext2, opn2 = sc->gap_open2 + sc->gap_ext2;
    const int win = (mode == NRO_MODE_WINDOW);

    int64_t best = PACK(0, win ? PBIAS : 0);
    int32_t best_j = -1, best_i = -1;

    for (int32_t j = 0; j < tl; ++j) {
        const uint8_t tc = t[j];
        const int64_t fresh = win ? PACK(0, PBIAS) : PACK(0, j);
        /* payload increments for this column (all 0 in ORIGIN mode) */
        int64_t pd_eq = 0, pd_ne = 0, pe_open = 0, pe_ext = 0, pf_open = 0, pf_ext = 0;
        if (win) {
            if (j >= wa && j < wb) { pd_eq = 2; pd_ne = -4; }
            const int32_t jn = j + 1;             /* E update below creates E(i, j+1) */
            if (jn >= wa && jn < wb) { pe_open = -4; pe_ext = (jn == wa) ? -4 : -2; }
            const int32_t ref_pos = j + 1;        /* F consumes a query base at ref_pos */
            if (ref_pos > wa && ref_pos < wb - 1) { pf_open = -4; pf_ext = -2; }
        }
        const int64_t e_open1 = PACK(-opn1, pe_open), e_ext1 = PACK(-ext1, pe_ext);
        const int64_t e_open2 = PACK(-opn2, pe_open), e_ext2 = PACK(-ext2, pe_ext);
        const int64_t f_open1 = PACK(-opn1, pf_open), f_ext1 = PACK(-ext1, pf_ext);
        const int64_t f_open2 = PACK(-opn2, pf_open), f_ext2 = PACK(-ext2, pf_ext);

        int64_t diag = NEGV, F = NEGV, F2 = NEGV;
        for (int32_t i = 0; i < ql; ++i) {
            const uint8_t qc = q[i];
            int64_t s, pd;
            if (qc >= 4 || tc >= 4) s = -amb; else s = (qc == tc) ? a : -b;
            pd = (qc == tc) ? pd_eq : pd_ne;
            const int64_t dsrc = max64(diag, fresh);
            const int64_t d = dsrc + PACK(s, pd);
            int64_t h = max64(d, E[i]);
            h = max64(h, F);
            h = max64(h, E2[i]);
            h = max64(h, F2);
            if (tr) {
                size_t o = (size_t)i * tl + j;
                tr->H[o] = h; tr->E[o] = E[i]; tr->F[o] = F; tr->E2[o] = E2[i]; tr->F2[o] = F2;
            }
            if (h > best) { best = h; best_j = j; best_i = i; }
            diag = Hp[i]; Hp[i] = h;
            E[i]  = max64(E[i]  + e_ext1, h + e_open1);
            E2[i] = max64(E2[i] + e_ext2, h + e_open2);
            F     = max64(F     + f_ext1, h + f_open1);
            F2    = max64(F2    + f_ext2, h + f_open2);
        }
    }
    free(Hp);
    res->best = best; res->best_j = best_j; res->best_i = best_i;
}

static inline int32_t unpack_s(int64_t v) { return (int32_t)(v >> 32); }
static inline int64_t unpack_p(int64_t v) { return v - ((int64_t)unpack_s(v) << 32); }

int32_t nro_align(const uint8_t* q, int32_t qlen, const uint8_t* t, int32_t tlen,
                  const nro_scoring_t* sc, int mode, int32_t wa, int32_t wb,
                  int32_t* payload, int32_t* tend)
{
    dp_result_t r;
    dp_core(q, qlen, t, tlen, sc, mode, wa, wb, &r, NULL);
    int32_t s = unpack_s(r.best);
    if (s <= 0 || r.best_j < 0) {
        if (payload) *payload = 0;
        if (tend) *tend = 0;
        return 0;
    }
    int64_t p = unpack_p(r.best);
    if (mode == NRO_MODE_WINDOW) p -= PBIAS;
    if (payload) *payload = (int32_t)p;
    if (tend) *tend = r.best_j + 1;
    return s;
}

/* ---- traceback (oracle-only: validates the WINDOW payload against the reference's
 *      CIGAR rescoring, and gives PAF-style extents) ------------------------------- */
typedef struct { char* buf; int32_t cap, n; char last; int32_t run; int ok; } cig_t;

static void cig_flush(cig_t* c)
{
    if (c->run == 0) return;
    char tmp[24];
    int m = snprintf(tmp, sizeof tmp, "%d%c", c->run, c->last);
    if (c->n + m + 1 > c->cap) { c->ok = 0; c->run = 0; return; }
    memcpy(c->buf + c->n, tmp, (size_t)m);
    c->n += m; c->run = 0;
}
static void cig_push(cig_t* c, char op)
{
    if (c->run > 0 && c->last == op) { c->run++; return; }
    cig_flush(c);
    c->last = op; c->run = 1;
}

int32_t nro_align_cigar(const uint8_t* q, int32_t qlen, const uint8_t* t, int32_t tlen,
                        const nro_scoring_t* sc, int mode, int32_t wa, int32_t wb,
                        char* cigar, int32_t cap,
                        int32_t* tstart, int32_t* tend, int32_t* qstart, int32_t* qend,
                        int32_t* payload)
{
    if (cap > 0) cigar[0] = 0;
    if (qlen <= 0 || tlen <= 0) return 0;
    size_t n = (size_t)qlen * tlen;
    trace_t tr;
    tr.H = (int64_t*)malloc(sizeof(int64_t) * 5 * n);
    tr.E = tr.H + n; tr.F = tr.E + n; tr.E2 = tr.F + n; tr.F2 = tr.E2 + n;
    dp_result_t r;
    dp_core(q, qlen, t, tlen, sc, mode, wa, wb, &r, &tr);
    int32_t score = unpack_s(r.best);
    if (score <= 0 || r.best_j < 0) { free(tr.H); return 0; }

    const int win = (mode == NRO_MODE_WINDOW);
    const int64_t a = sc->match, b = sc->mismatch, amb = sc->sc_ambi;
    const int64_t ext1 = sc->gap_ext1, opn1 = sc->gap_open1 + sc->gap_ext1;
    const int64_t ext2 = sc->gap_ext2, opn2 = sc->gap_open2 + sc->gap_ext2;

    /* reversed op list */
    char* ops = (char*)malloc((size_t)qlen + tlen + 2);
    int32_t nops = 0;
    int32_t i = r.best_i, j = r.best_j;
    int st = 0;                       /* 0 H, 1 E, 2 F, 3 E2, 4 F2 */
    int64_t V = r.best;
    int32_t ts = -1, qs = -1;
    for (;;) {
        size_t o = (size_t)i * tlen + j;
        if (st == 0) {
            /* which input of H(i,j) equals V?  order: d, E, F, E2, F2 (all co-optimal) */
            const uint8_t qc = q[i], tc = t[j];
            int64_t s = (qc >= 4 || tc >= 4) ? -amb : (qc == tc ? a : -b);
            int64_t pd = 0;
            if (win && j >= wa && j < wb) pd = (qc == tc) ? 2 : -4;
            int64_t fresh = win ? PACK(0, PBIAS) : PACK(0, j);
            int64_t hd = (i > 0 && j > 0) ? tr.H[o - tlen - 1] : NEGV;
            int64_t dsrc = max64(hd, fresh);
            if (dsrc + PACK(s, pd) == V) {
                ops[nops++] = (qc == tc) ? '=' : 'X';
                if (hd >= fresh) { V = hd; --i; --j; continue; }
                ts = j; qs = i; break;
            }
            if (tr.E[o] == V) { st = 1; continue; }
            if (tr.F[o] == V) { st = 2; continue; }
            if (tr.E2[o] == V) { st = 3; continue; }
            if (tr.F2[o] == V) { st = 4; continue; }
            fprintf(stderr, "nr_oracle: traceback lost at H(%d,%d)\n", i, j);
            abort();
        } else if (st == 1 || st == 3) {
            /* E(i,j): target base j against a gap */
            int64_t pe_open = 0, pe_ext = 0;
            if (win && j >= wa && j < wb) { pe_open = -4; pe_ext = (j == wa) ? -4 : -2; }
            const int64_t* Em = (st == 1) ? tr.E : tr.E2;
            int64_t ext = (st == 1) ? ext1 : ext2, opn = (st == 1) ? opn1 : opn2;
            ops[nops++] = 'D';
            if (j == 0) { fprintf(stderr, "nr_oracle: E at column 0\n"); abort(); }
            if (Em[o - 1] + PACK(-ext, pe_ext) == V) { V = Em[o - 1]; --j; continue; }
            if (tr.H[o - 1] + PACK(-opn, pe_open) == V) { V = tr.H[o - 1]; --j; st = 0; continue; }
            fprintf(stderr, "nr_oracle: traceback lost at E(%d,%d)\n", i, j);
            abort();
        } else {
            /* F(i,j): query base i against a gap, ref_pos = j+1 */
            int64_t pf_open = 0, pf_ext = 0;
            if (win && (j + 1) > wa && (j + 1) < wb - 1) { pf_open = -4; pf_ext = -2; }
            const int64_t* Fm = (st == 2) ? tr.F : tr.F2;
            int64_t ext = (st == 2) ? ext1 : ext2, opn = (st == 2) ? opn1 : opn2;
            ops[nops++] = 'I';
            if (i == 0) { fprintf(stderr, "nr_oracle: F at row 0\n"); abort(); }
            if (Fm[o - tlen] + PACK(-ext, pf_ext) == V) { V = Fm[o - tlen]; --i; continue; }
            if (tr.H[o - tlen] + PACK(-opn, pf_open) == V) { V = tr.H[o - tlen]; --i; st = 0; continue; }
            fprintf(stderr, "nr_oracle: traceback lost at F(%d,%d)\n", i, j);
            abort();
        }
    }
    cig_t c = { cigar, cap, 0, 0, 0, 1 };
    for (int32_t k = nops - 1; k >= 0; --k) cig_push(&c, ops[k]);
    cig_flush(&c);
    if (c.ok && c.n < cap) cigar[c.n] = 0; else c.ok = 0;
    free(ops); free(tr.H);
    if (!c.ok) return -1;
    if (tstart) *tstart = ts;
    if (tend) *tend = r.best_j + 1;
    if (qstart) *qstart = qs;
    if (qend) *qend = r.best_i + 1;
    if (payload) {
        int64_t p = unpack_p(r.best);
        if (win) p -= PBIAS;
        *payload = (int32_t)p;
    }
    return score;
}

/* ---- tk.py:368-373 compute_overlap_len ------------------------------------------ */
static inline int32_t overlap_len(int32_t s1, int32_t e1, int32_t s2, int32_t e2)
{
    int32_t ms = s1 > s2 ? s1 : s2;
    int32_t me = e1 < e2 ? e1 : e2;
    int32_t o = me - ms;
    return o > 0 ? o : 0;
}

/* ---- tk.py:435-500 target_region_alignment_stats_from_cigar (+ :380-401 parser) --- */
int32_t nro_cigar_region_score(const char* cigar, int32_t tstart, int32_t tend,
                               int32_t rs, int32_t re,
                               int32_t* num_match, int32_t* num_mismatch,
                               int32_t* num_ins, int32_t* num_del)
{
    const int32_t matching_score = 2, mismatching_penalty = -4;
    const int32_t gap_open_penalty = -4, gap_ext_penalty = -2;      /* tk.py:444-447 */
    int32_t nm = 0, nx = 0, ni = 0, nd = 0, score = 0;
    int32_t pos = tstart;
    if (!cigar || !cigar[0]) return INT32_MIN;
    const char* p = cigar;
    while (*p) {
        if (*p < '0' || *p > '9') return INT32_MIN;
        int64_t len = 0;
        while (*p >= '0' && *p <= '9') { len = len * 10 + (*p - '0'); ++p; }
        char op = *p++;
        int32_t l = (int32_t)len, ov;
        switch (op) {
        case '=':
            ov = overlap_len(pos, pos + l, rs, re);
            if (ov > 0) { nm += ov; score += ov * matching_score; }
            pos += l; break;
        case 'X':
            ov = overlap_len(pos, pos + l, rs, re);
            if (ov > 0) { nx += ov; score += ov * mismatching_penalty; }
            pos += l; break;
        case 'I':
            if (pos > rs && pos < re - 1) { ni += l; score += gap_open_penalty + (l - 1) * gap_ext_penalty; }
            break;
        case 'D':
            ov = overlap_len(pos, pos + l, rs, re);
            if (ov > 0) { nd += ov; score += gap_open_penalty + (ov - 1) * gap_ext_penalty; }
            pos += l; break;
        case 'S':
            continue;                 /* tk.py:486-487: skips the break test below too */
        default:
            return INT32_MIN;         /* tk.py:488-490 */
        }
        if (pos > re) break;          /* tk.py:491 */
    }
    if (tend < re) nx += re - tend;   /* tk.py:493-497: counters only, never the score */
    if (tstart > rs) nx += tstart - rs;
    if (num_match) *num_match = nm;
    if (num_mismatch) *num_mismatch = nx;
    if (num_ins) *num_ins = ni;
    if (num_del) *num_del = nd;
    return score;
}

/* ---- 1D ------------------------------------------------------------------------- */
static int check_scoring(const nro_scoring_t* sc)
{
    return sc && sc->match > 0 && sc->mismatch >= 0 && sc->gap_ext1 > 0 && sc->gap_ext2 > 0 &&
           sc->gap_open1 >= 0 && sc->gap_open2 >= 0 && sc->sc_ambi >= 0;
}

/* The 1D selector on one read's PAF records (nanoRepeat_bam.py:408-434): record i = (k, AS, tstart, tend, tlen); a
 * score < 0 is "no record" (below min_dp_score).  Sorted by AS descending, the records tied with the best AS that
 * reach into both flanks (tstart < |L| and tlen - tend < |R|, :426-428) give sum_k / n_ties (their mean is the
 * repeat size, :431); a best record that fails the flank test ends the loop: lower scores are never looked at (:425).
 * Returns the read status: 0 = size from ties, 1 = keep the round-2 size (:433), 2 = no record at all (:418). */
int nro_select_1d(int32_t n, const int32_t* k, const int32_t* score, const int32_t* tstart, const int32_t* tend,
                  const int32_t* tlen, int32_t left_len, int32_t right_len,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties)
{
    int32_t smax = -1;
    for (int32_t i = 0; i < n; ++i) if (score[i] > smax) smax = score[i];
    int64_t sk = 0; int32_t nt = 0;
    if (smax >= 0)
        for (int32_t i = 0; i < n; ++i)
            if (score[i] == smax && tstart[i] < left_len && tlen[i] - tend[i] < right_len) { sk += k[i]; ++nt; }
    if (best_score) *best_score = smax;
    if (sum_k) *sum_k = smax < 0 ? 0 : sk;
    if (n_ties) *n_ties = smax < 0 ? 0 : nt;
    return smax < 0 ? 2 : (nt > 0 ? 0 : 1);
}

int nro_round3_1d(const nro_region_t* regions, int32_t n_regions,
                  int32_t n_reads, const char* seqs, const int64_t* seq_off,
                  const int32_t* read_region,
                  const int32_t* kmin, const int32_t* kmax,
                  const nro_scoring_t* sc, int32_t flags,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend)
{
    if (!regions || n_regions <= 0 || n_reads < 0 || !check_scoring(sc)) return -1;
    if (n_reads > 0 && (!seqs || !seq_off || !kmin || !kmax || !best_score || !sum_k || !n_ties || !status)) return -1;
    if (n_regions > 1 && !read_region && n_reads > 0) return -1;
    /* bad input (a read of an unknown region, a negative k) is refused before anything is allocated or run */
    for (int32_t r = 0; r < n_reads; ++r) {
        if (kmin[r] > kmax[r]) continue;
        const int32_t g = read_region ? read_region[r] : 0;
        if (g < 0 || g >= n_regions || kmin[r] < 0 || seq_off[r + 1] < seq_off[r]) return -1;
    }

    /* candidate offsets */
    int64_t* coff = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_reads + 1));
    coff[0] = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        int64_t K = (int64_t)kmax[r] - kmin[r] + 1;
        coff[r + 1] = coff[r] + (K > 0 ? K : 0);
    }
    /* encoded regions */
    uint8_t** Lc = (uint8_t**)calloc((size_t)n_regions, sizeof(uint8_t*));
    uint8_t** Uc = (uint8_t**)calloc((size_t)n_regions, sizeof(uint8_t*));
    uint8_t** Rc = (uint8_t**)calloc((size_t)n_regions, sizeof(uint8_t*));
    for (int32_t g = 0; g < n_regions; ++g) {
        Lc[g] = (uint8_t*)malloc((size_t)regions[g].left_len + 1);
        Uc[g] = (uint8_t*)malloc((size_t)regions[g].unit_len + 1);
        Rc[g] = (uint8_t*)malloc((size_t)regions[g].right_len + 1);
        nro_encode(regions[g].left, regions[g].left_len, Lc[g]);
        nro_encode(regions[g].unit, regions[g].unit_len, Uc[g]);
        nro_encode(regions[g].right, regions[g].right_len, Rc[g]);
    }
    const int all_ext = flags & 1;
    int bad = 0;

    /* phase 1: every (read, k) candidate is an independent alignment -- K full DPs per read, exactly what
     * the reference hands its aligner (nanoRepeat_bam.py:478-497).  Parallel over candidates, so that a
     * single long read with a wide window still uses every core. */
    const int64_t total = coff[n_reads];
    int32_t* S_all = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(total > 0 ? total : 1));
    int32_t* TS_all = S_all + (total > 0 ? total : 1);
    int32_t* TE_all = TS_all + (total > 0 ? total : 1);
    int32_t* owner = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));   /* filled for every candidate below */
    uint8_t** Q = (uint8_t**)calloc((size_t)(n_reads > 0 ? n_reads : 1), sizeof(uint8_t*));
    for (int32_t r = 0; r < n_reads; ++r) {
        best_score[r] = 0; sum_k[r] = 0; n_ties[r] = 0;
        if (kmin[r] > kmax[r]) { status[r] = 3; continue; }
        status[r] = 0;
        const int32_t ql = (int32_t)(seq_off[r + 1] - seq_off[r]);
        Q[r] = (uint8_t*)malloc((size_t)ql + 1);
        nro_encode(seqs + seq_off[r], ql, Q[r]);
        for (int64_t c = coff[r]; c < coff[r + 1]; ++c) owner[c] = r;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nro_get_threads())
#endif
    for (int64_t c = 0; c < total; ++c) {
        const int32_t r = owner[c];
        if (!Q[r]) continue;                                 /* skipped / invalid read */
        const int32_t g = read_region ? read_region[r] : 0;
        const int32_t ll = regions[g].left_len, ul = regions[g].unit_len, rl = regions[g].right_len;
        const int32_t ql = (int32_t)(seq_off[r + 1] - seq_off[r]);
        const int32_t k = kmin[r] + (int32_t)(c - coff[r]);
        /* template k: left + unit*k + right (nanoRepeat_bam.py:479) */
        const int32_t tl = ll + ul * k + rl;
        uint8_t* tgt = (uint8_t*)malloc((size_t)tl + 1);
        memcpy(tgt, Lc[g], (size_t)ll);
        for (int32_t i = 0; i < k; ++i) memcpy(tgt + ll + (size_t)i * ul, Uc[g], (size_t)ul);
        memcpy(tgt + ll + (size_t)ul * k, Rc[g], (size_t)rl);
        int32_t ts = 0, te = 0;
        int32_t sc1 = nro_align(Q[r], ql, tgt, tl, sc, NRO_MODE_ORIGIN, 0, 0, &ts, &te);
        if (sc1 < sc->min_dp_score || sc1 <= 0) { S_all[c] = -1; TS_all[c] = -1; TE_all[c] = -1; }
        else { S_all[c] = sc1; TS_all[c] = ts; TE_all[c] = te; }
        free(tgt);
    }
    /* phase 2: the selector, nanoRepeat_bam.py:423-433 (nro_select_1d below) */
    int32_t* krec = (int32_t*)malloc(sizeof(int32_t) * 2 * 8);
    int32_t krec_cap = 8;
    for (int32_t r = 0; r < n_reads; ++r) {
        if (!Q[r]) continue;
        const int32_t g = read_region ? read_region[r] : 0;
        const int32_t ll = regions[g].left_len, ul = regions[g].unit_len, rl = regions[g].right_len;
        const int32_t k0 = kmin[r], k1 = kmax[r];
        const int32_t K = k1 - k0 + 1;
        const int32_t* S = S_all + coff[r]; const int32_t* TS = TS_all + coff[r]; const int32_t* TE = TE_all + coff[r];
        /* the PAF records of the read: one per candidate that reached min_dp_score (tname = k, tlen = |L| + m k + |R|) */
        if (K > krec_cap) { krec_cap = K; krec = (int32_t*)realloc(krec, sizeof(int32_t) * 2 * (size_t)krec_cap); }
        int32_t* rk = krec; int32_t* rtl = krec + krec_cap;
        for (int32_t c = 0; c < K; ++c) { rk[c] = k0 + c; rtl[c] = ll + ul * (k0 + c) + rl; }
        int32_t smax = -1;
        status[r] = (uint8_t)nro_select_1d(K, rk, S, TS, TE, rtl, ll, rl, &smax, &sum_k[r], &n_ties[r]);
        best_score[r] = smax < 0 ? 0 : smax;
        for (int32_t c = 0; c < K; ++c) {
            if (cand_score) cand_score[coff[r] + c] = S[c];
            int keep = all_ext || (smax >= 0 && S[c] == smax);
            if (cand_tstart) cand_tstart[coff[r] + c] = keep ? TS[c] : -1;
            if (cand_tend) cand_tend[coff[r] + c] = keep ? TE[c] : -1;
        }
        free(Q[r]);
    }
    free(Q); free(owner); free(S_all); free(krec);
    for (int32_t g = 0; g < n_regions; ++g) { free(Lc[g]); free(Uc[g]); free(Rc[g]); }
    free(Lc); free(Uc); free(Rc); free(coff);
    return bad ? -1 : 0;
}

/* ---- generic pairs (anchor finding nanoRepeat_bam.py:281, round 2 :362) ------------ */
int nro_align_pairs(int32_t n_seqs, const char* seqs, const int64_t* seq_off,
                    int64_t n_pairs, const int32_t* pair_query, const int32_t* pair_target,
                    const nro_scoring_t* sc, int32_t flags,
                    int32_t* score, int32_t* tstart, int32_t* tend)
{
    (void)flags;
    if (n_seqs < 0 || n_pairs < 0 || !check_scoring(sc)) return -1;
    if (n_pairs > 0 && (!seqs || !seq_off || !pair_query || !pair_target || !score || !tstart || !tend)) return -1;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (pair_query[i] < 0 || pair_query[i] >= n_seqs || pair_target[i] < 0 || pair_target[i] >= n_seqs) return -1;
    uint8_t* codes = (uint8_t*)malloc((size_t)seq_off[n_seqs] + 1);
    nro_encode(seqs, seq_off[n_seqs], codes);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nro_get_threads())
#endif
    for (int64_t i = 0; i < n_pairs; ++i) {
        const int32_t q = pair_query[i], t = pair_target[i];
        int32_t ts = 0, te = 0;
        int32_t s = nro_align(codes + seq_off[q], (int32_t)(seq_off[q + 1] - seq_off[q]),
                              codes + seq_off[t], (int32_t)(seq_off[t + 1] - seq_off[t]),
                              sc, NRO_MODE_ORIGIN, 0, 0, &ts, &te);
        if (s < sc->min_dp_score || s <= 0) { score[i] = -1; tstart[i] = -1; tend[i] = -1; }
        else { score[i] = s; tstart[i] = ts; tend[i] = te; }
    }
    free(codes);
    return 0;
}

/* ---- 2D ------------------------------------------------------------------------- */
static int32_t build_joint_template(uint8_t* tgt, const uint8_t* L, int32_t ll,
                                    const uint8_t* U1, int32_t m1, int32_t k1,
                                    const uint8_t* M, int32_t ml,
                                    const uint8_t* U2, int32_t m2, int32_t k2,
                                    const uint8_t* R, int32_t rl)
{
    /* nanoRepeat_joint.py:503 */
    int32_t n = 0;
    memcpy(tgt, L, (size_t)ll); n += ll;
    for (int32_t k = 0; k < k1; ++k) { memcpy(tgt + n, U1, (size_t)m1); n += m1; }
    memcpy(tgt + n, M, (size_t)ml); n += ml;
    for (int32_t k = 0; k < k2; ++k) { memcpy(tgt + n, U2, (size_t)m2); n += m2; }
    memcpy(tgt + n, R, (size_t)rl); n += rl;
    return n;
}

int nro_joint_2d(const nro_joint_region_t* reg,
                 int32_t n_reads, const char* seqs, const int64_t* seq_off,
                 int8_t* read_strand,
                 int64_t n_cells, const int32_t* cell_read,
                 const int32_t* cell_k1, const int32_t* cell_k2,
                 const nro_scoring_t* sc, int32_t flags,
                 int32_t* cell_score, int32_t* cell_wscore,
                 int32_t* best_wscore, int64_t* sum_k1, int64_t* sum_k2,
                 int32_t* n_ties, uint8_t* status)
{
    (void)flags;
    if (!reg || n_reads < 0 || n_cells < 0 || !check_scoring(sc)) return -1;
    if (n_reads > 0 && (!seqs || !seq_off || !best_wscore || !sum_k1 || !sum_k2 || !n_ties || !status)) return -1;
    if (n_cells > 0 && (!cell_read || !cell_k1 || !cell_k2)) return -1;
    const int32_t ll = reg->left_len, m1 = reg->unit1_len, ml = reg->mid_len, m2 = reg->unit2_len, rl = reg->right_len;
    uint8_t* L = (uint8_t*)malloc((size_t)ll + 1); nro_encode(reg->left, ll, L);
    uint8_t* U1 = (uint8_t*)malloc((size_t)m1 + 1); nro_encode(reg->unit1, m1, U1);
    uint8_t* M = (uint8_t*)malloc((size_t)ml + 1); nro_encode(reg->mid, ml, M);
    uint8_t* U2 = (uint8_t*)malloc((size_t)m2 + 1); nro_encode(reg->unit2, m2, U2);
    uint8_t* R = (uint8_t*)malloc((size_t)rl + 1); nro_encode(reg->right, rl, R);

    /* cells must be grouped by read: first/last cell index per read */
    int64_t* first = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_reads + 1));
    int64_t* cnt = (int64_t*)calloc((size_t)n_reads + 1, sizeof(int64_t));
    int bad = 0;
    for (int32_t r = 0; r < n_reads; ++r) first[r] = -1;
    for (int64_t c = 0; c < n_cells; ++c) {
        int32_t r = cell_read[c];
        if (r < 0 || r >= n_reads || (c > 0 && r < cell_read[c - 1]) || cell_k1[c] < 0 || cell_k2[c] < 0) { bad = 1; break; }
        if (first[r] < 0) first[r] = c;
        cnt[r]++;
    }
    if (bad) { free(L); free(U1); free(M); free(U2); free(R); free(first); free(cnt); return -1; }
    int32_t k1max = 0, k2max = 0;
    for (int64_t c = 0; c < n_cells; ++c) {
        if (cell_k1[c] > k1max) k1max = cell_k1[c];
        if (cell_k2[c] > k2max) k2max = cell_k2[c];
    }
    const size_t tcap = (size_t)ll + (size_t)m1 * k1max + ml + (size_t)m2 * k2max + rl + 1;

#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nro_get_threads())
#endif
    for (int32_t r = 0; r < n_reads; ++r) {
        best_wscore[r] = 0; sum_k1[r] = 0; sum_k2[r] = 0; n_ties[r] = 0; status[r] = 2;
        if (cnt[r] == 0) { if (read_strand && read_strand[r] == 0) read_strand[r] = 1; continue; }
        const int32_t ql = (int32_t)(seq_off[r + 1] - seq_off[r]);
        uint8_t* qf = (uint8_t*)malloc((size_t)ql * 2 + 2);
        uint8_t* qr = qf + ql + 1;
        nro_encode(seqs + seq_off[r], ql, qf);
        nro_revcomp_codes(qf, ql, qr);
        uint8_t* tgt = (uint8_t*)malloc(tcap);
        int8_t strand = read_strand ? read_strand[r] : 0;
        if (strand == 0) {
            /* both strands against the read's first listed cell; higher DP score wins, ties '+' */
            const int64_t c = first[r];
            int32_t tl = build_joint_template(tgt, L, ll, U1, m1, cell_k1[c], M, ml, U2, m2, cell_k2[c], R, rl);
            int32_t sf = nro_align(qf, ql, tgt, tl, sc, NRO_MODE_ORIGIN, 0, 0, NULL, NULL);
            int32_t sr = nro_align(qr, ql, tgt, tl, sc, NRO_MODE_ORIGIN, 0, 0, NULL, NULL);
            strand = (sr > sf) ? -1 : 1;
        }
        if (read_strand) read_strand[r] = strand;
        const uint8_t* q = strand > 0 ? qf : qr;
        int32_t wmax = INT32_MIN; int64_t s1 = 0, s2 = 0; int32_t nt = 0;
        for (int64_t c = first[r]; c < first[r] + cnt[r]; ++c) {
            const int32_t k1 = cell_k1[c], k2 = cell_k2[c];
            int32_t tl = build_joint_template(tgt, L, ll, U1, m1, k1, M, ml, U2, m2, k2, R, rl);
            /* nanoRepeat_joint.py:445-448 */
            int32_t wa = ll - 10; if (wa < 0) wa = 0;
            int32_t wb = ll + m1 * k1 + ml + m2 * k2 + 10; if (wb > tl) wb = tl;
            int32_t w = 0, te = 0;
            int32_t s = nro_align(q, ql, tgt, tl, sc, NRO_MODE_WINDOW, wa, wb, &w, &te);
            if (s < sc->min_dp_score || s <= 0) {
                if (cell_score) cell_score[c] = -1;
                if (cell_wscore) cell_wscore[c] = 0;
                continue;
            }
            if (cell_score) cell_score[c] = s;
            if (cell_wscore) cell_wscore[c] = w;
            if (nt == 0 || w > wmax) { wmax = w; s1 = k1; s2 = k2; nt = 1; }
            else if (w == wmax) { s1 += k1; s2 += k2; ++nt; }
        }
        if (nt > 0) { best_wscore[r] = wmax; sum_k1[r] = s1; sum_k2[r] = s2; n_ties[r] = nt; status[r] = 0; }
        free(tgt); free(qf);
    }
    free(L); free(U1); free(M); free(U2); free(R); free(first); free(cnt);
    return 0;
}
