/*
 * nr_oracle.h -- CPU restatement (oracle) of NanoRepeat's repeat-size scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nanorepeat_amd/ may link, import or call
 * this library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: the selection logic, window rule, PAF semantics and CIGAR window
 * rescoring are pinned by golden vectors captured from the reference's own Python
 * (tests/golden/, generator tests/golden/make_golden.py).  The aligner itself
 * (minimap2 2.30 via pyminimap2, un-vendored, absent offline) is restated from its
 * published objective: "aligner parity unpinned" (SURVEY.md 8c).
 *
 * Entry points mirror include/nanorepeat_amd.h one for one (nro_ instead of nra_,
 * no device argument) so parity tests call both through one loader.
 */
#ifndef NR_ORACLE_H
#define NR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nro_scoring {
    int32_t match, mismatch, gap_open1, gap_ext1, gap_open2, gap_ext2, sc_ambi, min_dp_score;
} nro_scoring_t;

typedef struct nro_region {
    const char* left; const char* unit; const char* right;
    int32_t left_len, unit_len, right_len;
} nro_region_t;

typedef struct nro_joint_region {
    const char* left; const char* unit1; const char* mid; const char* unit2; const char* right;
    int32_t left_len, unit1_len, mid_len, unit2_len, right_len;
} nro_joint_region_t;

#define NRO_MODE_ORIGIN 0   /* payload = tstart of the alignment  (1D)            */
#define NRO_MODE_WINDOW 1   /* payload = window score of tk.py:435-500 (2D)       */

void nro_default_scoring(nro_scoring_t* sc);
void nro_set_threads(int n);            /* OpenMP threads used by the batch entry points */
int  nro_get_threads(void);

/* ASCII -> codes A0 C1 G2 T3(U) other 4; either case. */
void nro_encode(const char* s, int64_t n, uint8_t* out);
/* reverse complement on codes (N stays 4) */
void nro_revcomp_codes(const uint8_t* in, int64_t n, uint8_t* out);

/* One optimal local alignment of query codes vs target codes (two-piece affine).
 * mode ORIGIN: *payload = tstart (0-based, ties -> largest tstart), *tend = exclusive end
 *              (ties -> smallest tend).
 * mode WINDOW: *payload = window score over target [wa, wb) of the max-(score,window)
 *              alignment; *tend as above.
 * Returns the score (>= 0; 0 = nothing aligned, payload/tend then 0). */
int32_t nro_align(const uint8_t* q, int32_t qlen, const uint8_t* t, int32_t tlen,
                  const nro_scoring_t* sc, int mode, int32_t wa, int32_t wb,
                  int32_t* payload, int32_t* tend);

/* Same alignment with traceback: writes a minimap2-style --eqx CIGAR ("12=1X3I...")
 * into cigar (capacity cap, NUL terminated) and the extents.  Returns score, or -1 if
 * cap is too small.  O(qlen*tlen) memory: small cases only. */
int32_t nro_align_cigar(const uint8_t* q, int32_t qlen, const uint8_t* t, int32_t tlen,
                        const nro_scoring_t* sc, int mode, int32_t wa, int32_t wb,
                        char* cigar, int32_t cap,
                        int32_t* tstart, int32_t* tend, int32_t* qstart, int32_t* qend,
                        int32_t* payload);

/* Restatement of tk.target_region_alignment_stats_from_cigar(...).score, tk.py:435-500.
 * Also returns the four counters when the pointers are non-NULL.  -2147483648 on a
 * malformed / unsupported CIGAR. */
int32_t nro_cigar_region_score(const char* cigar, int32_t tstart, int32_t tend,
                               int32_t ref_region_start, int32_t ref_region_end,
                               int32_t* num_match, int32_t* num_mismatch,
                               int32_t* num_ins, int32_t* num_del);

/* The 1D selector on one read's PAF records (nanoRepeat_bam.py:408-434), the function nro_round3_1d applies to every
 * read: record i = (k, AS, tstart, tend, tlen), AS < 0 = no record.  Returns the status (0 ties, 1 keep the round-2
 * size, 2 no record); the read's size is sum_k / n_ties. */
int nro_select_1d(int32_t n, const int32_t* k, const int32_t* score, const int32_t* tstart, const int32_t* tend,
                  const int32_t* tlen, int32_t left_len, int32_t right_len,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties);

/* Mirrors nra_round3_1d (include/nanorepeat_amd.h).  Always fills every candidate's
 * extents when flags has bit 0 set, else only for top-score ties (others -1). */
int nro_round3_1d(const nro_region_t* regions, int32_t n_regions,
                  int32_t n_reads, const char* seqs, const int64_t* seq_off,
                  const int32_t* read_region,
                  const int32_t* kmin, const int32_t* kmax,
                  const nro_scoring_t* sc, int32_t flags,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend);

/* nr_decomp.c -- NOT the oracle: the junction decomposition of the HIP sweeps as scalar C (same inputs and per-read outputs
 * as nro_round3_1d, no per-candidate arrays; regions need a base on either side of the repeat).  bench.py times it as the
 * CPU baseline of the SAME algorithm; the tests hold it against nro_round3_1d.  executed_cells (optional): DP cells updated. */
int nrd_round3_1d(const nro_region_t* regions, int32_t n_regions,
                  int32_t n_reads, const char* seqs, const int64_t* seq_off,
                  const int32_t* read_region, const int32_t* kmin, const int32_t* kmax,
                  const nro_scoring_t* sc,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int64_t* executed_cells);

/* Mirrors nra_align_pairs. */
int nro_align_pairs(int32_t n_seqs, const char* seqs, const int64_t* seq_off,
                    int64_t n_pairs, const int32_t* pair_query, const int32_t* pair_target,
                    const nro_scoring_t* sc, int32_t flags,
                    int32_t* score, int32_t* tstart, int32_t* tend);

/* Mirrors nra_joint_2d. */
int nro_joint_2d(const nro_joint_region_t* region,
                 int32_t n_reads, const char* seqs, const int64_t* seq_off,
                 int8_t* read_strand,
                 int64_t n_cells, const int32_t* cell_read,
                 const int32_t* cell_k1, const int32_t* cell_k2,
                 const nro_scoring_t* sc, int32_t flags,
                 int32_t* cell_score, int32_t* cell_wscore,
                 int32_t* best_wscore, int64_t* sum_k1, int64_t* sum_k2,
                 int32_t* n_ties, uint8_t* status);

#ifdef __cplusplus
}
#endif
#endif
