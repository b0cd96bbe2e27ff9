/*
 * nanorepeat_amd.h -- C ABI of the MI355X-native NanoRepeat repeat-size scorer.
 *
 * This is the drop-in boundary for NanoRepeat's per-read repeat-size estimation
 * hot path.  The reference (WGLab/NanoRepeat 1.8.3) has no FFI: its boundary is
 * the string call `pyminimap2.main(cmd) -> (stdout, stderr)` with FASTA files on
 * disk, issued once per read (1D) or once per grid cell (2D).  Each entry point
 * below replaces one reference function together with the aligner calls it
 * makes; the file:line it replaces is cited on the declaration.
 *
 * Conventions: extern "C", plain pointers and sizes, caller-owned buffers,
 * returns 0 on success and a negative NRA_E_* code on failure (message via
 * nra_last_error(), thread-local).  Nothing throws across the boundary.  The
 * library must not be initialised before fork(); use one process per GPU.
 *
 * The scoring model is minimap2's documented `-x map-ont` objective
 * (match +2, mismatch -4, two-piece affine gap min(4+2l, 24+l), N = -1),
 * solved as an *optimal* local alignment.  See DESIGN.md for the exact
 * recurrences and tie-break rules; oracle/nr_oracle.c is the CPU restatement
 * the HIP kernels must match bit for bit.
 */
#ifndef NANOREPEAT_AMD_H
#define NANOREPEAT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRA_ABI_VERSION 4

/* error codes */
#define NRA_OK            0
#define NRA_E_ARG        -1   /* bad argument (null pointer, negative size, window too large ...) */
#define NRA_E_DEVICE     -2   /* HIP runtime error / no device */
#define NRA_E_RANGE      -3   /* a sequence exceeds what the entry point holds (limits on each declaration) */
#define NRA_E_NOMEM      -4
#define NRA_E_STATE      -5   /* the batch is not in the state the call needs (nra_batch2d_refine: the caller takes the two-call path) */

/* per-read status (1D and 2D) */
#define NRA_READ_OK        0  /* at least one best-scoring record passed the selector */
#define NRA_READ_FALLBACK  1  /* 1D only: records exist but no top-score record passes the flank
                                 test -> caller keeps round2_repeat_size (nanoRepeat_bam.py:433) */
#define NRA_READ_NO_RECORD 2  /* no candidate reached min_dp_score: the reference gets an empty PAF
                                 and leaves the read's size unset (nanoRepeat_bam.py:421) */
#define NRA_READ_SKIPPED   3  /* kmin > kmax on input: read has no round-2 estimate
                                 (nanoRepeat_bam.py:460) */

/* flags */
#define NRA_F_ALL_EXTENTS  1  /* explicit extents DP (tstart/tend) for EVERY candidate; implies brute force */
#define NRA_F_TIE_EXTENTS  2  /* explicit extents DP for every top-score tie, so cand_tstart/cand_tend are
                                 filled for them (the one-shot call sets it when those arrays are given);
                                 without it only ties whose flank verdict is ambiguous are re-run */
#define NRA_F_TEST_CHAIN   8  /* testing only: sweep every read in chained 128-row blocks (the mechanism reads
                                 longer than 3072 bases use with 1536-row blocks) */
#define NRA_F_DPP_SWEEP    16 /* testing / comparison: sweep unchained reads with k_sweep_pk16 (DPP hand-off, combine on
                                 every step) instead of k_sweep_ring (LDS hand-off, combine on every m-th step) */
#define NRA_F_NO_HALF_WAVE 32 /* testing / comparison: reads of up to 768 bases take one pair per wave (k_sweep_ring) instead of
                                 two pairs per wave, 32 lanes each (k_sweep_ring32) */
#define NRA_F_NO_JOINT_PACK 64 /* testing / comparison, 2D: sweep the columns outside the scoring window in the int32 payload
                                 cells too, instead of packed int16 cells with two reads per wave (k_joint_pk16) */
#define NRA_F_SERIAL_CHAIN 128 /* testing / comparison, 1D: sweep the row blocks of a long read one after the other in one wave
                                 (k_sweep_ringchain) instead of as concurrent waves (k_sweep_ringmt) */
#define NRA_F_JOINT_TAILS 256  /* testing / comparison, 2D routed grids: junction at R[0] and one tail sweep over mid + unit2^k2 per
                                 (read, k1) -- what explicit cell lists use -- instead of the junction at the end of mid (extended
                                 reverse sweeps, MID sweeps, k_joint_combine) */
#define NRA_F_JOINT_NO_CHAIN 512 /* testing / comparison, 2D routed grids: the MID part (last prefix column + mid) as one systolic sweep per
                                 (read, k1) resuming from the prefix sweep's wave state, instead of column-parallel scans
                                 (k_joint_midscan) on the column states the prefix sweep leaves */
#define NRA_F_JOINT_NO_KEEP 1024 /* testing / comparison, 2D routed grids: every grid of a batch sweeps its reads again, instead of keeping
                                 the column states a coarse grid's sweeps leave at EVERY repeat count of a read's range, from which a
                                 finer grid inside those ranges (the reference's round 3 after round 2) needs no sweep at all */
#define NRA_F_NO_QUANTA 2048  /* testing / comparison, 1D: a bucket's reverse sweeps and forward sweeps as two launches (k_sweep_ring /
                                 k_sweep_ring32) instead of one launch of quanta taken by ticket -- the sweeps cut into parts of a few
                                 hundred steps (k_sweep_ringq) */
#define NRA_F_QUANTA_2L 4096  /* accepted and ignored since the sweeps' quanta became parts of a few hundred steps (it ran the three
                                 quanta of round 4's first form as two launches without tickets: measured no better than no quanta) */
#define NRA_F_BRUTE_FORCE  4  /* score the K candidates of a read as K independent alignments
                                 (k_score_pk16) instead of the junction decomposition (k_sweep_pk16) */

/* Scoring parameters: minimap2 `-x map-ont` defaults.
 * A gap of length l costs min(gap_open1 + l*gap_ext1, gap_open2 + l*gap_ext2). */
typedef struct nra_scoring {
    int32_t match;        /* +2  */
    int32_t mismatch;     /*  4  (penalty, positive) */
    int32_t gap_open1;    /*  4  */
    int32_t gap_ext1;     /*  2  */
    int32_t gap_open2;    /* 24  */
    int32_t gap_ext2;     /*  1  */
    int32_t sc_ambi;      /*  1  (penalty for N against anything) */
    int32_t min_dp_score; /* 80  records with a lower score are absent (minimap2 -s) */
} nra_scoring_t;

/* One 1D repeat region.  Candidate k is the sequence left + unit*k + right
 * (nanoRepeat_bam.py:478-481).  Sequences are ASCII (ACGTN, either case). */
typedef struct nra_region {
    const char* left;
    const char* unit;
    const char* right;
    int32_t left_len;
    int32_t unit_len;
    int32_t right_len;
} nra_region_t;

/* One joint (two adjacent motifs) region.  Candidate (k1,k2) is
 * left + unit1*k1 + mid + unit2*k2 + right (nanoRepeat_joint.py:499-505). */
typedef struct nra_joint_region {
    const char* left;
    const char* unit1;
    const char* mid;
    const char* unit2;
    const char* right;
    int32_t left_len;
    int32_t unit1_len;
    int32_t mid_len;
    int32_t unit2_len;
    int32_t right_len;
} nra_joint_region_t;

/* Work and timing counters of a batch (timings: HIP events on the streams the kernels run on). */
typedef struct nra_stats {
    int64_t n_alignments;     /* (read, candidate) pairs scored */
    int64_t algorithmic_cells;/* sum of qlen * tlen over those pairs */
    int64_t executed_cells;   /* DP cells the scoring kernels actually updated (padding included); far below
                                 algorithmic_cells when the junction decomposition shares work across k */
    int64_t algorithmic_bytes;/* HBM bytes the algorithm must move (packed reads + flanks + results) */
    int64_t n_extent_tasks;   /* alignments re-run by the extents kernel (top-score ties) */
    double  score_kernel_ms;  /* dominant kernel: sum of its launches, HIP events on the batch stream */
    double  extent_kernel_ms; /* second-pass kernel (1D extents) */
    double  total_ms;         /* first launch -> last launch of the run, HIP events */
    int32_t n_score_launches;
    int32_t n_runs;           /* completed runs (nra_batch_run + nra_batch_sync) the sums below cover */
    double  score_phase_ms;   /* wall time of the scoring phase on the device (first scoring launch ->
                                 last one done); < score_kernel_ms when launches of different read-length
                                 buckets overlap on their own streams */
    /* the four timings above belong to the LAST run; these are summed over all n_runs runs */
    double  sum_score_kernel_ms, sum_extent_kernel_ms, sum_total_ms, sum_score_phase_ms;
    int64_t intermediate_bytes; /* HBM bytes the decomposition itself moves per run between its kernels: the R side
                                   of the junction (1D) / the wave states and the R side (2D), written once, read once */
} nra_stats_t;

typedef struct nra_batch nra_batch_t;   /* device-resident inputs + outputs of one call */

/* ---- library ------------------------------------------------------------------ */
int         nra_abi_version(void);
const char* nra_version(void);
const char* nra_last_error(void);
int         nra_device_count(void);              /* < 0 on error */
void        nra_default_scoring(nra_scoring_t* sc);
/* The library keeps a few device chunks (<= 2 GiB per device), pinned staging buffers, streams and events of
 * destroyed batches for the next call (hipMalloc / hipStreamCreate dominate a small one-shot call otherwise), and
 * gives the chunks back by itself when a device allocation fails.  This call gives them back now (device < 0: on
 * every device); batches that are alive are not touched. */
int         nra_release_cached_memory(int device);

/* ---- 1D: replaces round3_estimation(data_type, fast_mode, repeat_region, num_cpu)
 *      nanoRepeat_bam.py:446-450 (= round3_align :452-500, one pymm2.main call per
 *      read at :497, + round3_estimation_from_alignment :436-444) ---------------------
 *
 * Inputs: n_regions regions; n_reads oriented core sequences concatenated in `seqs`
 * with offsets seq_off[n_reads+1]; read_region[i] = region index of read i (NULL when
 * n_regions == 1); candidate window kmin[i]..kmax[i] inclusive (kmin > kmax = skipped).
 * A read holds at most 200 000 bases and a candidate template 4 000 000 columns (NRA_E_RANGE beyond).
 * Reads of up to 3072 bases run two to a wave in packed int16 cells; longer ones (any round-2 size
 * the reference's window rule covers, nanoRepeat_bam.py:463-472) run one to a wave as chained row
 * blocks in int32 cells and need the junction decomposition: no NRA_F_BRUTE_FORCE / ALL_EXTENTS.
 *
 * Per-read outputs (all caller-allocated, n_reads entries):
 *   best_score  max AS over the read's candidates (0 when no record)
 *   sum_k,n_ties  sum and count of k over the records tied at best_score that pass
 *               tstart < left_len && tlen - tend < right_len  (nanoRepeat_bam.py:426-428);
 *               the repeat size is sum_k / n_ties in float64 (= np.mean at :431)
 *   status      NRA_READ_*
 * Optional per-candidate outputs (NULL to skip), sum_i max(0, kmax-kmin+1) entries in
 * read order then k order: cand_score (AS, or -1 when below min_dp_score), cand_tstart,
 * cand_tend (-1 unless the candidate ties the best score -- NRA_F_TIE_EXTENTS, implied when
 * these arrays are given -- or NRA_F_ALL_EXTENTS is set). */
int nra_round3_1d(int device,
                  const nra_region_t* regions, int32_t n_regions,
                  int32_t n_reads, const char* seqs, const int64_t* seq_off,
                  const int32_t* read_region,
                  const int32_t* kmin, const int32_t* kmax,
                  const nra_scoring_t* sc, int32_t flags,
                  int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                  int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend);

/* ---- 2D: replaces the aligner loop + selector of
 *      round2_estimation_of_repeat_size nanoRepeat_joint.py:397-421 and
 *      round3_estimation_of_repeat_size nanoRepeat_joint.py:315-347 (one pymm2.main
 *      call per grid cell at :417 / :341), followed by
 *      estimate_two_repeats_from_paf :427-478 and the CIGAR window rescoring
 *      tk.target_region_alignment_stats_from_cigar tk.py:435-500 ----------------------
 *
 * Inputs: one joint region; n_reads full reads (either strand, at most 200 000 bases; reads beyond
 * 3072 bases are scored uncut, cell by cell, in chained row blocks); a list of n_cells
 * (read, k1, k2) grid cells, grouped by read (cell_read non-decreasing).
 * read_strand (n_reads, in/out, may be NULL): 0 = choose the strand with the higher DP
 * score against the read's first listed cell (ties -> '+'), +1 / -1 = forced; on return
 * holds the strand used.
 *
 * Per-cell outputs (optional): cell_score = AS (-1 when below min_dp_score),
 * cell_wscore = window score over [max(0,L-10), min(tlen, L+m1*k1+mid+m2*k2+10))
 * (nanoRepeat_joint.py:445-449).  Per-read outputs: best_wscore, sum_k1, sum_k2, n_ties
 * over the cells tied at the maximum window score (:458-476), status OK / NO_RECORD. */
int nra_joint_2d(int device,
                 const nra_joint_region_t* region,
                 int32_t n_reads, const char* seqs, const int64_t* seq_off,
                 int8_t* read_strand,
                 int64_t n_cells, const int32_t* cell_read,
                 const int32_t* cell_k1, const int32_t* cell_k2,
                 const nra_scoring_t* sc, int32_t flags,
                 int32_t* cell_score, int32_t* cell_wscore,
                 int32_t* best_wscore, int64_t* sum_k1, int64_t* sum_k2,
                 int32_t* n_ties, uint8_t* status);

/* ---- generic batched local alignment: the DP engine of the path, exposed for the rows
 *      around it: anchor finding nanoRepeat_bam.py:260-286 (anchors vs
 *      reads, pymm2.main at :281) and the round-2 estimate :334-393 (cores vs left + unit*T,
 *      pymm2.main at :362), each of which is one aligner call over many reads in the reference.
 *
 * n_seqs sequences concatenated in `seqs` (offsets seq_off[n_seqs+1]); n_pairs pairs
 * (pair_query[i], pair_target[i]) of sequence indices.  A sequence used as a query holds at
 * most 200 000 bases (above 3072: chained row blocks), as a target at most 4 000 000 (a whole
 * long read as the target of an anchor); pairs whose score or extents outgrow the int32 cells
 * (score > 32000, target > 65000) run in int64 cells.  nra_align_pairs_cigar: query <= 3072,
 * target <= 65000.  Outputs per pair: score (AS; -1 when below
 * min_dp_score), tstart, tend (target coordinates, 0-based half-open; oracle tie-breaks:
 * largest tstart, then smallest tend; -1 when no record). */
int nra_align_pairs(int device,
                    int32_t n_seqs, const char* seqs, const int64_t* seq_off,
                    int64_t n_pairs, const int32_t* pair_query, const int32_t* pair_target,
                    const nra_scoring_t* sc, int32_t flags,
                    int32_t* score, int32_t* tstart, int32_t* tend);

/* ---- alignment paths in the reference's wire format: the same pairs as
 *      nra_align_pairs, plus the query extents and a minimap2-style --eqx CIGAR ("12=1X3I4D")
 *      of the co-optimal path the oracle's traceback picks (paf.py:32-79 carries it as cg:Z).
 * cigar: caller buffer of cigar_cap bytes; pair i's NUL-terminated string starts at
 * cigar + cigar_off[i] (cigar_off has n_pairs + 1 entries; empty string when no record).
 * The trace needs qlen * tlen bytes of device memory per pair (8 GiB per call at most). */
int nra_align_pairs_cigar(int device,
                          int32_t n_seqs, const char* seqs, const int64_t* seq_off,
                          int64_t n_pairs, const int32_t* pair_query, const int32_t* pair_target,
                          const nra_scoring_t* sc, int32_t flags,
                          int32_t* score, int32_t* tstart, int32_t* tend,
                          int32_t* qstart, int32_t* qend,
                          char* cigar, int64_t cigar_cap, int64_t* cigar_off);

/* ---- device-resident batches (what bench.py times): create = encode + H2D,
 *      run = kernels only (asynchronous on the batch's own stream), fetch = D2H ------- */
int  nra_batch1d_create(int device,
                        const nra_region_t* regions, int32_t n_regions,
                        int32_t n_reads, const char* seqs, const int64_t* seq_off,
                        const int32_t* read_region,
                        const int32_t* kmin, const int32_t* kmax,
                        const nra_scoring_t* sc, int32_t flags,
                        nra_batch_t** out);
int  nra_batch2d_create(int device,
                        const nra_joint_region_t* region,
                        int32_t n_reads, const char* seqs, const int64_t* seq_off,
                        const int8_t* read_strand,
                        int64_t n_cells, const int32_t* cell_read,
                        const int32_t* cell_k1, const int32_t* cell_k2,
                        const nra_scoring_t* sc, int32_t flags,
                        nra_batch_t** out);
/* The joint mode scores the same reads in two grid rounds (nanoRepeat_joint.py:266-269): the reads can be
 * packed and uploaded once (create_reads) and each round's cell list set on the resident batch
 * (set_cells; may be called again after a run -- device buffers are reused).  nra_batch2d_create is the two
 * calls in one. */
int  nra_batch2d_create_reads(int device,
                              const nra_joint_region_t* region,
                              int32_t n_reads, const char* seqs, const int64_t* seq_off,
                              const nra_scoring_t* sc, int32_t flags,
                              nra_batch_t** out);
int  nra_batch2d_set_cells(nra_batch_t* b, const int8_t* read_strand,
                           int64_t n_cells, const int32_t* cell_read,
                           const int32_t* cell_k1, const int32_t* cell_k2);
/* A whole grid round in one call -- the reference's routing of reads to grid cells, done by the library
 * (round 2: nanoRepeat_joint.py:397-409, round 3: :315-330).  Axis a (1, 2) has the grid values
 * g = start_a + i * step_a, i in [0, count_a) (round 2: range(round1_min, round1_max + 1, step), :397-398;
 * round 3: range(max(0, int(min size - s)), int(max size + s + 2)), :298-303); read r takes the values with
 * lo_a[r] <= g < hi_a[r] (round 2: its round-1 range [min, max), :407; round 3: [max(size - s, range min),
 * min(size + s, range max)), :325-330 -- doubles, because the round-2 sizes are means), and no cells when either
 * axis gives it none.  A read's cells are listed k1-major, k2 ascending, reads in input order: the order
 * nra_batch2d_set_cells wants and the per-cell arrays of nra_batch2d_fetch follow.
 * nra_joint_grid_cells is the routing alone, on the host (no device needed): returns the number of cells and, when
 * the three arrays are given (cap entries each), the list itself.  nra_batch2d_set_grid = the routing + set_cells,
 * without per-cell arrays crossing the boundary; *n_cells (optional) receives the number of cells. */
int64_t nra_joint_grid_cells(int32_t n_reads,
                             int32_t start1, int32_t step1, int32_t count1, const double* lo1, const double* hi1,
                             int32_t start2, int32_t step2, int32_t count2, const double* lo2, const double* hi2,
                             int64_t cap, int32_t* cell_read, int32_t* cell_k1, int32_t* cell_k2);
int  nra_batch2d_set_grid(nra_batch_t* b, const int8_t* read_strand,
                          int32_t start1, int32_t step1, int32_t count1, const double* lo1, const double* hi1,
                          int32_t start2, int32_t step2, int32_t count2, const double* lo2, const double* hi2,
                          int64_t* n_cells);
/* A later cell list reuses what an earlier one left on the device for the same read and strand: the packed sweeps of
 * the two flanks, and -- routed grids with every strand given -- the column states on either side of the junction, which
 * a grid's sweeps leave at EVERY repeat count k with lo_a[r] <= k < hi_a[r] (no further than one step beyond the read's
 * first / last grid value), not only at the grid's own values: a later grid whose cells all lie inside (round 3 after
 * round 2) runs no sweep at all (results identical; NRA_F_JOINT_NO_KEEP switches this off).  nra_batch2d_invalidate
 * drops all of it: the next list starts like the first (a benchmark repeating the two rounds on one resident batch
 * calls it at the top of every repetition). */
int  nra_batch2d_invalidate(nra_batch_t* b);
/* The flank sweeps ahead of the cell list.  What a joint run sweeps first -- L and rev(R) up to the scoring window, a third
 * of a round's device time -- depends on the reads and their strands only.  A caller that knows every strand (round 1 of
 * the reference does, nanoRepeat_joint.py:509-649) calls this right after nra_batch2d_create_reads / _invalidate, BEFORE it
 * derives step sizes, bounds and grids on the host (:239-259, :351-374): the kernels are enqueued and the call returns;
 * the cell lists that follow find the flank states valid for those strands and sweep no flank (a read with strand 0
 * here, or another strand later, is swept by its cell list as before).  Results are identical with and without the
 * call; a batch that sweeps no packed flanks (brute force, short flanks) does nothing. */
int  nra_batch2d_sweep_flanks(nra_batch_t* b, const int8_t* read_strand);
/* The reference's round 3 (round3_estimation_of_repeat_size, nanoRepeat_joint.py:275-349) as a REFINEMENT of the routed
 * grid whose run has just been enqueued (set_grid -> nra_batch_run -> this call, before anything waits for the run):
 * routed on the device from that grid's per-read results, without the host seeing them.  Read r with a result (status
 * OK, n_ties > 0) has the sizes size_a = sum_ka / n_ties (float64, the mean of its tied cells, :473-474) and takes the
 * unit-step cells k_a with  max(size_a - buf_a, lo_a[r]) <= k_a < min(size_a + buf_a, hi_a[r])  on both axes (:320-330:
 * buf_a = the grid's step on axis a, [lo, hi) = the read's round-1 range, the bounds the grid was routed with; lo >= 0);
 * a read without a result takes none.  The cells are scored from the column states the grid's sweeps kept (no sweep
 * runs), and the per-read outputs of nra_batch2d_fetch are then the refinement's (status NO_RECORD for a read without
 * cells); its per-cell arrays hold (2 buf1)(2 buf2) entries a read, the read's n1 x n2 cells first (k1-major),
 * and nra_stats_t.n_alignments counts the cells of both grids.  What the host saves: a fetch, a second routing and task
 * list, and the idle device between the two rounds.  NRA_E_STATE when the batch kept no column states for its current
 * grid (strands not all given, explicit cell list, NRA_F_JOINT_NO_KEEP / _TAILS / _NO_CHAIN, reads beyond 3072 bases,
 * kept states over budget), when some read's refinement could reach a count no column state was kept at (other buffers or
 * bounds than the grid's; a grid that itself ran from the states an earlier grid kept), or when the run was already waited
 * for: the caller then fetches and calls nra_batch2d_set_grid for the finer grid itself -- same results. */
int  nra_batch2d_refine(nra_batch_t* b, int32_t buf1, int32_t buf2,
                        const double* lo1, const double* hi1, const double* lo2, const double* hi2);
int  nra_batch_run(nra_batch_t* b);      /* enqueue every kernel of the path; returns at once */
int  nra_batch_sync(nra_batch_t* b);     /* wait for the batch stream */
int  nra_batch_stats(nra_batch_t* b, nra_stats_t* st);   /* after sync */
int  nra_batch1d_fetch(nra_batch_t* b,
                       int32_t* best_score, int64_t* sum_k, int32_t* n_ties, uint8_t* status,
                       int32_t* cand_score, int32_t* cand_tstart, int32_t* cand_tend);
int  nra_batch2d_fetch(nra_batch_t* b, int8_t* read_strand,
                       int32_t* cell_score, int32_t* cell_wscore,
                       int32_t* best_wscore, int64_t* sum_k1, int64_t* sum_k2,
                       int32_t* n_ties, uint8_t* status);
void nra_batch_destroy(nra_batch_t* b);

#ifdef __cplusplus
}
#endif
#endif /* NANOREPEAT_AMD_H */
