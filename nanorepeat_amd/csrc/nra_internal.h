// nra_internal.h -- structures shared by the host side (nra_host.cpp) and the gfx950
// kernels (nra_kernels.hip).  Not part of the public ABI (include/nanorepeat_amd.h).
#ifndef NRA_INTERNAL_H
#define NRA_INTERNAL_H

#include <stdint.h>
#include <hip/hip_runtime.h>

// Base codes.  Real bases 0..3, N = 4.  The two pad codes never compare equal to
// anything (including each other), so padded rows / columns only ever mismatch.
#define NRA_CODE_N    4
#define NRA_PAD_T     0x20   // target column outside [0, tlen)
#define NRA_PAD_Q     0x40   // query row >= qlen

// A candidate template is three consecutive pieces (1D: L+unit*k | - | R;
// 2D: L+unit1*k1 | mid+unit2*k2 | R).  Pieces 1 and 2 are stored pre-expanded to their
// largest k in the code pool, so a template is fully described by three lengths.
struct NraDevRegion {
    uint32_t p1_off;   // L + unit1 * k1max   (byte codes in the pool)
    uint32_t p2_off;   // mid + unit2 * k2max
    uint32_t p3_off;   // R
    uint32_t pr_off;   // rev(R) + rev(unit1) * k1max  (1D reverse sweep)
    int32_t  l1, m1;   // left_len, unit1_len
    int32_t  l2, m2;   // mid_len,  unit2_len (both 0 for 1D regions)
    int32_t  l3;       // right_len
};

struct NraDevRead {
    uint32_t qoff;     // first base of the read in the 2-bit pool (multiple of 16 bases)
    int32_t  qlen;
    int32_t  region;
    int32_t  rc;       // 1: the kernels read the reverse complement (2D '-' strand reads)
};

// Two candidates of one read scored by one wave (int16 halves A / B).
struct NraPairTask {
    int32_t read;
    int32_t k1a, k2a, k1b, k2b;
    int32_t out_a, out_b;  // index into the per-candidate score array; out_b < 0: no B half
    int32_t flags;         // bit 0: half B is the reverse complement of the template (strand probe);
                           // bit 1: store raw scores, no min_dp_score threshold
};

// Two reads of one region swept together (int16 halves A / B) by the junction-decomposition
// kernels; [kmin, kmax] is the union of their candidate windows.
struct NraSweepTask {
    int32_t read_a, read_b;   // read_b < 0: no second read
    int32_t kmin, kmax;
    uint64_t snap_off;        // the task's R-side snapshot (int32 index): 3 planes of R x 64 per row block
    int32_t read_c, read_d;   // k_sweep_ring32: the second pair of the wave (lanes 32..63); < 0: none
};

// R side of the junction as the half-wave kernel (k_sweep_ring32) keeps it: lane-major, [lane][H - o1 | E_in | E2_in][row],
// NRA_SNAP_LANE_STRIDE(R) dwords per lane (a multiple of 4: a lane stores its rows as 16-byte pieces, all in the one
// step it spends on the snapshot column, so every 64-byte sector it touches is complete when it leaves the wave; with
// the [row][lane] planes of the full-wave kernels two lanes per step add 4 bytes each to 128-byte lines that take 32
// steps to fill, and the half-wave kernel's WRITE_SIZE came to 2.6 x the bytes stored)
#define NRA_SNAP_LANE_STRIDE(R) ((3 * (R) + 3) / 4 * 4)

// One (task, row block) of the chained sweeps with concurrent blocks (k_sweep_ringmt): waves take these by ticket,
// in list order -- a producer (block b) precedes its consumer (block b + 1).
struct NraChainBlock {
    int32_t task;                  // index into the launch's sweep tasks
    int32_t blk, nblk;
    int32_t strip_in, strip_out;   // strips (5 planes of chain_cap 8-byte granules each) above / below the block; -1: none
};

// Tasks of the joint sweeps (nra_joint.hip).
//   tail sweep: one (read, k1) row of the 2D grid -- the read's cells with this k1 are
//     k2 = k2lo + n*k2step, n < n2, and sit at out + n in the cell arrays; `state` = where the
//     prefix sweep left the wave state for this (read, k1) (index into the int32 state buffer);
//   prefix sweep: one per read; its nk1 distinct k1 values, ascending, start at k1list[k1_off];
//     state of the i-th goes to state + i * (3R+7) * 64;
//   reverse sweep over R: one per read, only `read` is used.
//   resume != 0 (reverse and prefix sweeps): the first NRA_JOINT_PACKED_COLS columns were swept by k_joint_pk16;
//     the wave state it left is at pstate (index into the packed-state buffer), this read in half `phalf`.
//   Junction at the end of `mid` (routed grids, round 3 of the build): the reverse sweep runs on over rev(u2)^k2hi
//     and leaves, at the column of every k2 = k2lo + n*k2step, n < n2, its rows' column state in slot n of `state`
//     (3 planes of 64 * R int32: the wave's rows, padding included) and A(k2) at out + n; a MID sweep (one per (read, k1)) resumes like a tail, sweeps
//     the last prefix column + mid, and leaves its rows' column state at `pstate` and B(k1) at `out`.
//   k_joint_midscan: the column state of repeat count k1 is slot (k1 - task.k1) / task.k2step of `state` (the prefix
//     sweep may have left more of them than this grid asks for: kept column states).
struct NraJointTask {
    int32_t read;
    int32_t k1;
    int32_t k2lo, k2step, n2;
    int32_t out;
    int32_t k1_off, nk1;
    uint64_t state;
    uint64_t pstate;
    int32_t phalf, resume;
};

// One read's share of a routed grid (nra_batch2d_set_grid): its cells are the product of n1 values of k1 from k1lo in
// steps of the grid's step1 and n2 values of k2 from k2lo in steps of step2, k1-major.
struct NraGridRow { int32_t k1lo, n1, k2lo, n2; };

// One read of a routed grid for k_joint_combine: its n1 x n2 cells (k1-major at `out`) from the column states the
// MID sweeps (fs: n1 slots of 3 x qlen) and the extended reverse sweep (rs: n2 slots) left, B(k1) at fb, A(k2) at ra.
// The R side's slots may hold more k2 values than the grid asks for (kept column states): the grid's n-th k2 is slot
// rs_first + n * rs_stride of `rs` and of `ra`.
struct NraJointCombineTask {
    int32_t read, n1, n2, out;
    int32_t fb, ra, rs_first, rs_stride;
    uint64_t fs, rs;
    int32_t rs_plane, pad0;                     // rows of a plane of `rs`: 64 * (rows per lane of the read's bucket)
};

// Two reads of one rows-per-lane bucket whose payload-free columns -- L up to the scoring window, or rev(R) up to
// it -- are swept together in packed int16 cells (k_joint_pk16); the wave state goes to `state` (index into the
// packed-state buffer): NRA_JOINT_NPSTATE(R) planes of 64 dwords, both reads in the halves of every dword.
struct NraJointPairTask {
    int32_t read_a, read_b;   // read_b < 0: no second read
    uint64_t state;
};
#define NRA_JOINT_NPSTATE(R) (3 * (R) + 4)     // Hq, E_in, E2_in per row; diagonal, F, F2, running maximum
// columns without window payload at the start of a joint sweep: the window takes the last 10 bases of L
// (forward, dir 1) and the first 10 of R (the reverse sweep reads R backwards, dir 0)
#define NRA_JOINT_PACKED_COLS(flank) ((flank) > 10 ? (flank) - 10 : 0)
#define NRA_JOINT_PACKED_MIN_COLS 64           // shorter stretches stay in the int32 sweep

// One (query, target) pair whose path is wanted (nra_trace.hip).
struct NraTraceTask {
    int32_t read;        // query (2-bit pool)
    int32_t region;      // target = piece 1 of this region
    int32_t ops_cap;     // qlen + tlen
    int32_t pad;
    uint64_t trace_off;  // qlen * tlen bytes, row-major
    uint64_t ops_off;
};

// One candidate scored with a payload (extents / window kernels).
struct NraTask {
    int32_t read;
    int32_t k1, k2;
    int32_t out;
};

struct NraScoreParams {
    int32_t match, mismatch;       // +a, b (penalty)
    int32_t open1, ext1;           // q+e, e   (cost of a gap's first base / each further base)
    int32_t open2, ext2;           // q2+e2, e2
    int32_t ambi;                  // N penalty
    int32_t min_score;
};

// rows-per-lane instantiations (a read of qlen rows uses the smallest R with 64*R >= qlen)
#ifndef NRA_R_LIST
#define NRA_R_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) \
                      X(15) X(16) X(18) X(20) X(22) X(24) X(28) X(32) X(40) X(48)
#endif
#define NRA_MAX_R 48
#define NRA_MAX_QLEN_1BLOCK (64 * NRA_MAX_R)   // rows one wave holds in registers
// longer reads are swept in row blocks of 64*NRA_CHAIN_R rows chained through a scratch strip, one read
// per wave in int32 cells (1D sweeps) or int32 / int64 payload cells (extents, windows, free pairs)
#define NRA_CHAIN_R 24
#define NRA_CHAIN_R_TEST 2                     // tiny row blocks, for the tests (NRA_F_TEST_CHAIN)
#define NRA_CHAIN_STRIPS 512                   // waves of a chained launch = scratch strips, at most (the host sizes them by
                                               // the tasks there are and by NRA_CHAIN_SCRATCH_BUDGET)
#define NRA_CHAIN_SCRATCH_BUDGET (3ull << 30)  // bytes of scratch strips per buffer: a wider template gets fewer waves
#define NRA_MAX_QLEN 200000                    // read bases (chained row blocks above NRA_MAX_QLEN_1BLOCK)
// int32 values per lane in one dumped wave state of the 2D prefix sweep (3 per row + 7), shared by
// the kernel and the host so that the two cannot disagree
#define NRA_JOINT_NSTATE(R) (3 * (R) + 7)
// a COLUMN state of the prefix sweep (DIR 5, for k_joint_midscan) lives in the same slot, lane-major: per lane
// [Hq | E_in | E2_in | Hup_prev, M] padded to 16-byte pieces (<= 3R + 5 < NSTATE dwords)
#define NRA_JOINT_COLSTATE(R) ((3 * (R) + 2 + 3) / 4 * 4)
// wave states of one group of 2D prefix sweeps: at most this many int32 (16 GiB)
#define NRA_JOINT_STATE_CAP_INTS (4ull << 30)
#define NRA_JOINT_KEEP_BUDGET (96ull << 30)    // bytes of column states kept from one routed grid for the next (else: not kept): a third of the
                                               // device, and never more than half of what is free.  Kept states pay at every size measured (config 3's
                                               // reads x 12 / x 16: 33 / 45 GiB kept, 76.6 -> 64.0 / 101.8 -> 88.1 ms per run); the loss round 3 saw
                                               // at 32 GB was the runtime pinning pageable task arrays of >= 1 MB under running kernels (copy_h2d)
#define NRA_MAX_TLEN 65000     // int32 payload cells: tstart is 16 bits; + 64 pipeline columns
#define NRA_MAX_TLEN_WIDE 4000000   // int64 payload cells (score << 32 | payload)
// rows per lane of the int64 payload kernels' unchained instantiations (a rare path: three sizes suffice)
#define NRA_WIDE_R_SMALL 16
#define NRA_WIDE_R_LARGE 48

#ifdef __cplusplus
extern "C" {
#endif

// launchers (nra_kernels.hip).  All asynchronous on `st`; return hipError_t as int.
int nra_launch_score_pk16(int R, int has_n, hipStream_t st, int n_tasks,
                          const NraPairTask* tasks, const NraDevRead* reads,
                          const NraDevRegion* regions, const uint8_t* pool,
                          const uint32_t* q2bit, const uint32_t* qnmask,
                          NraScoreParams sp, int32_t* out_score);

// payload kernels walk a device-side queue of *count tasks with a grid stride.  ORIGIN outputs (score, tstart,
// tend); WINDOW outputs (score, wscore).  chain_buf != NULL: row-block chaining (n_waves strips of
// 6 * chain_cap cells each); wide: int64 cells (R = NRA_WIDE_R_SMALL / NRA_WIDE_R_LARGE, or chained)
int nra_launch_payload_origin(int R, int has_n, hipStream_t st, int n_waves,
                              const NraTask* tasks, const int32_t* count,
                              const NraDevRead* reads, const NraDevRegion* regions,
                              const uint8_t* pool, const uint32_t* q2bit, const uint32_t* qnmask,
                              NraScoreParams sp,
                              int32_t* out_score, int32_t* out_p, int32_t* out_tend,
                              void* chain_buf, int chain_cap, int wide);
int nra_launch_payload_window(int R, int has_n, hipStream_t st, int n_waves,
                              const NraTask* tasks, const int32_t* count,
                              const NraDevRead* reads, const NraDevRegion* regions,
                              const uint8_t* pool, const uint32_t* q2bit, const uint32_t* qnmask,
                              NraScoreParams sp,
                              int32_t* out_score, int32_t* out_p, int32_t* out_tend,
                              void* chain_buf, int chain_cap, int wide);

// junction decomposition (nra_sweep.hip): the reverse sweep over rev(R) writes the R-side snapshot and A
// (per read: read_a); the forward sweep combines and writes Score(k) + the flank-test verdict (0 fail,
// 1 pass, 2 ambiguous).  chain: int32 cells, one read per task, min(n_tasks, n_strips) waves
int nra_launch_sweep_bwd(int R, int has_n, int chain, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                         const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                         int32_t* snap,
                         int32_t* read_a, int32_t* chain_buf, int chain_cap, int n_strips);
int nra_launch_sweep_fwd(int R, int has_n, int chain, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                         const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                         int32_t* snap,
                         int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag, int32_t* chain_buf,
                         int chain_cap, int n_strips);

// the same sweeps with the lane-to-lane hand-off through an LDS ring (k_sweep_ring: one read block per wave,
// forward sweep skewed by the unit length so that the junction combine runs on every m-th step only);
// unchained reads, unit length <= NRA_SWEEP_RING_MAX_M
#define NRA_SWEEP_RING_MAX_M 8
int nra_launch_sweep_ring_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                              const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                              const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                              const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                              int32_t* snap, int32_t* read_a);
int nra_launch_sweep_ring_fwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                              const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                              const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                              const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                              int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag);

// half-wave LDS-ring sweeps (k_sweep_ring32): reads of up to 32 * NRA_RING32_MAX_R bases, two read pairs of one
// region per wave (32 lanes each); R from NRA_R_LIST up to NRA_RING32_MAX_R (16: 301.8, 20: 299.7, 24: 297.4 ms on config 4;
// 32, which would take config 2's 950-base reads too: 5.65 -> 7.1 ms there, 281 -> 288 ms on config 4)
#define NRA_RING32_MAX_R 24
int nra_launch_sweep_ring32_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                int32_t* snap, int32_t* read_a);
int nra_launch_sweep_ring32_fwd(int R, int has_n, hipStream_t st, int n_tasks, const NraSweepTask* tasks,
                                const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag);

// a bucket's reverse and forward sweeps as one launch of quanta taken by ticket (k_sweep_ringq).  A sweep is cut every
// `qsteps` steps (a multiple of 64; a forward sweep also at NRA_Q_CUT of its first boundary step) into parts; qlist holds one entry per part, direction << 31 | part << NRA_Q_PART_SHIFT |
// task, ordered [reverse parts 0 of every task | reverse parts 1 | ... | forward parts 0 | ...]; the host and the kernel
// count a sweep's steps with the same macros.  `arrivals`: two counters per task (finished reverse / forward parts) and
// `ticket`, zeroed before the launch; `giveup` the launch-wide give-up word; qstate: slots of NRA_QSTATE_INTS(R) x 64 int32
// (the registers, the ring's places, then the pending outputs and boundary accumulators), one per forward sweep and then,
// where a reverse sweep has more than one part, one per reverse sweep
#define NRA_QSTATE_INTS(R) (((4 * (R) + 2 + 3) / 4 + NRA_SWEEP_RING_MAX_M + 1) * 4)
#define NRA_Q_PART_SHIFT 27
#define NRA_Q_PART_MASK 15u
#define NRA_Q_TASK_MASK ((1u << NRA_Q_PART_SHIFT) - 1u)
#define NRA_Q_MAX_PARTS 16
#define NRA_Q_STEPS 384          // steps of a part (config 2: a reverse sweep in 3 parts, a forward sweep in 7)
#define NRA_Q_CUT(jfirst) ((jfirst) < 0 ? 0 : (jfirst) / 64 * 64)      /* a forward sweep's first cut: before its first boundary step */
#define NRA_Q_WHOLE (1 << 30)   // "steps of a part" of a batch too small for more cuts (longer than any sweep): a reverse sweep, a forward sweep to
                                // its first cut, the rest
#define NRA_Q_STEPS_REV(l3, half) ((l3) + ((half) ? 31 : 63))
#define NRA_Q_STEPS_FWD(l1, m, kmax, half) ((l1) + (m) * (kmax) + ((half) ? 31 : 63) * (m))
int nra_launch_sweep_ringq(int R, int has_n, hipStream_t st, int n_quanta, const uint32_t* qlist, int qsteps, int n_tasks, int32_t* ticket,
                           int32_t* arrivals, int32_t* giveup, int32_t* qstate, const NraSweepTask* tasks,
                           const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                           const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                           const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                           int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag);
int nra_launch_sweep_ringq32(int R, int has_n, hipStream_t st, int n_quanta, const uint32_t* qlist, int qsteps, int n_tasks, int32_t* ticket,
                             int32_t* arrivals, int32_t* giveup, int32_t* qstate, const NraSweepTask* tasks,
                             const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                             const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                             const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                             int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag);

// chained LDS-ring sweeps (k_sweep_ringchain): reads of more than NRA_RING_CHAIN_MIN_ROWS rows as row blocks of
// 64 * NRA_RING_CHAIN_R; wide = 0: two reads per wave in packed int16, 1: one read per wave in int32 cells.
// chain_buf: n_strips strips of 10 * chain_cap int32; the launch has min(n_tasks, n_strips) waves
#define NRA_RING_CHAIN_R 20
#define NRA_RING_CHAIN_MIN_ROWS NRA_MAX_QLEN_1BLOCK   // shorter reads stay unchained: a block costs a pipeline fill and half a block of padding
#define NRA_RING_CHAIN_STRIPS 4096
// rows per lane of the chained sweeps whose row blocks run as concurrent waves (k_sweep_ringmt): 15 keeps the forward
// sweep within the 168 registers of three waves per SIMD (20: two)
#ifndef NRA_RING_MT_R
#define NRA_RING_MT_R 15
#endif
#define NRA_RING_MT_FROM 1536                  // reads of more rows than this may run as row blocks: one register block would take 28 - 48
                                               // rows per lane, one wave per SIMD (the batch takes the cheaper form: nra_batch1d_create)
#define NRA_RING_MT_R_MIN 12                   // ... down to 12: the host takes the height that pads a bucket's reads least
int nra_launch_sweep_ringchain_bwd(int R, int has_n, int wide, hipStream_t st, int n_tasks,
                                   const NraSweepTask* tasks, const NraDevRead* reads,
                                   const NraDevRegion* regions, const uint8_t* pool,
                                   const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                   const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                   int32_t* snap, int32_t* read_a, int32_t* chain_buf, int chain_cap, int n_strips);
int nra_launch_sweep_ringchain_fwd(int R, int has_n, int wide, hipStream_t st, int n_tasks,
                                   const NraSweepTask* tasks, const NraDevRead* reads,
                                   const NraDevRegion* regions, const uint8_t* pool,
                                   const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                   const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                   int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag,
                                   int32_t* chain_buf, int chain_cap, int n_strips);

// chained LDS-ring sweeps with the row blocks of a read as concurrent waves (k_sweep_ringmt).  `ticket`: one int32
// (zeroed by the launcher); `strips`: n strips x 5 x chain_cap granules, zeroed once; `epoch`: unique per launch,
// never 0; `error`: launch-wide give-up word (zeroed by the host before the run, read back after it)
int nra_launch_sweep_ringmt_bwd(int R, int has_n, int wide, hipStream_t st, int n_blocks,
                                const NraChainBlock* blocks, int32_t* ticket, const NraSweepTask* tasks,
                                const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                int32_t* snap, int32_t* read_a, uint64_t* strips, int chain_cap,
                                uint32_t epoch, int32_t* error);
int nra_launch_sweep_ringmt_fwd(int R, int has_n, int wide, hipStream_t st, int n_blocks,
                                const NraChainBlock* blocks, int32_t* ticket, const NraSweepTask* tasks,
                                const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                const int32_t* kmin, const int32_t* kmax, const uint32_t* coff,
                                int32_t* snap, int32_t* read_a, int32_t* cand_score, uint8_t* cand_flag,
                                uint64_t* strips, int chain_cap, uint32_t epoch, int32_t* error);

// 2D junction decomposition (nra_joint.hip)
int nra_launch_joint_bwd(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                         int32_t* snap, int32_t* read_a, const int32_t* pstate);
int nra_launch_joint_prefix(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                            const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                            const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                            const int32_t* k1list, int32_t* state, const int32_t* pstate);
// packed int16 sweep of the payload-free columns (dir 1: L, dir 0: rev(R)), two reads per wave; dir < 0: both sides in one
// launch, an L-side task marked in bit 63 of its state index and leaving its state at pstate_l
int nra_launch_joint_pk16(int R, int has_n, hipStream_t st, int n_tasks, const NraJointPairTask* tasks,
                          const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                          const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp, int dir,
                          int32_t* pstate, int32_t* pstate_l);
// junction at the end of mid: extended reverse sweeps, MID sweeps, the per-cell combine
int nra_launch_joint_bwd_ext(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                             const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                             const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                             int32_t* rsnap, int32_t* ra, const int32_t* pstate);
int nra_launch_joint_mid(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                         const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                         const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                         int32_t* state, int32_t* fsnap, int32_t* fb);
// ... with the MID part as column-parallel scans (k_joint_midscan): the prefix sweep leaves column states (each lane on its own step)
int nra_launch_joint_prefix_cols(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                                 const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                 const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                 const int32_t* k1list, int32_t* state, const int32_t* pstate);
int nra_launch_joint_midscan(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                             const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                             const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                             const int32_t* k1list, const int32_t* state, int32_t* fsnap, int32_t* fb,
                             const NraGridRow* rows);
int nra_launch_joint_combine(hipStream_t st, int n_tasks, const NraJointCombineTask* tasks, const NraDevRead* reads,
                             NraScoreParams sp, const int32_t* fsnap, const int32_t* rsnap, const int32_t* fb,
                             const int32_t* ra, int32_t* cell_score, int32_t* cell_wscore, const NraGridRow* rows);
// a refinement routed on the device (nra_batch2d_refine): rows != NULL in the two launchers above
int nra_launch_joint_refine_route(hipStream_t st, int n_reads, const uint8_t* status, const int32_t* n_ties,
                                  const int64_t* sum_k1, const int64_t* sum_k2, const double* lo1, const double* hi1,
                                  const double* lo2, const double* hi2, int buf1, int buf2, const NraGridRow* keep,
                                  NraGridRow* rows, uint32_t* cell_cnt, const int32_t* rowspad, const NraDevRead* reads,
                                  const NraDevRegion* regions, unsigned long long* words);
int nra_launch_joint_tail(int R, int has_n, hipStream_t st, int n_tasks, const NraJointTask* tasks,
                          const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                          const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                          int32_t* state, int32_t* snap, int32_t* read_a, int32_t* cell_score,
                          int32_t* cell_wscore);

// alignment paths (nra_trace.hip)
int nra_launch_trace_fill(int R, int has_n, hipStream_t st, int n_tasks, const NraTraceTask* tasks,
                          const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                          const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                          uint8_t* trace, int32_t* out);
int nra_launch_trace_back(hipStream_t st, int n_tasks, const NraTraceTask* tasks, const NraDevRead* reads,
                          const NraDevRegion* regions, const uint8_t* trace, const int32_t* fill_out,
                          uint8_t* ops, int32_t* out);

// 1D selectors (one wave per read).  append_mode: 0 none, 1 ambiguous ties only, 2 every tie
int nra_launch_select_best_1d(hipStream_t st, int n_reads, const int32_t* kmin, const int32_t* kmax,
                              const uint32_t* coff, const int32_t* cand_score, const uint8_t* cand_flag,
                              const int32_t* read_bucket, const uint32_t* bucket_task_base,
                              int append_mode, NraTask* ext_tasks, int32_t* ext_count,
                              int32_t* best_score);
int nra_launch_select_final_1d(hipStream_t st, int n_reads, const int32_t* kmin, const int32_t* kmax,
                               const uint32_t* coff, const NraDevRead* reads,
                               const NraDevRegion* regions,
                               const int32_t* cand_score, const uint8_t* cand_flag,
                               const int32_t* cand_tstart, const int32_t* cand_tend,
                               const int32_t* best_score,
                               int64_t* sum_k, int32_t* n_ties, uint8_t* status);
// 2D: strand choice from the probe scores, then the per-read selector over its cells
int nra_launch_pick_strand(hipStream_t st, int n_reads, const int32_t* probe_score,
                           const int8_t* strand_in, int8_t* strand_out, NraDevRead* reads);
int nra_launch_select_2d(hipStream_t st, int n_reads, const uint32_t* cell_first,
                         const uint32_t* cell_cnt, const int32_t* cell_k1, const int32_t* cell_k2,
                         const NraGridRow* grid_rows, int grid_step1, int grid_step2,
                         const int32_t* cell_score, const int32_t* cell_wscore,
                         int32_t* best_w, int64_t* sum_k1, int64_t* sum_k2, int32_t* n_ties,
                         uint8_t* status);

#ifdef __cplusplus
}
#endif
#endif
