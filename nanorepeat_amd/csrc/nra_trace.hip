// nra_trace.hip -- alignment paths (CIGAR) for chosen (query, target) pairs (gfx950).
// Row 8f-2 of SURVEY.md: PAF/CIGAR emission in the reference's wire format (paf.py:32-79,
// consumers tk.py:405-500), for the few alignments a caller wants to look at -- not the hot path.
//
//   k_trace_fill<R,HAS_N>  the ORIGIN-payload DP of k_payload_i32 that also writes one byte per
//                          cell: which input gave H (with the oracle's preference d, E, F, E2, F2
//                          and "continue" vs "start here" for the diagonal), whether each of the
//                          four gap states leaving the cell extends or opens (extend preferred on
//                          ties, like the oracle's traceback), and whether the bases are equal.
//   k_trace_back           one thread per alignment walks the bytes back from the best cell
//                          (smallest column, then smallest row, holding the maximum) and writes
//                          the operations in reverse; the host run-length encodes them.
// The result is the string the CPU oracle's traceback produces, bit for bit.
#include "nra_device.h"

#ifndef NRA_PART
#define NRA_PART 0
#endif
#define NRA_HAS_PART(n) (NRA_PART == 0 || NRA_PART == (n))

#define TNEG (-(1 << 29))
#define T_SRC_DIAG 0      // H came through the diagonal and the path continues at (i-1, j-1)
#define T_SRC_E 1
#define T_SRC_F 2
#define T_SRC_E2 3
#define T_SRC_F2 4
#define T_SRC_START 5     // diagonal from an empty alignment: the path starts at this cell
#define T_E_EXT 0x08      // E(i, j+1) extends E(i, j)   (else it opens from H(i, j))
#define T_F_EXT 0x10      // F(i+1, j) extends F(i, j)
#define T_E2_EXT 0x20
#define T_F2_EXT 0x40
#define T_EQ 0x80         // the two bases are equal ('=' rather than 'X')

template <int R, bool HAS_N>
__global__ __launch_bounds__(WAVE) void k_trace_fill(int n_tasks, const NraTraceTask* __restrict__ tasks,
                                                     const NraDevRead* __restrict__ reads,
                                                     const NraDevRegion* __restrict__ regions,
                                                     const uint8_t* __restrict__ pool,
                                                     const uint32_t* __restrict__ q2bit,
                                                     const uint32_t* __restrict__ qnmask,
                                                     NraScoreParams sp, uint8_t* __restrict__ trace,
                                                     int32_t* __restrict__ out)   // 5 per task: score, tstart, tend, best_i, best_j
{
    const int task = blockIdx.x;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x;
    const NraTraceTask tk = tasks[task];
    const NraDevRead rd = reads[tk.read];
    const NraDevRegion rg = regions[tk.region];
    const uint8_t* __restrict__ tgt = pool + rg.p1_off;
    const int ncols = rg.l1;
    uint8_t* __restrict__ tr = trace + tk.trace_off;

    int qc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) qc[i] = query_code<HAS_N>(rd, q2bit, qnmask, lane * R + i);

    int Hprev[R], E[R], E2[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { Hprev[i] = TNEG; E[i] = TNEG; E2[i] = TNEG; }
    int Hbot = TNEG, Fout = TNEG, F2out = TNEG, Hup_prev = TNEG;
    int best = 0xffff, bestj = -1, besti = -1;
    int tt = NRA_PAD_T;
    int j = -lane;
    const int sA = sp.match << 16, sB = -(sp.mismatch << 16), sN = -(sp.ambi << 16);
    const int o1 = -(sp.open1 << 16), x1 = -(sp.ext1 << 16);
    const int o2 = -(sp.open2 << 16), x2 = -(sp.ext2 << 16);

    const int nchunks = (ncols + 63 + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int col = c * 64 + lane;
        int feed = col < ncols ? tgt[col] : NRA_PAD_T;
#pragma unroll 1
        for (int s = 0; s < 64; ++s) {
            int F = dpp_shr1(TNEG, Fout);
            int F2 = dpp_shr1(TNEG, F2out);
            tt = dpp_shr1(feed, tt);
            feed = dpp_rol1(feed);
            const int fresh = j;
            int diag = Hup_prev;
            Hup_prev = dpp_shr1(TNEG, Hbot);
            const bool live = j >= 0 && j < ncols;
            int colmax = TNEG, rowmax = 0, h = TNEG;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bool eq = qc[i] == tt;
                int sc = eq ? sA : sB;
                if (HAS_N) {
                    if ((qc[i] | tt) & 4) sc = sN;
                }
                const int d = imax(diag, fresh) + sc;
                h = imax(imax(d, E[i]), F);
                h = imax(imax(h, E2[i]), F2);
                int src = d == h ? (diag >= fresh ? T_SRC_DIAG : T_SRC_START)
                                 : (E[i] == h ? T_SRC_E : (F == h ? T_SRC_F : (E2[i] == h ? T_SRC_E2 : T_SRC_F2)));
                if (h > colmax) { colmax = h; rowmax = i; }
                diag = Hprev[i];
                Hprev[i] = h;
                const int ee = E[i] + x1, eo = h + o1;
                const int fe = F + x1, fo = h + o1;
                const int ee2 = E2[i] + x2, eo2 = h + o2;
                const int fe2 = F2 + x2, fo2 = h + o2;
                if (ee >= eo) src |= T_E_EXT;
                if (fe >= fo) src |= T_F_EXT;
                if (ee2 >= eo2) src |= T_E2_EXT;
                if (fe2 >= fo2) src |= T_F2_EXT;
                if (eq) src |= T_EQ;
                E[i] = imax(ee, eo);
                F = imax(fe, fo);
                E2[i] = imax(ee2, eo2);
                F2 = imax(fe2, fo2);
                const int row = lane * R + i;
                if (live && row < rd.qlen) tr[(size_t)row * ncols + j] = (uint8_t)src;
            }
            if (colmax > best) { best = colmax; bestj = j; besti = lane * R + rowmax; }
            Hbot = h; Fout = F; F2out = F2;
            ++j;
        }
    }
    // wave reduce: max packed value, then smallest column, then smallest row
    int vmax = best;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = imax(vmax, __shfl_xor(vmax, off, WAVE));
    int jm = (best == vmax) ? bestj : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) jm = imin(jm, __shfl_xor(jm, off, WAVE));
    int im = (best == vmax && bestj == jm) ? besti : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) im = imin(im, __shfl_xor(im, off, WAVE));
    if (lane == 0) {
        int32_t* o = out + (size_t)task * 5;
        const int sc = vmax >> 16;
        const int lo = sp.min_score > 1 ? sp.min_score : 1;
        if (sc >= lo && jm >= 0 && jm != 0x7fffffff) {
            o[0] = sc; o[1] = vmax & 0xffff; o[2] = jm + 1; o[3] = im; o[4] = jm;
        } else {
            o[0] = -1; o[1] = -1; o[2] = -1; o[3] = -1; o[4] = -1;
        }
    }
}

// ops are written back to front: ops[cap-1], ops[cap-2], ...; n_ops and the start cell are returned
__global__ void k_trace_back(int n_tasks, const NraTraceTask* __restrict__ tasks,
                             const NraDevRead* __restrict__ reads, const NraDevRegion* __restrict__ regions,
                             const uint8_t* __restrict__ trace, const int32_t* __restrict__ fill_out,
                             uint8_t* __restrict__ ops, int32_t* __restrict__ out)   // 3 per task: n_ops, qstart, tstart
{
    const int task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= n_tasks) return;
    const NraTraceTask tk = tasks[task];
    const int ncols = regions[tk.region].l1;
    const int32_t* f = fill_out + (size_t)task * 5;
    int32_t* o = out + (size_t)task * 3;
    if (f[0] < 0) { o[0] = 0; o[1] = -1; o[2] = -1; return; }
    const uint8_t* __restrict__ tr = trace + tk.trace_off;
    uint8_t* __restrict__ op = ops + tk.ops_off;
    const int cap = tk.ops_cap;
    int i = f[3], j = f[4], st = 0, n = 0;
    int qs = -1, ts = -1;
    // bounded by qlen + tlen operations: every step moves up, left or both
    for (int guard = reads[tk.read].qlen + ncols + 2; guard > 0 && n < cap; --guard) {
        const int b = tr[(size_t)i * ncols + j];
        if (st == 0) {
            const int src = b & 7;
            if (src == T_SRC_DIAG || src == T_SRC_START) {
                op[cap - 1 - n++] = (b & T_EQ) ? '=' : 'X';
                if (src == T_SRC_START) { qs = i; ts = j; break; }
                --i; --j;
            } else st = src;                       // 1 E, 2 F, 3 E2, 4 F2: no operation yet
        } else if (st == T_SRC_E || st == T_SRC_E2) {
            op[cap - 1 - n++] = 'D';               // target base j against a gap
            const int prev = tr[(size_t)i * ncols + (j - 1)];
            if (!(prev & (st == T_SRC_E ? T_E_EXT : T_E2_EXT))) st = 0;
            --j;
        } else {
            op[cap - 1 - n++] = 'I';               // query base i against a gap
            const int prev = tr[(size_t)(i - 1) * ncols + j];
            if (!(prev & (st == T_SRC_F ? T_F_EXT : T_F2_EXT))) st = 0;
            --i;
        }
    }
    o[0] = n; o[1] = qs; o[2] = ts;
}

#if NRA_HAS_PART(9)
extern "C" int nra_launch_trace_fill(int R, int has_n, hipStream_t st, int n_tasks, const NraTraceTask* tasks,
                                     const NraDevRead* reads, const NraDevRegion* regions, const uint8_t* pool,
                                     const uint32_t* q2bit, const uint32_t* qnmask, NraScoreParams sp,
                                     uint8_t* trace, int32_t* out)
{
    if (n_tasks <= 0) return 0;
#define CASE(r)                                                                                     \
    case r:                                                                                         \
        if (has_n) k_trace_fill<r, true><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, trace, out); \
        else k_trace_fill<r, false><<<n_tasks, WAVE, 0, st>>>(n_tasks, tasks, reads, regions, pool, q2bit, qnmask, sp, trace, out);       \
        break;
    switch (R) {
        NRA_R_LIST(CASE)
    default: return (int)hipErrorInvalidValue;
    }
#undef CASE
    return (int)hipGetLastError();
}

extern "C" int nra_launch_trace_back(hipStream_t st, int n_tasks, const NraTraceTask* tasks, const NraDevRead* reads,
                                     const NraDevRegion* regions, const uint8_t* trace, const int32_t* fill_out,
                                     uint8_t* ops, int32_t* out)
{
    if (n_tasks <= 0) return 0;
    k_trace_back<<<(n_tasks + 63) / 64, 64, 0, st>>>(n_tasks, tasks, reads, regions, trace, fill_out, ops, out);
    return (int)hipGetLastError();
}
#endif
