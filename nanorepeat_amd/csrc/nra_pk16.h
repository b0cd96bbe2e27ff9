// nra_pk16.h -- the packed int16 cell of the sweeps (two reads per wave), shared by nra_sweep.hip (1D) and
// nra_joint.hip (the payload-free stretches of the 2D sweeps).
#ifndef NRA_PK16_H
#define NRA_PK16_H
#include "nra_device.h"

// Registers hold (value + BIAS) in both int16 halves and every state stays inside [0x0400, 0x7bff]: there
// the bit pattern read as an f16 is a positive normal number whose order is the integer order, so
// gfx950's 3-input v_pk_maximum3_f16 is a packed 3-input INTEGER max (checked on hardware together
// with its issue cost: tools/ubench/valu_max3.hip, profiles/r02_valu_max3.txt) -- the 5-way max of a
// cell is two instructions instead of four.  Being non-negative, the halves never carry into each
// other, so adding or subtracting a small constant or another biased value is a plain 32-bit
// v_add/v_sub_u32.  The substitution score is one v_perm_b32 through a 4-entry byte table that travels
// with the template base.
//   range: one bias 2048 + doubled score <= 27000 -> 29048; the junction combine adds two biased
//   values, 4096 + the doubled score of ONE alignment -> <= 31096 < 0x7bff; the smallest real state is
//   2048 - 2 * (gap open 2 + a mismatch) ~ 1990, "minus infinity" is NEGB and survives one subtraction of a
//   gap extension (host: nra_host.cpp keeps doubled scores <= 27000 in these cells and sends scoring
//   schemes that do not fit the bounds to the brute-force kernel).
#define BIAS 2048
#define NEGB 1280                     // biased "minus infinity": below any real state, >= 0x0400 after - ext

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int half_lo(int v) { return v & 0xffff; }
__device__ __forceinline__ int half_hi(int v) { return (v >> 16) & 0xffff; }
__device__ __forceinline__ int pack2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int pmaxi(int a, int b) { return as_i(pmax(as_s(a), as_s(b))); }
// packed 3-input max of values in [0x0400, 0x7bff]: one v_pk_maximum3_f16
__device__ __forceinline__ int pmax3(int a, int b, int c)
{
    const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
    return __builtin_bit_cast(int, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}

// The LDS-ring hand-off (k_sweep_ring*, k_joint_pk16): lane l stores what enters lane l+1 and reads its own place one
// or more steps later.  The hardware needs no barrier -- a wave's LDS operations execute in program order, and every
// lane reads its place of a slot before any lane writes that place again -- but the order has to survive the
// compiler: per thread a step's store (place wr) and the next step's load (place lane) never alias, so nothing in
// the language keeps the load behind the store.  This fence does: wavefront scope, no instruction on gfx950 (the
// ISA of every kernel is the same with and without it), only the ordering of the LDS accesses around it.
__device__ __forceinline__ void ring_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// cell arithmetic of the sweeps: packed int16 pairs (two reads per wave, biased), or -- WIDE, the chained
// sweeps of reads longer than one register block -- plain int32 (one read per wave, no bias, no range limit)
template <bool W> __device__ __forceinline__ int mx2(int a, int b) { return W ? imax(a, b) : pmaxi(a, b); }
template <bool W> __device__ __forceinline__ int mx3(int a, int b, int c) { return W ? imax(imax(a, b), c) : pmax3(a, b, c); }

// One virtual systolic cell: rows [OFF, OFF+N) of the lane's arrays, one template column.
// diag = H(row above, j-1) - o1; F/F2 enter from the row above at this column and leave for the
// row below.  Returns nothing; the cell's last-row Hq is Hq[OFF+N-1].
//
// The floor of the local alignment (an alignment may start anywhere: max(H, 0) on the diagonal).  FF = false:
// on the diagonal, one 2-input max per cell, v_floor = 0 - o1 (+ the origin bit of THIS column).  FF = true:
// in the vertical-gap state, whose 2-input max has an input to spare -- F never falls below v_floor = 0 (+ the
// origin bit of the NEXT column: an alignment entering through H(i,j) = 0 starts at column j+1), so every H is
// >= 0 and the diagonal needs no max: 14.5 instead of 15.5 instructions per cell.  The caller feeds F >= v_floor
// and a diagonal >= v_floor - o1 into the lane's first row.  (What the floor adds -- "empty alignment, then a
// gap" states -- scores below the alignment that starts after the gap: never optimal, never a tie.)
template <int OFF, int N, int R, bool W = false, bool FF = false>
__device__ __forceinline__ void sweep_cell(int (&Hq)[R], int (&Hq2)[R], int (&E)[R], int (&E2)[R],
                                           const int (&qc)[R], int diag, int& F, int& F2, int& M,
                                           int tbl, int tbl_hi, int v_floor, int v_e1, int v_e2,
                                           int v_o1, int v_o2)
{
    if (N == 0) return;
    int d = (FF ? diag : mx2<W>(diag, v_floor)) + (int)__builtin_amdgcn_perm(tbl_hi, tbl, qc[OFF]);
    int h_prev = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const int i = OFF + n;
        int d_next = d;
        if (n + 1 < N)                         // uses H(i, j-1) before it is overwritten below
            d_next = (FF ? Hq[i] : mx2<W>(Hq[i], v_floor)) + (int)__builtin_amdgcn_perm(tbl_hi, tbl, qc[i + 1]);
        const int ein = mx2<W>(E[i] - v_e1, Hq[i]);           // E(i,j) from column j-1, lazily
        const int e2in = mx2<W>(E2[i] - v_e2, Hq2[i]);
        const int h = mx3<W>(mx3<W>(d, ein, F), e2in, F2);    // H(i,j)
        if (n & 1) M = mx3<W>(M, h_prev, h);                  // running maximum, two rows per instruction
        else if (n == N - 1) M = mx2<W>(M, h);
        else h_prev = h;
        E[i] = ein;
        E2[i] = e2in;
        const int hq = h - v_o1;               // stored instead of H: feeds E, F and the diagonal
        Hq[i] = hq;
        const int hq2 = h - v_o2;
        Hq2[i] = hq2;
        F = FF ? mx3<W>(F - v_e1, hq, v_floor) : mx2<W>(F - v_e1, hq);
        F2 = mx2<W>(F2 - v_e2, hq2);
        d = d_next;
    }
}

#endif  // NRA_PK16_H
