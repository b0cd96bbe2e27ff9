// nra_device.h -- device-side helpers shared by nra_kernels.hip and nra_sweep.hip.
#ifndef NRA_DEVICE_H
#define NRA_DEVICE_H
#include "nra_internal.h"

#define WAVE 64

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------
// cross-lane moves (DPP, full-wave shifts exist on gfx9-family CDNA)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_shr1(int old, int src)   // lane l <- lane l-1 ; lane 0 keeps old
{
    return __builtin_amdgcn_update_dpp(old, src, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_rol1(int src)            // lane l <- lane l+1 ; lane 63 <- lane 0
{
    return __builtin_amdgcn_update_dpp(src, src, 0x134 /*wave_rol:1*/, 0xf, 0xf, false);
}
// the same moves for 64-bit cells (two 32-bit DPP moves)
__device__ __forceinline__ long long dpp_shr1(long long old, long long src)
{
    const int lo = dpp_shr1((int)(unsigned)(old & 0xffffffffll), (int)(unsigned)(src & 0xffffffffll));
    const int hi = dpp_shr1((int)(old >> 32), (int)(src >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ long long dpp_rol1(long long src)
{
    const int lo = dpp_rol1((int)(unsigned)(src & 0xffffffffll)), hi = dpp_rol1((int)(src >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ s16x2 as_s(int v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ int as_i(s16x2 v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 splat(int v) { s16x2 r; r.x = (short)v; r.y = (short)v; return r; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ long long imax(long long a, long long b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------
// implicit template: piece1[0:len1) + piece2[0:len2) + piece3[0:len3), optional revcomp
// ------------------------------------------------------------------------------------
struct Tmpl {
    const uint8_t* p1; const uint8_t* p2; const uint8_t* p3;
    int len1, len2, tlen, rc;
};

__device__ __forceinline__ Tmpl make_tmpl(const NraDevRegion& rg, const uint8_t* pool, int k1, int k2, int rc)
{
    Tmpl t;
    t.p1 = pool + rg.p1_off; t.p2 = pool + rg.p2_off; t.p3 = pool + rg.p3_off;
    t.len1 = rg.l1 + rg.m1 * k1;
    t.len2 = rg.l2 + rg.m2 * k2;
    t.tlen = t.len1 + t.len2 + rg.l3;
    t.rc = rc;
    return t;
}

__device__ __forceinline__ int tmpl_code(const Tmpl& t, int col)
{
    if (col < 0 || col >= t.tlen) return NRA_PAD_T;
    int j = t.rc ? (t.tlen - 1 - col) : col;
    int c;
    if (j < t.len1) c = t.p1[j];
    else {
        j -= t.len1;
        c = (j < t.len2) ? t.p2[j] : t.p3[j - t.len2];
    }
    if (t.rc && c < 4) c = 3 - c;
    return c;
}

// query base of global row gi (PAD_Q beyond the read); 2-bit pool + optional N bitmap
template <bool HAS_N>
__device__ __forceinline__ int query_code(const NraDevRead& rd, const uint32_t* q2bit,
                                          const uint32_t* qnmask, int gi)
{
    if (gi >= rd.qlen) return NRA_PAD_Q;
    uint32_t b = rd.qoff + (uint32_t)(rd.rc ? (rd.qlen - 1 - gi) : gi);
    int c = (q2bit[b >> 4] >> ((b & 15u) * 2u)) & 3u;
    if (rd.rc) c = 3 - c;
    if (HAS_N) {
        if ((qnmask[b >> 5] >> (b & 31u)) & 1u) c = NRA_CODE_N;
    }
    return c;
}

#endif  // NRA_DEVICE_H
