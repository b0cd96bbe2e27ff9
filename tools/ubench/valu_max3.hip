// valu_max3.hip -- gfx950: is v_pk_maximum3_f16 usable as a packed 3-input INTEGER max, and what does it cost?
//
// On bit patterns in [0x0400, 0x7BFF] (positive normal f16) the f16 order equals the integer order, so a
// 3-input packed maximum of biased non-negative int16 cells is one instruction instead of two v_pk_max_i16.
// Part 1 checks the results on edge patterns (incl. what denormals / zero do); part 2 measures issue cost
// alone and in the instruction mix of the sweep kernel's cell (old: 11 max2 per row; new: 5 max2 + 2.5 max3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 2048

__global__ void k_check(const unsigned* a, const unsigned* b, const unsigned* c, unsigned* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    out[i] = r;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int seed)
{
    int a[6]; for (int i = 0; i < 6; ++i) a[i] = 0x20002000 + ((seed + threadIdx.x * (i + 1)) & 0x0fff0fff);
    int b = 0x20002000 + ((seed * 3 + 1) & 0x0fff0fff), c = 0x00020002;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {          // 18 x v_pk_max_i16
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 1) {   // 18 x v_pk_maximum3_f16
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n v_pk_maximum3_f16 %0, %0, %1, %2\n v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == 2) {   // 18 x v_max3_i32
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == 3) {   // old cell, two rows: 2 perm + 22 max2 + 12 sub = 36
#pragma unroll
            for (int i = 0; i < 6; i += 3)
                asm volatile("v_perm_b32 %0, %0, %3, %4\n"
                             "v_pk_max_i16 %0, %0, %3\n v_sub_u32 %0, %0, %4\n v_pk_max_i16 %1, %1, %3\n v_sub_u32 %1, %1, %4\n"
                             "v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_max_i16 %2, %2, %0\n"
                             "v_pk_max_i16 %0, %0, %3\n v_pk_max_i16 %1, %1, %3\n v_sub_u32 %2, %2, %4\n v_sub_u32 %0, %0, %4\n"
                             "v_pk_max_i16 %2, %2, %3\n v_sub_u32 %1, %1, %4\n v_pk_max_i16 %0, %0, %3\n v_sub_u32 %2, %2, %4\n v_pk_max_i16 %1, %1, %3\n"
                             : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]) : "v"(b), "v"(c));
        } else if (MODE == 4) {   // new cell, two rows: 2 perm + 10 max2 + 5 max3 + 12 sub = 29
#pragma unroll
            for (int i = 0; i < 6; i += 3) {
                asm volatile("v_perm_b32 %0, %0, %3, %4\n"
                             "v_pk_max_i16 %0, %0, %3\n v_sub_u32 %0, %0, %4\n v_pk_max_i16 %1, %1, %3\n v_sub_u32 %1, %1, %4\n"
                             "v_pk_max_i16 %2, %2, %3\n v_pk_maximum3_f16 %0, %0, %1, %2\n v_pk_maximum3_f16 %1, %1, %2, %0\n"
                             "v_sub_u32 %2, %2, %4\n v_sub_u32 %0, %0, %4\n"
                             "v_pk_max_i16 %2, %2, %3\n v_sub_u32 %1, %1, %4\n v_pk_max_i16 %0, %0, %3\n v_sub_u32 %2, %2, %4\n"
                             : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]) : "v"(b), "v"(c));
            }
            asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[0]) : "v"(a[3]), "v"(a[4]));
        } else if (MODE == 5) {   // 18 x v_pk_add_u16 (is a packed add as cheap as v_add_u32 inside the mix?)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 6) {   // max3 / sub alternating
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2\n v_sub_u32 %0, %0, %2\n v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    int s = 0; for (int i = 0; i < 6; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int blocks, int* d, double n_instr)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, r); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double trips = 5.0 * blocks * 4.0 * ITER;
    double cyc = (ms * 1e-3) * 2.39e9 * 1024.0 / trips;
    printf("%-58s waves/SIMD=%d  %7.2f cycles/trip  %5.2f cycles/instr\n", name, blocks / 256, cyc, cyc / n_instr);
}

static unsigned short imax3(unsigned short a, unsigned short b, unsigned short c) { return std::max(a, std::max(b, c)); }

int main()
{
    // ---- part 1: results on integer patterns ----
    std::vector<unsigned short> pat = {0x0000, 0x0001, 0x03ff, 0x0400, 0x0401, 0x04ff, 0x0500, 0x1000, 0x1fff, 0x2000, 0x2001,
                                       0x3c00, 0x4000, 0x5e80, 0x7000, 0x7bfe, 0x7bff};
    for (int i = 0; i < 64; ++i) pat.push_back((unsigned short)(0x0400 + (rand() % (0x7bff - 0x0400))));
    std::vector<unsigned> ha, hb, hc;
    for (unsigned short x : pat) for (unsigned short y : pat) for (int z = 0; z < 9; ++z) {
        unsigned short w = pat[(size_t)(rand() % pat.size())];
        unsigned short x2 = pat[(size_t)(rand() % pat.size())], y2 = pat[(size_t)(rand() % pat.size())], w2 = pat[(size_t)(rand() % pat.size())];
        ha.push_back(x | ((unsigned)x2 << 16)); hb.push_back(y | ((unsigned)y2 << 16)); hc.push_back(w | ((unsigned)w2 << 16));
    }
    const int n = (int)ha.size();
    unsigned *da, *db, *dc, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), n * 4, hipMemcpyHostToDevice);
    k_check<<<(n + 255) / 256, 256>>>(da, db, dc, dout, n);
    std::vector<unsigned> ho(n);
    hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
    long bad_normal = 0, bad_sub = 0, n_normal = 0, n_sub = 0;
    for (int i = 0; i < n; ++i) {
        for (int h = 0; h < 2; ++h) {
            unsigned short x = (unsigned short)(ha[i] >> (16 * h)), y = (unsigned short)(hb[i] >> (16 * h)), w = (unsigned short)(hc[i] >> (16 * h));
            unsigned short r = (unsigned short)(ho[i] >> (16 * h));
            const bool all_normal = x >= 0x0400 && y >= 0x0400 && w >= 0x0400;
            if (all_normal) { ++n_normal; if (r != imax3(x, y, w)) ++bad_normal; }
            else { ++n_sub; if (r != imax3(x, y, w)) { if (bad_sub < 6) printf("  subnormal operand: max3(%04x,%04x,%04x) = %04x (integer max %04x)\n", x, y, w, r, imax3(x, y, w)); ++bad_sub; } }
        }
    }
    printf("v_pk_maximum3_f16 == integer max3: operands all in [0x0400,0x7bff]: %ld mismatches of %ld; with an operand < 0x0400: %ld of %ld\n",
           bad_normal, n_normal, bad_sub, n_sub);

    // ---- part 2: issue cost ----
    int* d; hipMalloc(&d, 256 * 8192 * 4 * 4);
    for (int blocks : {256 * 2, 256 * 3, 256 * 4, 256 * 5, 256 * 8}) {
        run<0>("18 x v_pk_max_i16", blocks, d, 18);
        run<1>("18 x v_pk_maximum3_f16", blocks, d, 18);
        run<2>("18 x v_max3_i32", blocks, d, 18);
        run<5>("18 x v_pk_add_u16", blocks, d, 18);
        run<6>("6 x (maximum3, v_sub_u32, v_pk_max_i16)", blocks, d, 18);
        run<3>("old cell mix, 2 rows (2 perm + 22 max2 + 12 sub = 36)", blocks, d, 36);
        run<4>("new cell mix, 2 rows (2 perm + 10 max2 + 5 max3 + 12 sub = 29)", blocks, d, 29);
    }
    return 0;
}
