// valu_group.hip -- does it pay to group the full-rate VALU ops of a mixed stream into runs? (gfx950)
// 18 instructions per trip, 12 v_pk_max_i16 (half rate) + 6 v_sub_u32 (full rate), either
// interleaved (pk pk sub) x 6 or grouped 12 pk then 6 sub, on 6 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int seed)
{
    int a[6]; for (int i = 0; i < 6; ++i) a[i] = seed + threadIdx.x * (i + 1);
    int b = seed * 3 + 1;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_pk_max_i16 %0, %0, %6\n v_pk_max_i16 %1, %1, %6\n v_pk_max_i16 %2, %2, %6\n v_pk_max_i16 %3, %3, %6\n v_pk_max_i16 %4, %4, %6\n v_pk_max_i16 %5, %5, %6\n"
                         "v_pk_max_i16 %0, %0, %6\n v_pk_max_i16 %1, %1, %6\n v_pk_max_i16 %2, %2, %6\n v_pk_max_i16 %3, %3, %6\n v_pk_max_i16 %4, %4, %6\n v_pk_max_i16 %5, %5, %6\n"
                         "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6\n v_sub_u32 %4, %4, %6\n v_sub_u32 %5, %5, %6"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "v"(b));
        } else if (MODE == 2) {
            asm volatile("v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6\n v_sub_u32 %4, %4, %6\n v_sub_u32 %5, %5, %6\n"
                         "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6\n v_sub_u32 %4, %4, %6\n v_sub_u32 %5, %5, %6\n"
                         "v_sub_u32 %0, %0, %6\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %6\n v_sub_u32 %4, %4, %6\n v_sub_u32 %5, %5, %6"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "v"(b));
        } else if (MODE == 3) {   // grouped in runs of 2 full-rate ops: pk pk pk pk sub sub
#pragma unroll
            for (int i = 0; i < 6; i += 2)
                asm volatile("v_pk_max_i16 %0, %0, %2\n v_pk_max_i16 %1, %1, %2\n v_pk_max_i16 %0, %0, %2\n v_pk_max_i16 %1, %1, %2\n v_sub_u32 %0, %0, %2\n v_sub_u32 %1, %1, %2"
                             : "+v"(a[i]), "+v"(a[i + 1]) : "v"(b));
        }
    }
    int s = 0; for (int i = 0; i < 6; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks, int* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, r); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double trips = 5.0 * blocks * 4.0 * ITER;
    printf("%-52s waves/SIMD=%d  %6.2f cycles per 18-instruction trip\n", name, blocks / 256, (ms * 1e-3) * 2.39e9 * 1024.0 / trips);
}
int main()
{
    int* d; hipMalloc(&d, 256 * 8192 * 4 * 4);
    for (int blocks : {256 * 2, 256 * 3, 256 * 5, 256 * 8}) {
        run<0>("interleaved (pk pk sub) x 6", blocks, d);
        run<3>("runs of two: (pk pk pk pk sub sub) x 3", blocks, d);
        run<1>("grouped: 12 pk then 6 sub", blocks, d);
        run<2>("18 sub (all full rate)", blocks, d);
    }
    return 0;
}
