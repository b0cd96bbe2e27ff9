// valu_mix.hip -- issue cost of MIXED half-rate / full-rate VALU streams on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 1024
template <int MODE>
__global__ __launch_bounds__(256) void k(int* out, int seed)
{
    int a[8]; for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    int b = seed * 3 + 1;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            if (MODE == 1) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            if (MODE == 2) { asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            if (MODE == 3) { asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[(i + 4) & 7]) : "v"(b)); }
            if (MODE == 4) { asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
        }
    }
    int s = 0; for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int blocks, int* d, int per_iter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, r); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = 5.0 * blocks * 4.0 * ITER * 8 * per_iter;
    printf("%-44s waves/SIMD=%d  %5.2f cycles per wave-instruction\n", name, blocks / 256, (ms * 1e-3) * 2.39e9 * 1024.0 / inst);
}
int main()
{
    int* d; hipMalloc(&d, 256 * 8192 * 4 * 4);
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        run<0>("pk_max, pk_max (same chain)", blocks, d, 2);
        run<1>("sub_u32, sub_u32 (same chain)", blocks, d, 2);
        run<2>("pk_max, sub_u32 (same chain)", blocks, d, 2);
        run<3>("pk_max, sub_u32 (different chains)", blocks, d, 2);
        run<4>("pk_max, pk_max, sub_u32 (same chain)", blocks, d, 3);
    }
    return 0;
}
