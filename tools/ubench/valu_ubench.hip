// valu_ubench.hip -- per-instruction VALU throughput on gfx950 for the ops the DP kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 valu_ubench.hip -o valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef short s16x2 __attribute__((ext_vector_type(2)));

#define ITER 2048
#define NACC 8

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int seed)
{
    int a[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = seed + threadIdx.x * (i + 1);
    int b = seed * 3 + 1;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % NACC]));
            if (OP == 6) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % NACC]));
            if (OP == 7) asm volatile("v_pk_mad_i16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == 11) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_mov_b32_dpp %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % NACC]));
            if (OP == 13) asm volatile("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % NACC]));
            if (OP == 14) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) % NACC]));
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int blocks, int* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, 256>>>(d, r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 5.0 * blocks * 256.0 * ITER * NACC;
    double t = ops / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%5d  %8.2f T lane-ops/s  (%.1f%% of 78.6)\n", name, blocks, t, 100 * t / 78.64);
    return t;
}

__global__ void clk(unsigned long long* out)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int x = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = x; }
}

int main()
{
    int* d; hipMalloc(&d, 256 * 8192 * 4 * 4);
    for (int blocks : {256 * 2, 256 * 8}) {   // 2 and 8 waves per SIMD
        run<0>("v_pk_max_i16", blocks, d);
        run<1>("v_pk_add_i16", blocks, d);
        run<8>("v_pk_sub_i16", blocks, d);
        run<14>("v_pk_min_i16", blocks, d);
        run<7>("v_pk_mad_i16", blocks, d);
        run<2>("v_max_i32", blocks, d);
        run<3>("v_max3_i32", blocks, d);
        run<4>("v_add_u32", blocks, d);
        run<9>("v_xor_b32", blocks, d);
        run<15>("v_add_i16", blocks, d);
        run<16>("v_max_i16", blocks, d);
        run<10>("v_cndmask_b32", blocks, d);
        run<11>("v_cmp_eq_u32", blocks, d);
        run<17>("v_mov_b32", blocks, d);
        run<5>("v_mov_dpp wave_shr:1", blocks, d);
        run<12>("v_mov_dpp wave_rol:1", blocks, d);
        run<6>("v_mov_dpp row_shr:1", blocks, d);
        run<13>("v_max_i32_dpp row_shr:1", blocks, d);
    }
    unsigned long long* c; hipMalloc(&c, 64);
    clk<<<256 * 8, 256>>>(c);
    hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
    printf("clock under VALU load: %.3f GHz (memtime %llu / realtime %llu @100MHz)\n", (double)h[0] / h[1] * 0.1, h[0], h[1]);
    return 0;
}
