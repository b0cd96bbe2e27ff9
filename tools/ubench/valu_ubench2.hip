#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 1024
#define NACC 8
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k0(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k1(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k2(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_i32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k3(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_i32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k4(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_min_i32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k5(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k6(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k7(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k8(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k9(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k10(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k11(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k12(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k13(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k14(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k15(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k16(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k17(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k18(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k19(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_i16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k20(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k21(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k22(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_sub_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k23(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_f16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k24(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_f16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k25(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k26(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k27(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k28(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k29(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k30(int* out, int seed) {
  v2i a[NACC]; for (int i=0;i<NACC;++i) { a[i].x = seed + threadIdx.x*(i+1); a[i].y = seed*7 + i; } v2i b; b.x = seed*3+1; b.y = seed+5;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i].x ^ a[i].y; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k31(int* out, int seed) {
  v2i a[NACC]; for (int i=0;i<NACC;++i) { a[i].x = seed + threadIdx.x*(i+1); a[i].y = seed*7 + i; } v2i b; b.x = seed*3+1; b.y = seed+5;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i].x ^ a[i].y; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k32(int* out, int seed) {
  v2i a[NACC]; for (int i=0;i<NACC;++i) { a[i].x = seed + threadIdx.x*(i+1); a[i].y = seed*7 + i; } v2i b; b.x = seed*3+1; b.y = seed+5;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i].x ^ a[i].y; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k33(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k34(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k35(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20","s21");
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k36(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20","s21");
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k37(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k38(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_bfe_i32 %0, %0, 1, 3" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k39(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k40(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_mad_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k41(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i+1)%NACC]));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k42(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i+1)%NACC]));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k43(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i+1)%NACC]));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }
__global__ __launch_bounds__(256) void k44(int* out, int seed) {
  int a[NACC]; for (int i=0;i<NACC;++i) a[i] = seed + threadIdx.x*(i+1); int b = seed*3+1;
  for (int it=0; it<ITER; ++it) {
  _Pragma("unroll") for (int i=0;i<NACC;++i) {
    asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b));
  } }
  int s=0; for (int i=0;i<NACC;++i) s ^= a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s; }

typedef void (*kfn)(int*, int);
static double run(const char* name, kfn f, int blocks, int* d) {
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 1); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r=0;r<5;++r) hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double inst = 5.0*blocks*4.0*ITER*NACC;   // wave-instructions
  double cyc = (ms*1e-3)*2.39e9*1024.0/inst; // SIMD-cycles per wave-instruction (at 2.39 GHz)
  printf("%-26s blocks=%5d  %6.2f cycles/wave-instr  (%6.2f T lane-ops/s)\n", name, blocks, cyc, inst*64/(ms*1e-3)/1e12);
  return cyc; }
int main(){ int* d; hipMalloc(&d, 256*8192*4*4);
 for (int blocks : {2048}) {
  run("v_add_u32_e32", k0, blocks, d);
  run("v_sub_u32_e32", k1, blocks, d);
  run("v_max_i32_e32", k2, blocks, d);
  run("v_max_i32_e64", k3, blocks, d);
  run("v_min_i32_e32", k4, blocks, d);
  run("v_max_u32_e32", k5, blocks, d);
  run("v_and_b32_e32", k6, blocks, d);
  run("v_lshlrev_b32_e32", k7, blocks, d);
  run("v_max_f32_e32", k8, blocks, d);
  run("v_add_f32_e32", k9, blocks, d);
  run("v_sub_f32_e32", k10, blocks, d);
  run("v_mul_f32_e32", k11, blocks, d);
  run("v_fma_f32", k12, blocks, d);
  run("v_fmac_f32_e32", k13, blocks, d);
  run("v_max3_f32", k14, blocks, d);
  run("v_max3_i32", k15, blocks, d);
  run("v_med3_i32", k16, blocks, d);
  run("v_add3_u32", k17, blocks, d);
  run("v_add_f32_e64_abs", k18, blocks, d);
  run("v_max_i16_e32", k19, blocks, d);
  run("v_max_u16_e32", k20, blocks, d);
  run("v_add_u16_e32", k21, blocks, d);
  run("v_sub_u16_e32", k22, blocks, d);
  run("v_max_f16_e32", k23, blocks, d);
  run("v_add_f16_e32", k24, blocks, d);
  run("v_pk_add_f16", k25, blocks, d);
  run("v_pk_max_f16", k26, blocks, d);
  run("v_pk_add_u16", k27, blocks, d);
  run("v_pk_max_u16", k28, blocks, d);
  run("v_pk_max_i16", k29, blocks, d);
  run("v_pk_add_f32", k30, blocks, d);
  run("v_pk_mul_f32", k31, blocks, d);
  run("v_pk_fma_f32", k32, blocks, d);
  run("v_max_i16_sdwa_w1", k33, blocks, d);
  run("v_add_u32_sdwa", k34, blocks, d);
  run("v_cndmask_sgpr", k35, blocks, d);
  run("v_cmp_eq_u32_e64_sgpr", k36, blocks, d);
  run("v_perm_b32", k37, blocks, d);
  run("v_bfe_i32", k38, blocks, d);
  run("v_mad_i32_i24", k39, blocks, d);
  run("v_mad_u16", k40, blocks, d);
  run("v_mov_dpp_row_shr1", k41, blocks, d);
  run("v_add_u32_dpp_row_shr1", k42, blocks, d);
  run("v_mov_dpp_wave_shr1", k43, blocks, d);
  run("ds_bpermute(lds xbar)", k44, blocks, d);
 } return 0; }
