// What one instruction costs a wavefront that has its SIMD to itself, on gfx950:
// chains of dependent instructions against independent ones, for the kinds of
// instruction the Potts step and the trace-back walk are made of.  Clock ticks
// (s_memtime) per instruction, one wavefront per workgroup, one workgroup per launch
// slot so that nothing else shares the SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/dep_latency tools/dep_latency.hip && tools/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2048;

template <int KIND>
__global__ void chain_kernel(float *out, unsigned long long *ticks, float seed) {
  float x = seed + threadIdx.x, y0 = x + 1, y1 = x + 2, y2 = x + 3, y3 = x + 4, y4 = x + 5, y5 = x + 6, y6 = x + 7, y7 = x + 8;
  int c = threadIdx.x & 31;
  unsigned kw = 0x55555555u * (threadIdx.x + 1), i1 = threadIdx.x & 31;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; it++) {
    if constexpr (KIND == 0) {  // 8 dependent adds
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(x) : "v"(seed));
    } else if constexpr (KIND == 1) {  // 8 independent adds
      asm volatile("v_add_f32_e32 %0, %0, %8\n\tv_add_f32_e32 %1, %1, %8\n\tv_add_f32_e32 %2, %2, %8\n\tv_add_f32_e32 %3, %3, %8\n\t"
                   "v_add_f32_e32 %4, %4, %8\n\tv_add_f32_e32 %5, %5, %8\n\tv_add_f32_e32 %6, %6, %8\n\tv_add_f32_e32 %7, %7, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    } else if constexpr (KIND == 2) {  // 8 dependent cross-lane minima (quad swap), each with its two wait states
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
    } else if constexpr (KIND == 3) {  // 8 dependent row mirrors
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
    } else if constexpr (KIND == 4) {  // 8 x (copy, row swap, minimum)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float t;
        asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_min_f32_e32 %0, %0, %1" : "+v"(x), "=&v"(t));
      }
    } else if constexpr (KIND == 5) {  // 8 walk steps: bit test, compare, select
#pragma unroll
      for (int j = 0; j < 8; j++) {
        unsigned b;
        asm volatile("v_bfe_u32 %1, %2, %0, 1\n\tv_cmp_eq_u32_e32 vcc, 0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(c), "=&v"(b) : "v"(kw), "v"(i1) : "vcc");
      }
    } else if constexpr (KIND == 6) {  // 8 walk steps: signed bit, bitwise select
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int m;
        asm volatile("v_bfe_i32 %1, %2, %0, 1\n\tv_bfi_b32 %0, %1, %0, %3" : "+v"(c), "=&v"(m) : "v"(kw), "v"(i1));
      }
    } else {  // one full 32-lane minimum as the Potts step does it, then the median and the add that close the step
      float r, t;
      asm volatile("v_add_f32_e32 %0, %2, %3\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_min_f32_e32 %0, %0, %1\n\t"
                   "v_med3_f32 %2, %2, %0, %4\n\tv_add_f32_e32 %2, %2, %3"
                   : "=&v"(r), "=&v"(t), "+v"(x) : "v"(seed), "v"(-__builtin_inff()));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = x + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + (float)c;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *what, int per_iter, float *d_out, unsigned long long *d_t) {
  hipLaunchKernelGGL(chain_kernel<KIND>, dim3(64), dim3(64), 0, 0, d_out, d_t, 0.5f);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(chain_kernel<KIND>, dim3(64), dim3(64), 0, 0, d_out, d_t, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(64);
  hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  s /= 64;
  printf("%-58s %7.2f ticks each (%d per iteration), %7.1f ns per iteration\n", what, s / kIters / per_iter, per_iter, 1e6 * ms / kIters);
}

int main() {
  float *d_out; unsigned long long *d_t;
  if (hipMalloc((void **)&d_out, 64 * 64 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
  hipMalloc((void **)&d_t, 64 * sizeof(unsigned long long));
  run<0>("dependent v_add_f32", 8, d_out, d_t);
  run<1>("independent v_add_f32", 8, d_out, d_t);
  run<2>("dependent v_min_f32_dpp quad_perm (+ s_nop 1)", 8, d_out, d_t);
  run<3>("dependent v_min_f32_dpp row_mirror (+ s_nop 1)", 8, d_out, d_t);
  run<4>("dependent copy, v_permlane16_swap, minimum", 8, d_out, d_t);
  run<5>("walk step: v_bfe_u32, v_cmp, v_cndmask", 8, d_out, d_t);
  run<6>("walk step: v_bfe_i32, v_bfi_b32", 8, d_out, d_t);
  run<7>("Potts chain: add, 32-lane minimum, median, add", 1, d_out, d_t);
  hipFree(d_out); hipFree(d_t);
  return 0;
}
