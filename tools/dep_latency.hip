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
    } else if constexpr (KIND == 8) {  // 8 independent cross-lane adds (row shift)
      asm volatile("v_add_f32_dpp %0, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %1, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %2, %8, %8 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %3, %8, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %4, %8, %8 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %5, %8, %8 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_add_f32_dpp %6, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32_dpp %7, %8, %8 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(y4), "=&v"(y5), "=&v"(y6), "=&v"(y7) : "v"(seed));
    } else if constexpr (KIND == 9) {  // 8 independent compares into scalar register pairs
      unsigned long long m0, m1, m2, m3;
      asm volatile("v_cmp_eq_f32_e64 %0, %4, %5\n\tv_cmp_eq_f32_e64 %1, %4, %6\n\tv_cmp_eq_f32_e64 %2, %4, %7\n\tv_cmp_eq_f32_e64 %3, %4, %8\n\t"
                   "v_cmp_eq_f32_e64 %0, %5, %6\n\tv_cmp_eq_f32_e64 %1, %5, %7\n\tv_cmp_eq_f32_e64 %2, %5, %8\n\tv_cmp_eq_f32_e64 %3, %6, %7"
                   : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(seed), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
      c += (int)(m0 + m1 + m2 + m3 == 77);
    } else if constexpr (KIND == 10) {  // 4 compares and the 4 selects on them, three instructions apart
      unsigned long long m0, m1, m2;
      asm volatile("v_cmp_eq_f32_e64 %1, %4, %5\n\tv_cmp_eq_f32_e64 %2, %4, %6\n\tv_cmp_eq_f32_e64 %3, %4, %7\n\t"
                   "v_cndmask_b32_e64 %0, %0, 3, %1\n\tv_cmp_eq_f32_e64 %1, %4, %8\n\tv_cndmask_b32_e64 %0, %0, 2, %2\n\t"
                   "v_cndmask_b32_e64 %0, %0, 1, %3\n\tv_cndmask_b32_e64 %0, %0, 0, %1"
                   : "+v"(c), "=&s"(m0), "=&s"(m1), "=&s"(m2) : "v"(seed), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    } else if constexpr (KIND == 11) {  // 8 independent three-way minima
      asm volatile("v_min3_f32 %0, %0, %8, %1\n\tv_min3_f32 %1, %1, %8, %2\n\tv_min3_f32 %2, %2, %8, %3\n\tv_min3_f32 %3, %3, %8, %4\n\t"
                   "v_min3_f32 %4, %4, %8, %5\n\tv_min3_f32 %5, %5, %8, %6\n\tv_min3_f32 %6, %6, %8, %7\n\tv_min3_f32 %7, %7, %8, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    } else if constexpr (KIND == 12) {  // 8 independent 64-bit shifts of a scalar pair
      unsigned long long a0, a1, a2, a3;
      const unsigned long long sm = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
      asm volatile("v_lshrrev_b64 %0, %4, %5\n\tv_lshrrev_b64 %1, %4, %5\n\tv_lshrrev_b64 %2, %4, %5\n\tv_lshrrev_b64 %3, %4, %5\n\t"
                   "v_lshrrev_b64 %0, %4, %5\n\tv_lshrrev_b64 %1, %4, %5\n\tv_lshrrev_b64 %2, %4, %5\n\tv_lshrrev_b64 %3, %4, %5"
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(c), "s"(sm));
      kw += (unsigned)(a0 + a1 + a2 + a3);
    } else if constexpr (KIND == 13) {  // 8 wait states, one instruction each
      asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
    } else if constexpr (KIND == 14) {  // 8 dependent scalar adds
      int sa = it;
      asm volatile("s_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\t"
                   "s_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1\n\ts_add_i32 %0, %0, 1" : "+s"(sa) : : "scc");
      c += sa & 1;
    } else if constexpr (KIND == 15) {  // 8 independent vector adds with a scalar add behind each
      int sa = it;
      asm volatile("v_add_f32_e32 %0, %0, %9\n\ts_add_i32 %8, %8, 1\n\tv_add_f32_e32 %1, %1, %9\n\ts_add_i32 %8, %8, 1\n\t"
                   "v_add_f32_e32 %2, %2, %9\n\ts_add_i32 %8, %8, 1\n\tv_add_f32_e32 %3, %3, %9\n\ts_add_i32 %8, %8, 1\n\t"
                   "v_add_f32_e32 %4, %4, %9\n\ts_add_i32 %8, %8, 1\n\tv_add_f32_e32 %5, %5, %9\n\ts_add_i32 %8, %8, 1\n\t"
                   "v_add_f32_e32 %6, %6, %9\n\ts_add_i32 %8, %8, 1\n\tv_add_f32_e32 %7, %7, %9\n\ts_add_i32 %8, %8, 1"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7), "+s"(sa) : "v"(seed) : "scc");
      c += sa & 1;
    } else if constexpr (KIND == 16) {  // 2 dependent adds with 3 independent ones between them
      asm volatile("v_add_f32_e32 %0, %0, %8\n\tv_add_f32_e32 %1, %1, %8\n\tv_add_f32_e32 %2, %2, %8\n\tv_add_f32_e32 %3, %3, %8\n\t"
                   "v_add_f32_e32 %0, %0, %8\n\tv_add_f32_e32 %1, %1, %8\n\tv_add_f32_e32 %2, %2, %8\n\tv_add_f32_e32 %3, %3, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    } else if constexpr (KIND == 17) {  // 8 independent packed adds
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {y0, y1}, p1 = {y2, y3}, p2 = {y4, y5}, p3 = {y6, y7}, sd = {seed, seed};
      asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                   "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(sd));
      y0 = p0.x; y1 = p0.y; y2 = p1.x; y3 = p1.y; y4 = p2.x; y5 = p2.y; y6 = p3.x; y7 = p3.y;
    } else if constexpr (KIND == 18 || KIND == 19) {  // the Potts step with the masks of the step before threaded through it
      float cc, r, t;
      unsigned long long m_lt, m_le, m_eq;
      static unsigned long long vk, ve;
      float pc = y0, pC = y1, pM = y2;
      if constexpr (KIND == 18) {
      asm volatile(
                   "v_add_f32_e32 %[c], %[M], %[wp]\n\t"
                   "v_cmp_eq_f32_e32 vcc, %[pc], %[pC]\n\t"
                   "v_cmp_lt_f32_e64 %[LT], %[pM], %[pC]\n\t"
                   "v_min_f32_dpp %[r], %[c], %[c] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_cmp_le_f32_e64 %[LE], %[pM], %[pC]\n\t"
                   "s_mov_b64 %[E], vcc\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "s_sub_u32 vcc_lo, vcc_lo, 1\n\t"
                   "s_subb_u32 vcc_hi, vcc_hi, 1\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "s_andn2_b64 vcc, vcc, %[E]\n\t"
                   "s_and_b64 vcc, vcc, %[LE]\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "s_or_b64 %[LE], vcc, %[LT]\n\t"
                   "s_bfm_b64 exec, 1, %[ls]\n\t"
                   "v_mov_b64 %[vk], %[LE]\n\t"
                   "v_mov_b64 %[ve], %[E]\n\t"
                   "s_mov_b64 exec, -1\n\t"
                   "v_mov_b32 %[t], %[r]\n\t"
                   "s_nop 1\n\t"
                   "v_permlane16_swap_b32 %[r], %[t]\n\t"
                   "v_min_f32_e32 %[pC], %[r], %[t]\n\t"
                   "v_med3_f32 %[t], %[M], %[pC], %[pad]\n\t"
                   "v_mov_b32 %[pM], %[M]\n\t"
                   "v_mov_b32 %[pc], %[c]\n\t"
                   "v_add_f32_e32 %[M], %[th], %[t]"
                   : [c] "=&v"(cc), [r] "=&v"(r), [t] "=&v"(t), [LT] "=&s"(m_lt), [LE] "=&s"(m_le), [E] "=&s"(m_eq), [vk] "+v"(vk), [ve] "+v"(ve),
                     [M] "+v"(x), [pc] "+v"(pc), [pC] "+v"(pC), [pM] "+v"(pM)
                   : [wp] "v"(seed), [th] "v"(y3), [pad] "v"(-__builtin_inff()), [ls] "s"(it & 63)
                   : "vcc", "scc");
      } else {  // the vector instructions alone
      asm volatile("v_add_f32_e32 %[c], %[M], %[wp]\n\t"
                   "v_min_f32_dpp %[r], %[c], %[c] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\t"
                   "v_min_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_mov_b32 %[t], %[r]\n\t"
                   "s_nop 1\n\t"
                   "v_permlane16_swap_b32 %[r], %[t]\n\t"
                   "v_min_f32_e32 %[pC], %[r], %[t]\n\t"
                   "v_med3_f32 %[t], %[M], %[pC], %[pad]\n\t"
                   "v_add_f32_e32 %[M], %[th], %[t]"
                   : [c] "=&v"(cc), [r] "=&v"(r), [t] "=&v"(t), [M] "+v"(x), [pC] "+v"(pC)
                   : [wp] "v"(seed), [th] "v"(y3), [pad] "v"(-__builtin_inff()));
      }
      y0 = pc; y1 = pC; y2 = pM;
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
  run<8>("independent v_add_f32_dpp row shifts", 8, d_out, d_t);
  run<9>("independent v_cmp_eq_f32 into scalar pairs", 8, d_out, d_t);
  run<10>("4 compares + 4 selects, three apart", 8, d_out, d_t);
  run<11>("independent v_min3_f32", 8, d_out, d_t);
  run<12>("independent v_lshrrev_b64", 8, d_out, d_t);
  run<13>("s_nop 0", 8, d_out, d_t);
  run<14>("dependent s_add_i32", 8, d_out, d_t);
  run<15>("v_add_f32 and s_add_i32 alternating (16 instructions)", 16, d_out, d_t);
  run<16>("v_add_f32, dependent at distance four", 8, d_out, d_t);
  run<17>("v_pk_add_f32, dependent at distance four", 8, d_out, d_t);
  run<18>("Potts step, masks of the step before threaded through", 1, d_out, d_t);
  run<19>("Potts step, its chain of vector instructions alone", 1, d_out, d_t);
  run<7>("Potts chain: add, 32-lane minimum, median, add", 1, d_out, d_t);
  hipFree(d_out); hipFree(d_t);
  return 0;
}
