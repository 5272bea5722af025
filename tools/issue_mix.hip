// How the wavefronts on one SIMD of gfx950 share its issue slots: W wavefronts per SIMD (one
// 64-thread workgroup each, 1024 x W workgroups; the placement is read back from HW_ID and
// printed), each running trips of 8 independent instructions of one kind.  Prints clock ticks
// per trip and wavefront for W = 1, 2, 4: a wavefront alone issues one instruction per 4.8 ticks
// whatever its kind; with two or more on a SIMD most vector instructions (packed add, two- and
// three-input minimum, select, bit-field insert, lane read) cost 4.4 ticks of the SIMD each,
// v_add_f32 2.4, and a 16-byte LDS read handed to all 64 lanes 16 to 32.
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_mix tools/issue_mix.hip && tools/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

constexpr int kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(64) void mix_kernel(float *out, unsigned long long *ticks, unsigned long long *where, float seed, int sseed) {
  float y0 = seed + threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  f2 p0 = {y0, y1}, p1 = {y2, y3}, p2 = {y4, y5}, p3 = {y6, y7}, pseed = {seed, seed};
  f4 q0, q1, q2, q3, q4, q5, q6, q7;
  __shared__ float s_buf[64];
  s_buf[threadIdx.x] = y0;
  const unsigned ldsaddr = (unsigned)(size_t)s_buf;
  double d0 = y0, d1 = y1, d2 = y2, d3 = y3, dseed = seed, d2in = seed + 1.0;
  const f2 pone = {1.0f, 1.0f};
  const unsigned long long msk = 0x5555555555555555ull * (unsigned)sseed;
  const unsigned vmsk = (threadIdx.x & 1) ? ~0u : 0u;
  int s0 = sseed, s1 = sseed + 1, s2 = sseed + 2, s3 = sseed + 3, s4 = sseed + 4, s5 = sseed + 5, s6 = sseed + 6, s7 = sseed + 7;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; it++) {
    if constexpr (KIND == 0 || KIND == 2) {
      asm volatile("v_add_f32_e32 %0, %0, %8\n\tv_add_f32_e32 %1, %1, %8\n\tv_add_f32_e32 %2, %2, %8\n\tv_add_f32_e32 %3, %3, %8\n\t"
                   "v_add_f32_e32 %4, %4, %8\n\tv_add_f32_e32 %5, %5, %8\n\tv_add_f32_e32 %6, %6, %8\n\tv_add_f32_e32 %7, %7, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    }
    if constexpr (KIND == 1 || KIND == 2) {
      asm volatile("s_add_i32 %0, %0, %8\n\ts_add_i32 %1, %1, %8\n\ts_add_i32 %2, %2, %8\n\ts_add_i32 %3, %3, %8\n\t"
                   "s_add_i32 %4, %4, %8\n\ts_add_i32 %5, %5, %8\n\ts_add_i32 %6, %6, %8\n\ts_add_i32 %7, %7, %8"
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(sseed) : "scc");
    }
    if constexpr (KIND == 4) {  // 8 independent packed adds
      asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                   "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pseed));
    }
    if constexpr (KIND == 5) {  // 8 independent three-input minima
      asm volatile("v_min3_f32 %0, %0, %8, %1\n\tv_min3_f32 %1, %1, %8, %2\n\tv_min3_f32 %2, %2, %8, %3\n\tv_min3_f32 %3, %3, %8, %4\n\t"
                   "v_min3_f32 %4, %4, %8, %5\n\tv_min3_f32 %5, %5, %8, %6\n\tv_min3_f32 %6, %6, %8, %7\n\tv_min3_f32 %7, %7, %8, %0"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    }
    if constexpr (KIND == 6) {  // 8 LDS reads of 16 bytes, every lane the same address, then the wait
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                   "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(ldsaddr) : "memory");
      y0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
    }
    if constexpr (KIND == 7) {  // 8 lane reads into scalar registers
      asm volatile("v_readlane_b32 %0, %8, 0\n\tv_readlane_b32 %1, %8, 1\n\tv_readlane_b32 %2, %8, 2\n\tv_readlane_b32 %3, %8, 3\n\t"
                   "v_readlane_b32 %4, %8, 4\n\tv_readlane_b32 %5, %8, 5\n\tv_readlane_b32 %6, %8, 6\n\tv_readlane_b32 %7, %8, 7"
                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(y0));
    }
    if constexpr (KIND == 8) {  // 8 selects on one mask
      asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %8, vcc\n\tv_cndmask_b32_e32 %2, %2, %8, vcc\n\tv_cndmask_b32_e32 %3, %3, %8, vcc\n\t"
                   "v_cndmask_b32_e32 %4, %4, %8, vcc\n\tv_cndmask_b32_e32 %5, %5, %8, vcc\n\tv_cndmask_b32_e32 %6, %6, %8, vcc\n\tv_cndmask_b32_e32 %7, %7, %8, vcc"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed) : "vcc");
    }
    if constexpr (KIND == 9) {  // 8 LDS reads of 4 bytes, every lane the same address, then the wait
      asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:16\n\tds_read_b32 %2, %8 offset:32\n\tds_read_b32 %3, %8 offset:48\n\t"
                   "ds_read_b32 %4, %8 offset:64\n\tds_read_b32 %5, %8 offset:80\n\tds_read_b32 %6, %8 offset:96\n\tds_read_b32 %7, %8 offset:112\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3), "=v"(y4), "=v"(y5), "=v"(y6), "=v"(y7) : "v"(ldsaddr) : "memory");
    }
    if constexpr (KIND == 10) {  // 4 x (three-input minimum, packed add): two streams that do not depend on each other
      asm volatile("v_pk_add_f32 %0, %0, %8\n\tv_min3_f32 %4, %4, %9, %5\n\tv_pk_add_f32 %1, %1, %8\n\tv_min3_f32 %5, %5, %9, %6\n\t"
                   "v_pk_add_f32 %2, %2, %8\n\tv_min3_f32 %6, %6, %9, %7\n\tv_pk_add_f32 %3, %3, %8\n\tv_min3_f32 %7, %7, %9, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(pseed), "v"(seed));
    }
    if constexpr (KIND == 11) {  // 8 selects on a mask in a scalar register pair
      asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\t"
                   "v_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "s"(msk));
    }
    if constexpr (KIND == 12) {  // 8 bitwise selects on a mask in a vector register
      asm volatile("v_bfi_b32 %0, %9, %0, %8\n\tv_bfi_b32 %1, %9, %1, %8\n\tv_bfi_b32 %2, %9, %2, %8\n\tv_bfi_b32 %3, %9, %3, %8\n\t"
                   "v_bfi_b32 %4, %9, %4, %8\n\tv_bfi_b32 %5, %9, %5, %8\n\tv_bfi_b32 %6, %9, %6, %8\n\tv_bfi_b32 %7, %9, %7, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "v"(vmsk));
    }
    if constexpr (KIND == 13) {  // 8 two-input minima
      asm volatile("v_min_f32_e32 %0, %0, %8\n\tv_min_f32_e32 %1, %1, %8\n\tv_min_f32_e32 %2, %2, %8\n\tv_min_f32_e32 %3, %3, %8\n\t"
                   "v_min_f32_e32 %4, %4, %8\n\tv_min_f32_e32 %5, %5, %8\n\tv_min_f32_e32 %6, %6, %8\n\tv_min_f32_e32 %7, %7, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
    }
    if constexpr (KIND == 14) {  // 8 compares into scalar register pairs
      unsigned long long m0, m1, m2, m3;
      asm volatile("v_cmp_eq_f32_e64 %0, %4, %5\n\tv_cmp_eq_f32_e64 %1, %4, %6\n\tv_cmp_eq_f32_e64 %2, %4, %7\n\tv_cmp_eq_f32_e64 %3, %4, %8\n\t"
                   "v_cmp_eq_f32_e64 %0, %5, %6\n\tv_cmp_eq_f32_e64 %1, %5, %7\n\tv_cmp_eq_f32_e64 %2, %5, %8\n\tv_cmp_eq_f32_e64 %3, %6, %7"
                   : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(y4));
      s0 += (int)m0 + (int)m1 + (int)m2 + (int)m3;
    }
    if constexpr (KIND == 15) {  // 4 x (compare into vcc, select on it two instructions later)
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %4\n\tv_add_f32_e32 %1, %1, %4\n\tv_add_f32_e32 %2, %2, %4\n\tv_cndmask_b32_e32 %3, %3, %4, vcc\n\t"
                   "v_cmp_lt_f32_e32 vcc, %1, %4\n\tv_add_f32_e32 %0, %0, %4\n\tv_add_f32_e32 %2, %2, %4\n\tv_cndmask_b32_e32 %3, %3, %4, vcc"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(seed) : "vcc");
    }
    if constexpr (KIND == 16) {  // 8 conversions int32 -> binary32
      asm volatile("v_cvt_f32_i32_e32 %0, %0\n\t"
                   "v_cvt_f32_i32_e32 %1, %1\n\t"
                   "v_cvt_f32_i32_e32 %2, %2\n\t"
                   "v_cvt_f32_i32_e32 %3, %3\n\t"
                   "v_cvt_f32_i32_e32 %4, %4\n\t"
                   "v_cvt_f32_i32_e32 %5, %5\n\t"
                   "v_cvt_f32_i32_e32 %6, %6\n\t"
                   "v_cvt_f32_i32_e32 %7, %7"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "v"(vmsk));
    }
    if constexpr (KIND == 17) {  // 8 x (conversion binary32 -> binary64, binary64 add)
      asm volatile("v_cvt_f64_f32_e32 %0, %4\n\tv_add_f64 %2, %2, %0\n\tv_cvt_f64_f32_e32 %1, %5\n\tv_add_f64 %3, %3, %1\n\t"
                   "v_cvt_f64_f32_e32 %0, %6\n\tv_add_f64 %2, %2, %0\n\tv_cvt_f64_f32_e32 %1, %7\n\tv_add_f64 %3, %3, %1"
                   : "=&v"(d0), "=&v"(d1), "+v"(d2), "+v"(d3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
    }
    if constexpr (KIND == 18) {  // 8 binary64 adds
      asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t"
                   "v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dseed));
    }
    if constexpr (KIND == 19) {  // 8 conversions binary32 -> binary64
      asm volatile("v_cvt_f64_f32_e32 %0, %4\n\tv_cvt_f64_f32_e32 %1, %5\n\tv_cvt_f64_f32_e32 %2, %6\n\tv_cvt_f64_f32_e32 %3, %7\n\t"
                   "v_cvt_f64_f32_e32 %0, %5\n\tv_cvt_f64_f32_e32 %1, %6\n\tv_cvt_f64_f32_e32 %2, %7\n\tv_cvt_f64_f32_e32 %3, %4"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));
      y4 += (float)d0 + (float)d1 + (float)d2 + (float)d3;
    }
    if constexpr (KIND == 20) {  // 8 64-bit moves
      asm volatile("v_mov_b64 %0, %4\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %4\n\tv_mov_b64 %3, %4\n\t"
                   "v_mov_b64 %0, %5\n\tv_mov_b64 %1, %5\n\tv_mov_b64 %2, %5\n\tv_mov_b64 %3, %5"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(dseed), "v"(d2in));
      y4 += (float)d0 + (float)d1 + (float)d2 + (float)d3;
    }
    if constexpr (KIND == 21) {  // 8 24-bit multiply-adds
      asm volatile("v_mad_i32_i24 %0, %0, %9, %8\n\t"
                   "v_mad_i32_i24 %1, %1, %9, %8\n\t"
                   "v_mad_i32_i24 %2, %2, %9, %8\n\t"
                   "v_mad_i32_i24 %3, %3, %9, %8\n\t"
                   "v_mad_i32_i24 %4, %4, %9, %8\n\t"
                   "v_mad_i32_i24 %5, %5, %9, %8\n\t"
                   "v_mad_i32_i24 %6, %6, %9, %8\n\t"
                   "v_mad_i32_i24 %7, %7, %9, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "v"(vmsk));
    }
    if constexpr (KIND == 22) {  // 8 three-input medians
      asm volatile("v_med3_f32 %0, %0, %8, %9\n\t"
                   "v_med3_f32 %1, %1, %8, %9\n\t"
                   "v_med3_f32 %2, %2, %8, %9\n\t"
                   "v_med3_f32 %3, %3, %8, %9\n\t"
                   "v_med3_f32 %4, %4, %8, %9\n\t"
                   "v_med3_f32 %5, %5, %8, %9\n\t"
                   "v_med3_f32 %6, %6, %8, %9\n\t"
                   "v_med3_f32 %7, %7, %8, %9"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "v"(vmsk));
    }
    if constexpr (KIND == 23) {  // 8 packed multiplies
      asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                   "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pone));
    }
    if constexpr (KIND == 24) {  // 8 multiplies
      asm volatile("v_mul_f32_e32 %0, %0, %8\n\t"
                   "v_mul_f32_e32 %1, %1, %8\n\t"
                   "v_mul_f32_e32 %2, %2, %8\n\t"
                   "v_mul_f32_e32 %3, %3, %8\n\t"
                   "v_mul_f32_e32 %4, %4, %8\n\t"
                   "v_mul_f32_e32 %5, %5, %8\n\t"
                   "v_mul_f32_e32 %6, %6, %8\n\t"
                   "v_mul_f32_e32 %7, %7, %8"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed), "v"(vmsk));
    }
    if constexpr (KIND == 25) {  // 8 cross-lane minima two instructions apart (DPP quad swap), adds between
      asm volatile("v_min_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %5, %5, %8\n\tv_add_f32_e32 %6, %6, %8\n\t"
                   "v_min_f32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %7, %7, %8\n\tv_add_f32_e32 %5, %5, %8\n\t"
                   "v_min_f32_dpp %2, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_e32 %6, %6, %8\n\tv_add_f32_e32 %7, %7, %8\n\t"
                   "v_min_f32_dpp %3, %2, %2 row_mirror row_mask:0xf bank_mask:0xf"
                   : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(seed));
      y4 += y3;
    }
    if constexpr (KIND == 3) {  // interleaved: v s v s ...
      asm volatile("v_add_f32_e32 %0, %0, %16\n\ts_add_i32 %8, %8, %17\n\tv_add_f32_e32 %1, %1, %16\n\ts_add_i32 %9, %9, %17\n\t"
                   "v_add_f32_e32 %2, %2, %16\n\ts_add_i32 %10, %10, %17\n\tv_add_f32_e32 %3, %3, %16\n\ts_add_i32 %11, %11, %17\n\t"
                   "v_add_f32_e32 %4, %4, %16\n\ts_add_i32 %12, %12, %17\n\tv_add_f32_e32 %5, %5, %16\n\ts_add_i32 %13, %13, %17\n\t"
                   "v_add_f32_e32 %6, %6, %16\n\ts_add_i32 %14, %14, %17\n\tv_add_f32_e32 %7, %7, %16\n\ts_add_i32 %15, %15, %17"
                   : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7),
                     "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                   : "v"(seed), "s"(sseed) : "scc");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  y0 += (float)(d0 + d1 + d2 + d3);
  y0 += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  out[blockIdx.x * 64 + threadIdx.x] = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
  if (threadIdx.x == 0) {
    ticks[blockIdx.x] = t1 - t0;
    // where the wavefront ran: die (XCC_ID, register 20) and the HW_ID word (register 4)
    where[blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  }
}

template <int KIND>
void run(const char *what, int W) {
  const int blocks = 1024 * W;
  float *out; unsigned long long *ticks, *where;
  hipMalloc(&out, sizeof(float) * 64 * blocks);
  hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
  hipMalloc(&where, sizeof(unsigned long long) * blocks);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(mix_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, ticks, where, 1.0f, 3);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks), wh(blocks);
  hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  hipMemcpy(wh.data(), where, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  // wavefronts per SIMD: HW_ID bits 5:4 SIMD, 11:8 compute unit, 12 shader array, 15:13 shader engine
  std::map<unsigned long long, int> per_simd;
  for (int i = 0; i < blocks; i++) {
    const unsigned hw = (unsigned)wh[i];
    const unsigned long long key = ((wh[i] >> 32) & 15) << 16 | ((hw >> 4) & 3) | ((hw >> 8) & 0xFF) << 2;
    per_simd[key]++;
  }
  int lo = 1 << 30, hi = 0;
  for (auto &kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
  std::sort(h.begin(), h.end());
  printf("%-34s W=%d: median %.1f ticks per trip, 90th percentile %.1f; %zu SIMDs used, %d to %d wavefronts on each\n", what, W,
         (double)h[blocks / 2] / kIters, (double)h[blocks * 9 / 10] / kIters, per_simd.size(), lo, hi);
  hipFree(out); hipFree(ticks); hipFree(where);
}

int main() {
  for (int W : {1, 2, 4}) {
    run<0>("8 vector adds", W);
    run<1>("8 scalar adds", W);
    run<2>("8 vector adds, then 8 scalar adds", W);
    run<3>("8 x (vector add, scalar add)", W);
    run<4>("8 packed adds", W);
    run<5>("8 three-input minima", W);
    run<10>("4 x (packed add, 3-input minimum)", W);
    run<11>("8 selects on a scalar pair", W);
    run<12>("8 bitwise selects (v_bfi)", W);
    run<13>("8 two-input minima", W);
    run<14>("8 compares into scalar pairs", W);
    run<15>("2 x (compare, 2 adds, select)", W);
    run<7>("8 lane reads", W);
    run<24>("8 multiplies", W);
    run<23>("8 packed multiplies", W);
    run<22>("8 three-input medians", W);
    run<21>("8 24-bit multiply-adds", W);
    run<16>("8 conversions int32 -> binary32", W);
    run<19>("8 conversions binary32 -> binary64", W);
    run<18>("8 binary64 adds", W);
    run<17>("4 x (convert to binary64, add)", W);
    run<20>("8 64-bit moves", W);
    run<25>("4 DPP minima + 6 adds", W);
    run<6>("8 LDS reads of 16 B + wait", W);
    run<9>("8 LDS reads of 4 B + wait", W);
  }
  return 0;
}
