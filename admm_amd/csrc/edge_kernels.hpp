// Edge (Bethe sum-product) solver kernels for gfx950: the reference's own
// subproblem, one tree per edge (Tree.h:20-51).  A group of KP lanes owns one
// edge; lane i holds entry i of the node vectors and walks row / column i of
// the k x k table, so every running sum is accumulated in the same order as
// the reference's scalar loops.  log / exp are evaluated in binary64 (exp: the
// device library's; log: edge_log.h, a quarter of the device library's
// instructions) and rounded to binary32 exactly where the reference stores
// into a float.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "edge_log.h"

namespace admm {

__device__ const LogPiece d_log_table[kLogPieces] = {ADMM_LOG_TABLE};

struct EdgeArgs {
  int E, V, k;
  float penalty, penaltyPrime, rho;
  const int *edgeList;    // [E][2]
  float *mu_node;         // [E][2k]  x_mu_node
  const float *pot_node;  // [E][2k]  x_potential_node
  float *lam_node;        // [E][2k]  lambda_node
  float *mu_edge;         // [E][k*k] x_mu_edge
  const float *pot_edge;  // [E][k*k] x_potential_edge
  const float *m_vert;    // [V][k]   consensus (what edge_mu_node holds)
  double *obj_part;       // [workgroups] linear objective of the new marginals, or null
};

__device__ __forceinline__ void edge_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

constexpr int kEdgeBlock = 256;

// num / den in binary32, bit for bit, for a divisor that many numerators share: den as binary64
// and r = 1 / den (binary64, correctly rounded) are formed once.  q0 = num * r is within one unit
// of the binary64 quotient, one residual step (Markstein) makes it the correctly rounded binary64
// quotient q1, and rounding q1 to binary32 gives the binary32 division's result: the quotient
// of two binary32 values is either representable in binary64 (then q1 is exact, ties between
// two subnormal results included) or at least 2^-50 away, relatively, from every point where the
// rounding to binary32 changes.  With an infinite operand the residual is NaN and q0 already is
// the answer.  A numerator of -0 gives +0 (the callers' numerators are products of non-negative
// values, or the quotient is subtracted from 0).  tools/check_recip_div.cpp runs the same
// sequence on the host against the division: 5e8 pairs, none differ.
__device__ __forceinline__ float edge_divide(float num, double den, double r) {
  const double n = (double)num;
  const double q0 = n * r;
  const double q1 = __builtin_fma(__builtin_fma(-q0, den, n), r, q0);
  return (float)(q1 != q1 ? q0 : q1);
}

// exp of a binary32 argument in binary64, for the three exponentials per table entry
// (Tree.cpp:196, 219, 261).  The device library's algorithm with its coefficients, hence its
// bits: n = rint(x log2 e), r = x - n ln 2 in two pieces, a degree-11 polynomial, ldexp.  The
// library guards the ends of the range with two compares and two selects per call (beyond
// 1024 the result is +inf, below -1075 it is 0); here the ARGUMENT is clamped to [-1076, 1025]
// instead, where ldexp delivers +inf and 0 by itself: one instruction.  A clamp does not pass
// a NaN on, so every call also notes whether its argument was one (`nan`: one bit per lane,
// kept in a scalar register pair; one compare), and the caller turns the lane's sums into
// NaN afterwards, which is what a NaN term makes of them in the reference.
__device__ __forceinline__ double edge_exp(float c, unsigned long long &nan) {
  nan |= __builtin_amdgcn_ballot_w64(c != c);
  const double x = (double)__builtin_amdgcn_fmed3f(c, -1076.0f, 1025.0f);
  auto K = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
  const double n = __builtin_rint(x * K(0x3ff71547652b82feull));
  double r = __builtin_fma(n, K(0xbfe62e42fefa39efull), x);
  r = __builtin_fma(n, K(0xbc7abc9e3b39803full), r);
  double q = __builtin_fma(r, K(0x3e5ade156a5dcb37ull), K(0x3e928af3fca7ab0cull));
  q = __builtin_fma(r, q, K(0x3ec71dee623fde64ull));
  q = __builtin_fma(r, q, K(0x3efa01997c89e6b0ull));
  q = __builtin_fma(r, q, K(0x3f2a01a014761f6eull));
  q = __builtin_fma(r, q, K(0x3f56c16c1852b7b0ull));
  q = __builtin_fma(r, q, K(0x3f81111111122322ull));
  q = __builtin_fma(r, q, K(0x3fa55555555502a1ull));
  q = __builtin_fma(r, q, K(0x3fc5555555555511ull));
  q = __builtin_fma(r, q, K(0x3fe000000000000bull));
  q = __builtin_fma(r, q, 1.0);
  q = __builtin_fma(r, q, 1.0);
  return __builtin_amdgcn_ldexp(q, (int)n);
}

// Values of edge_exp and of the device library's exp on a list of binary32 arguments
// (accuracy and equality tests).
__global__ void edge_exp_values_kernel(double *__restrict__ own, double *__restrict__ lib, unsigned char *__restrict__ was_nan,
                                       const float *__restrict__ x, long long n) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    unsigned long long nan = 0;
    own[idx] = edge_exp(x[idx], nan);
    lib[idx] = exp((double)x[idx]);
    was_nan[idx] = (unsigned char)((nan >> (threadIdx.x & 63)) & 1);
  }
}

// Tree.cpp:64-274 for one edge per KP-lane group.
template <int KP>
__global__ __launch_bounds__(kEdgeBlock) void edge_solve_kernel(const EdgeArgs p) {
  constexpr int EPW = 64 / KP;
  constexpr int GPB = (kEdgeBlock / 64) * EPW;  // edge groups per workgroup
  constexpr int LD = KP + 1;                    // padded row of the k x k scratch
  __shared__ float s_ep[GPB][KP * LD];
  __shared__ float s_np[GPB][2 * KP];
  __shared__ float s_lmu[GPB][KP];  // log mu of the first endpoint (every lane reads every entry)
  __shared__ float s_buf[GPB][KP];
  __shared__ LogPiece s_log[kLogPieces];
  static_assert(kLogPieces * 2 == kEdgeBlock, "half a piece of the table per thread");
  reinterpret_cast<double *>(s_log)[threadIdx.x] = reinterpret_cast<const double *>(d_log_table)[threadIdx.x];
  __syncthreads();

  const int grp = threadIdx.x / KP;
  const int i = threadIdx.x % KP;
  const long long e_raw = (long long)blockIdx.x * GPB + grp;
  const bool evalid = e_raw < p.E;
  const long long e = evalid ? e_raw : p.E - 1;
  const int k = p.k, k2 = k * k;
  const bool iv = i < k;
  const bool wr = evalid && iv;
  const float EPS = 0.00000000000000001;  // Tree.h:15
  const float pen = p.penalty, pp = p.penaltyPrime, rho = p.rho;
  // x / penaltyPrime (Tree.cpp:160, 172) for a power of two (1 by default) is x * (1 / penaltyPrime):
  // the same real number, so the same rounding, without the division's dozen instructions
  const float inv_pp = 1.0f / pp;
  const double rpp = 1.0 / (double)pp;  // for edge_divide
  const bool pp_pow2 = (__float_as_uint(pp) & 0x807fffffu) == 0 && (__float_as_uint(pp) >> 23) != 0 &&
                       (__float_as_uint(pp) >> 23) != 255 && (__float_as_uint(inv_pp) & 0x7f800000u) != 0 &&
                       inv_pp * pp == 1.0f;

  float *ep = s_ep[grp];
  float *np = s_np[grp];
  float *lmu = s_lmu[grp];
  float *buf = s_buf[grp];

  // node vectors: UpdateLambda, AdjustPotentials, UpdateMu, node derivatives
  // and node potentials (Tree.cpp:64-73, 84-86, 48-51, 120-122, 141-143, 157-160)
  float pot_h[2];
  float lmu2 = 0.0f;  // log mu of the second endpoint, this lane's entry
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int v = p.edgeList[2 * e + h];
    float mu = 1.0f, m = 1.0f, lam = 0.0f, pot = 0.0f;
    if (iv) {
      mu = p.mu_node[e * 2 * k + h * k + i];
      m = p.m_vert[(long long)v * k + i];
      lam = p.lam_node[e * 2 * k + h * k + i];
      pot = p.pot_node[e * 2 * k + h * k + i];
    }
    pot_h[h] = pot;
    lam = lam + pen * (mu - m);
    if (wr) p.lam_node[e * 2 * k + h * k + i] = lam;
    const float xnode = rho * (pot + lam) - pen * (m - mu);
    const float muc = ((double)mu <= 0.0) ? EPS : mu;
    const float lv = (float)edge_log((double)muc, s_log);
    const float nd = (float)(-1.0 - (double)lv);
    float val = xnode + pp * nd;
    val = pp_pow2 ? val * inv_pp : val / pp;
    np[h * KP + i] = 0.0f - val;
    if (h == 0) lmu[i] = lv;
    else lmu2 = lv;
  }
  edge_lds_fence();

  // edge derivatives and potentials (Tree.cpp:123-133, 144-150, 98-99, 169-172), kLoadAhead rows
  // of the table at a time: their loads leave together, so a wavefront waits for memory once per
  // batch and not once per row.  Lane b walks column b, which is also what message 1 -> 2
  // (Tree.cpp:191-205: lane b sums over a) reads: its terms are added as the column is formed.
  constexpr int kLoadAhead = 8;
  float acc = 0.0f;
  unsigned long long nan12 = 0;  // lanes that met a NaN argument of an exponential (see edge_exp)
  // how x / penaltyPrime is formed: 0 any value, 1 a power of two, 2 penaltyPrime = rho = 1 (the
  // drivers' values, admm.cpp:44-47: the multiplications by 1 are left out, x * 1 being x)
  auto derivatives = [&](auto how_c) {
    constexpr int HOW = decltype(how_c)::value;  // the branch on it stays outside the loop
    for (int a0 = 0; a0 < k; a0 += kLoadAhead) {
      float mue_b[kLoadAhead], pe_b[kLoadAhead];
#pragma unroll
      for (int u = 0; u < kLoadAhead; u++) {
        const bool ok = iv && a0 + u < k;
        mue_b[u] = ok ? p.mu_edge[e * k2 + (a0 + u) * k + i] : 1.0f;
        pe_b[u] = ok ? p.pot_edge[e * k2 + (a0 + u) * k + i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kLoadAhead; u++) {
        const int a = a0 + u;
        if (a < k) {
          const float mue = mue_b[u], pe = pe_b[u];
          const float muec = ((double)mue <= 0.0) ? EPS : mue;
          float ed = (float)(((double)lmu[a] - edge_log((double)muec, s_log)) + 1.0);
          ed = ed + lmu2;
          float val;
          if constexpr (HOW == 2) {
            val = pe + ed;
          } else {
            const float xe = rho * pe;
            val = xe + pp * ed;
            if constexpr (HOW == 1) val = val * inv_pp;
            else val = edge_divide(val, (double)pp, rpp);
          }
          const float epv = 0.0f - val;
          ep[a * LD + i] = epv;
          const float c = epv + np[a];
          acc = (float)((double)acc + edge_exp(c, nan12));
        }
      }
    }
  };
  if (pp == 1.0f && rho == 1.0f) derivatives(std::integral_constant<int, 2>{});
  else if (pp_pow2) derivatives(std::integral_constant<int, 1>{});
  else derivatives(std::integral_constant<int, 0>{});
  const float qnan = __builtin_nanf("");
  if ((nan12 >> (threadIdx.x & 63)) & 1) acc = qnan;
  edge_lds_fence();
  buf[i] = acc;
  edge_lds_fence();
  float sum = 0.0f;
  for (int j = 0; j < k; j++) sum += buf[j];
  const float m12 = acc / sum;
  edge_lds_fence();

  // message 2 -> 1 (Tree.cpp:212-229): lane a sums over b; the same walk over row a forms the
  // exponentials of the edge marginal and their sum (Tree.cpp:256-268), which need no message
  acc = 0.0f;
  float row_sum = 0.0f;
  const float np_own = np[i];
  unsigned long long nan21 = 0, nan_row = 0;
  for (int b = 0; b < k; b++) {
    const float epv = ep[i * LD + b], n2 = np[KP + b];
    const float c = epv + n2;
    acc = (float)((double)acc + edge_exp(c, nan21));
    float q = epv + np_own;
    q = q + n2;
    const float ex = (float)edge_exp(q, nan_row);
    ep[i * LD + b] = ex;
    row_sum = row_sum + ex;
  }
  // a NaN term makes the lane's message NaN; a NaN exponential of the edge marginal makes the
  // row's sum NaN and with it, in the division below, every entry of the row, as in the reference
  if ((nan21 >> (threadIdx.x & 63)) & 1) acc = qnan;
  if ((nan_row >> (threadIdx.x & 63)) & 1) row_sum = qnan;
  buf[i] = acc;
  edge_lds_fence();
  sum = 0.0f;
  for (int j = 0; j < k; j++) sum += buf[j];
  const float m21 = acc / sum;
  edge_lds_fence();

  // node marginals (Tree.cpp:236-254)
  float t1 = (float)(exp((double)np[i]) * (double)m21);
  buf[i] = t1;
  edge_lds_fence();
  sum = 0.0f;
  for (int j = 0; j < k; j++) sum += buf[j];
  const float mu1 = t1 / sum;
  edge_lds_fence();
  float t2 = (float)(exp((double)np[KP + i]) * (double)m12);
  buf[i] = t2;
  edge_lds_fence();
  sum = 0.0f;
  for (int j = 0; j < k; j++) sum += buf[j];
  const float mu2 = t2 / sum;
  if (wr) {
    p.mu_node[e * 2 * k + i] = mu1;
    p.mu_node[e * 2 * k + k + i] = mu2;
  }

  // edge marginal (Tree.cpp:269-273): lane a owns row a; mu1 * ex / sum for every b with one
  // reciprocal (edge_divide)
  const double accd = (double)row_sum, racc = 1.0 / accd;
  const bool acc_nz = row_sum != 0.0f;
  // a positive finite sum in every lane of the wavefront (then no exponential of the row is
  // infinite either): the residual step cannot meet an infinity, and its guard is left out
  if (__builtin_amdgcn_ballot_w64(!__builtin_amdgcn_classf(row_sum, 0x180)) == 0) {  // +normal or +subnormal
    for (int b = 0; b < k; b++) {
      const double n = (double)(mu1 * ep[i * LD + b]);
      const double q0 = n * racc;
      ep[i * LD + b] = (float)__builtin_fma(__builtin_fma(-q0, accd, n), racc, q0);
    }
  } else {
    for (int b = 0; b < k; b++) {
      const float ex = ep[i * LD + b];
      const float q = edge_divide(mu1 * ex, accd, racc);
      ep[i * LD + b] = acc_nz ? q : 0.0f;
    }
  }
  edge_lds_fence();
  if (p.obj_part == nullptr) {
    if (wr)
      for (int a = 0; a < k; a++) p.mu_edge[e * k2 + a * k + i] = ep[a * LD + i];
    return;
  }
  // ... and the linear objective of the marginals just written (Grid.cpp:1665-1680: binary32
  // products accumulated in binary64), one partial sum per workgroup, so that asking for the
  // objective after a solve does not read the marginals and the potentials again
  __shared__ double s_obj[kEdgeBlock / 64];
  double oacc = 0.0;
  if (wr) {
    const float n1 = pot_h[0] * mu1, n2 = pot_h[1] * mu2;
    oacc = (double)n1 + (double)n2;
    // (the potentials are read a second time.  Keeping the lane's column of them in registers needs
    // the loop over the rows written out with fixed register names, 3 forms x 32 rows: the code no
    // longer fits the instruction cache and the solve loses 6 % at 32 labels; an LDS copy halves
    // the wavefronts per SIMD)
    for (int a0 = 0; a0 < k; a0 += kLoadAhead) {
      float pe_b[kLoadAhead];
#pragma unroll
      for (int u = 0; u < kLoadAhead; u++) pe_b[u] = a0 + u < k ? p.pot_edge[e * k2 + (a0 + u) * k + i] : 0.0f;
#pragma unroll
      for (int u = 0; u < kLoadAhead; u++) {
        const int a = a0 + u;
        if (a < k) {
          const float m = ep[a * LD + i];
          p.mu_edge[e * k2 + a * k + i] = m;
          const float pr = pe_b[u] * m;
          oacc += (double)pr;
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) oacc += __shfl_xor(oacc, d);
  if ((threadIdx.x & 63) == 0) s_obj[threadIdx.x >> 6] = oacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = s_obj[0];
    for (int w = 1; w < kEdgeBlock / 64; w++) t += s_obj[w];
    p.obj_part[blockIdx.x] = t;
  }
}

// Tree.cpp:64-274 for more labels than a wavefront has lanes (64 < k <= KMAX): one edge per
// wavefront, lane l owns the labels l, l + 64, ...; the k x k table of adjusted potentials
// and, after it, of exponentials lives in HBM, in place of the edge marginal it is worked
// out from and becomes (every entry is read before it is overwritten), the node vectors
// in LDS.  Every sum runs in the reference's order and in the reference's precision, with
// the arithmetic of edge_solve_kernel's general form (which equals its specialised forms
// bit for bit), so the two kernels agree wherever both apply.  The reference allocates for
// any k (Tree.cpp:15-22); this is the path that keeps that open, not a fast one: the
// row walks are strided across lanes.
constexpr int kEdgeBigMaxK = 1024;
// edges (= wavefronts) per workgroup: what the node vectors of KMAX labels leave room for in LDS
constexpr int edge_big_epb(int KMAX) { return KMAX <= 256 ? 4 : 1; }
template <int KMAX>
__global__ __launch_bounds__(64 * edge_big_epb(KMAX)) void edge_solve_big_kernel(const EdgeArgs p) {
  constexpr int EPB = edge_big_epb(KMAX);
  __shared__ float s_np[EPB][2 * KMAX];   // node potentials of both endpoints
  __shared__ float s_lmu[EPB][2 * KMAX];  // log mu of both endpoints
  __shared__ float s_buf[EPB][KMAX];
  __shared__ float s_m12[EPB][KMAX], s_m21[EPB][KMAX], s_mu1[EPB][KMAX], s_rs[EPB][KMAX];
  __shared__ LogPiece s_log[kLogPieces];
  __shared__ double s_obj[EPB];
  for (int t = threadIdx.x; t < 2 * kLogPieces; t += 64 * EPB)
    reinterpret_cast<double *>(s_log)[t] = reinterpret_cast<const double *>(d_log_table)[t];
  __syncthreads();

  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long e = (long long)blockIdx.x * EPB + wv;
  const int k = p.k;
  const long long k2 = (long long)k * k;
  const float EPS = 0.00000000000000001;  // Tree.h:15
  const float pen = p.penalty, pp = p.penaltyPrime, rho = p.rho;
  const float inv_pp = 1.0f / pp;
  const double rpp = 1.0 / (double)pp;
  const bool pp_pow2 = (__float_as_uint(pp) & 0x807fffffu) == 0 && (__float_as_uint(pp) >> 23) != 0 &&
                       (__float_as_uint(pp) >> 23) != 255 && (__float_as_uint(inv_pp) & 0x7f800000u) != 0 &&
                       inv_pp * pp == 1.0f;
  float *np = s_np[wv], *lmu = s_lmu[wv], *buf = s_buf[wv];
  double oacc = 0.0;
  if (e < p.E) {
    float *__restrict__ tab = p.mu_edge + e * k2;          // x_mu_edge, then the scratch, then x_mu_edge again
    const float *__restrict__ pe_tab = p.pot_edge + e * k2;
    // node vectors (Tree.cpp:64-73, 84-86, 48-51, 120-122, 141-143, 157-160)
    for (int h = 0; h < 2; h++) {
      const int v = p.edgeList[2 * e + h];
      for (int i = lane; i < k; i += 64) {
        const float mu = p.mu_node[e * 2 * k + h * k + i], m = p.m_vert[(long long)v * k + i];
        float lam = p.lam_node[e * 2 * k + h * k + i];
        const float pot = p.pot_node[e * 2 * k + h * k + i];
        lam = lam + pen * (mu - m);
        p.lam_node[e * 2 * k + h * k + i] = lam;
        const float xnode = rho * (pot + lam) - pen * (m - mu);
        const float muc = ((double)mu <= 0.0) ? EPS : mu;
        const float lv = (float)edge_log((double)muc, s_log);
        const float nd = (float)(-1.0 - (double)lv);
        float val = xnode + pp * nd;
        val = pp_pow2 ? val * inv_pp : val / pp;
        np[h * KMAX + i] = 0.0f - val;
        lmu[h * KMAX + i] = lv;
      }
    }
    edge_lds_fence();
    // edge derivatives and potentials, message 1 -> 2 (Tree.cpp:123-133, 144-150, 98-99, 169-172,
    // 191-205): lane b walks column b
    constexpr int kAhead = 8;
    for (int b = lane; b < k; b += 64) {
      const float lmu2 = lmu[KMAX + b];
      float acc = 0.0f;
      for (int a0 = 0; a0 < k; a0 += kAhead) {
        float mue_b[kAhead], pe_b[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
          const int a = min(a0 + u, k - 1);
          mue_b[u] = tab[(long long)a * k + b];
          pe_b[u] = pe_tab[(long long)a * k + b];
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
          const int a = a0 + u;
          if (a < k) {
            const float mue = mue_b[u];
            const float muec = ((double)mue <= 0.0) ? EPS : mue;
            float ed = (float)(((double)lmu[a] - edge_log((double)muec, s_log)) + 1.0);
            ed = ed + lmu2;
            const float xe = rho * pe_b[u];
            float val = xe + pp * ed;
            val = pp_pow2 ? val * inv_pp : edge_divide(val, (double)pp, rpp);
            const float epv = 0.0f - val;
            tab[(long long)a * k + b] = epv;
            const float c = epv + np[a];
            acc = (float)((double)acc + exp((double)c));
          }
        }
      }
      buf[b] = acc;
    }
    // the row walks below read what other lanes of this wavefront wrote to memory
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    edge_lds_fence();
    float sum = 0.0f;
    for (int j = 0; j < k; j++) sum += buf[j];
    for (int b = lane; b < k; b += 64) s_m12[wv][b] = buf[b] / sum;
    edge_lds_fence();
    // message 2 -> 1 and the exponentials of the edge marginal (Tree.cpp:212-229, 256-268): lane a walks row a
    for (int a = lane; a < k; a += 64) {
      const float np_own = np[a];
      float acc = 0.0f, row_sum = 0.0f;
      for (int b0 = 0; b0 < k; b0 += kAhead) {
        float ep_b[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) ep_b[u] = tab[(long long)a * k + min(b0 + u, k - 1)];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
          const int b = b0 + u;
          if (b < k) {
            const float epv = ep_b[u], n2 = np[KMAX + b];
            const float c = epv + n2;
            acc = (float)((double)acc + exp((double)c));
            float q = epv + np_own;
            q = q + n2;
            const float ex = (float)exp((double)q);
            tab[(long long)a * k + b] = ex;
            row_sum = row_sum + ex;
          }
        }
      }
      buf[a] = acc;
      s_rs[wv][a] = row_sum;
    }
    edge_lds_fence();
    sum = 0.0f;
    for (int j = 0; j < k; j++) sum += buf[j];
    for (int a = lane; a < k; a += 64) s_m21[wv][a] = buf[a] / sum;
    edge_lds_fence();
    // node marginals (Tree.cpp:236-254)
    for (int a = lane; a < k; a += 64) buf[a] = (float)(exp((double)np[a]) * (double)s_m21[wv][a]);
    edge_lds_fence();
    sum = 0.0f;
    for (int j = 0; j < k; j++) sum += buf[j];
    for (int a = lane; a < k; a += 64) {
      const float mu1 = buf[a] / sum;
      s_mu1[wv][a] = mu1;
      p.mu_node[e * 2 * k + a] = mu1;
      const float pr = p.pot_node[e * 2 * k + a] * mu1;
      oacc += (double)pr;
    }
    edge_lds_fence();
    for (int b = lane; b < k; b += 64) buf[b] = (float)(exp((double)np[KMAX + b]) * (double)s_m12[wv][b]);
    edge_lds_fence();
    sum = 0.0f;
    for (int j = 0; j < k; j++) sum += buf[j];
    for (int b = lane; b < k; b += 64) {
      const float mu2 = buf[b] / sum;
      p.mu_node[e * 2 * k + k + b] = mu2;
      const float pr = p.pot_node[e * 2 * k + k + b] * mu2;
      oacc += (double)pr;
    }
    // edge marginal (Tree.cpp:269-273) and its share of the linear objective (Grid.cpp:1665-1680)
    for (int a = lane; a < k; a += 64) {
      const float row_sum = s_rs[wv][a], mu1 = s_mu1[wv][a];
      const double accd = (double)row_sum, racc = 1.0 / accd;
      const bool acc_nz = row_sum != 0.0f;
      for (int b0 = 0; b0 < k; b0 += kAhead) {
        float ex_b[kAhead], pe_b[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
          const int b = min(b0 + u, k - 1);
          ex_b[u] = tab[(long long)a * k + b];
          pe_b[u] = pe_tab[(long long)a * k + b];
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
          const int b = b0 + u;
          if (b < k) {
            const float q = edge_divide(mu1 * ex_b[u], accd, racc);
            const float m = acc_nz ? q : 0.0f;
            tab[(long long)a * k + b] = m;
            const float pr = pe_b[u] * m;
            oacc += (double)pr;
          }
        }
      }
    }
  }
  if (p.obj_part == nullptr) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) oacc += __shfl_xor(oacc, d);
  if (lane == 0) s_obj[wv] = oacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = s_obj[0];
    for (int w = 1; w < EPB; w++) t += s_obj[w];
    p.obj_part[blockIdx.x] = t;
  }
}

// Sum of the solve's per-workgroup objective terms: one workgroup, one value.
__global__ __launch_bounds__(1024) void edge_objective_sum_kernel(double *__restrict__ dst,
                                                                   const double *__restrict__ part, int n) {
  __shared__ double s_red[16];
  double acc = 0.0;
  // eight loads in flight per thread (a thread has up to 64 terms at 512 x 512, k = 32)
  for (int q0 = threadIdx.x; q0 < n; q0 += 8 * 1024) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = q0 + u * 1024 < n ? part[q0 + u * 1024] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) acc += t[u];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = s_red[0];
    for (int w = 1; w < 16; w++) t += s_red[w];
    dst[0] = t;
  }
}

// The solve's logarithm on a list of values (accuracy tests).
__global__ __launch_bounds__(kEdgeBlock) void edge_log_values_kernel(double *__restrict__ y, const double *__restrict__ x,
                                                                     long long n) {
  __shared__ LogPiece s_log[kLogPieces];
  reinterpret_cast<double *>(s_log)[threadIdx.x] = reinterpret_cast<const double *>(d_log_table)[threadIdx.x];
  __syncthreads();
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x)
    y[idx] = edge_log(x[idx], s_log);
}

// Entry l of the copies inc[b .. en) of a vertex, added in that (edge-list) order.  Four copies at
// a time (a grid vertex has at most four): their positions, then their values, leave together.
__device__ __forceinline__ float edge_sum_copies(const float *__restrict__ mu_node, const int *__restrict__ inc,
                                                 int b, int en, int k, int l) {
  float s = 0.0f;
  for (int q0 = b; q0 < en; q0 += 4) {
    int pos[4];
    float val[4];
#pragma unroll
    for (int u = 0; u < 4; u++) pos[u] = inc[min(q0 + u, en - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) val[u] = mu_node[(long long)pos[u] * k + l];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (q0 + u < en) s += val[u];
  }
  return s;
}

// Consensus (Grid.cpp:144-148 + Tree.cpp:244,253 + Grid.cpp:163-181): the
// copies of vertex v are summed in edge-list order, as the serial loop of
// Grid::Compute adds them, then divided by the copy count.  inc holds, per
// vertex, the positions (2*edge + endpoint) of its copies in ascending order.
__global__ void edge_consensus_kernel(float *__restrict__ m_vert, const float *__restrict__ mu_node,
                                      const int *__restrict__ inc_off, const int *__restrict__ inc,
                                      int V, int k) {
  const long long total = (long long)V * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx / k), l = (int)(idx % k);
    const int b = inc_off[v], en = inc_off[v + 1];
    const float s = edge_sum_copies(mu_node, inc, b, en, k, l);
    const int cnt = en - b;
    // a vertex no edge touches (possible in a caller's graph; the reference's vertex
    // sets hold endpoints only, Grid.cpp:815-856) keeps the uniform vector admm_init wrote
    if (cnt > 0) m_vert[idx] = s / cnt;
  }
}

// The same on several ranks (Grid.cpp:1639-1663, 163-181).  A rank first sums
// its own copies of every vertex its edges touch, in edge-list order ...
__global__ void edge_partial_kernel(float *__restrict__ m_part, const float *__restrict__ mu_node,
                                    const int *__restrict__ inc_off, const int *__restrict__ inc,
                                    int V, int k) {
  const long long total = (long long)V * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx / k), l = (int)(idx % k);
    m_part[idx] = edge_sum_copies(mu_node, inc, inc_off[v], inc_off[v + 1], k, l);
  }
}

// ... packs the sums of the vertices it shares with each neighbouring rank
// (send_idx: local vertex per shared entry, neighbour-major, Grid.cpp:1648-1651) ...
__global__ void edge_pack_kernel(float *__restrict__ sendv, const float *__restrict__ m_part,
                                 const int *__restrict__ send_idx, int S, int k) {
  const long long total = (long long)S * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x)
    sendv[idx] = m_part[(long long)send_idx[idx / k] * k + idx % k];
}

// ... and, once the neighbours' sums have arrived (recvv, neighbour-major in
// ascending rank order), adds them in that order and divides by the global
// copy count.  rcv_pos lists, per vertex, its entries of recvv in ascending order.
__global__ void edge_finalize_kernel(float *__restrict__ m_vert, const float *__restrict__ m_part,
                                     const float *__restrict__ recvv, const int *__restrict__ rcv_off,
                                     const int *__restrict__ rcv_pos, const int *__restrict__ cnt,
                                     int V, int k) {
  const long long total = (long long)V * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx / k), l = (int)(idx % k);
    float s = m_part[idx];
    for (int q = rcv_off[v]; q < rcv_off[v + 1]; q++) s += recvv[(long long)rcv_pos[q] * k + l];
    m_vert[idx] = s / cnt[v];
  }
}

// Initial split of the unaries (Grid.cpp:1478-1487, 952-956): theta_v added
// once per copy, divided by the copy count, then by rho * count.
__global__ void edge_init_node_kernel(float *__restrict__ pot_node, float *__restrict__ mu_node,
                                      float *__restrict__ lam_node,
                                      const float *__restrict__ vert_pot,
                                      const int *__restrict__ edgeList,
                                      const int *__restrict__ inc_off, int E, int k, float rho,
                                      float inv_k) {
  const long long total = (long long)E * 2 * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long ep = idx / k;  // 2*edge + endpoint
    const int l = (int)(idx % k);
    const int v = edgeList[ep];
    const int cnt = inc_off[v + 1] - inc_off[v];
    const float th = vert_pot[(long long)v * k + l];
    float s = 0.0f;
    for (int q = 0; q < cnt; q++) s += th;
    s = s / cnt;
    pot_node[idx] = s / (rho * cnt);
    mu_node[idx] = inv_k;
    lam_node[idx] = 0.0f;
  }
}

__global__ void edge_init_edge_kernel(float *__restrict__ pot_edge, float *__restrict__ mu_edge,
                                      const float *__restrict__ edge_pot, int per_edge,
                                      long long total, int k2, float rho, float inv_k2) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const float w = per_edge ? edge_pot[idx] : edge_pot[idx % k2];
    pot_edge[idx] = w / rho;  // Grid.cpp:975-976
    mu_edge[idx] = inv_k2;    // Grid.cpp:977
  }
}

__global__ void fill_kernel(float *__restrict__ p, long long n, float v) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
       idx += (long long)gridDim.x * blockDim.x)
    p[idx] = v;
}

__global__ void fill_int_kernel(int *__restrict__ p, long long n, int v) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n;
       idx += (long long)gridDim.x * blockDim.x)
    p[idx] = v;
}

// Gather of the consensus into per-edge layout (what Grid::Update copies into
// edge_mu_node, Grid.cpp:188-192); used when that state is downloaded.
__global__ void edge_gather_m_kernel(float *__restrict__ m_node, const float *__restrict__ m_vert,
                                     const int *__restrict__ edgeList, int E, int k) {
  const long long total = (long long)E * 2 * k;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long ep = idx / k;
    m_node[idx] = m_vert[(long long)edgeList[ep] * k + idx % k];
  }
}

// Linear objective (Grid.cpp:1665-1680): binary32 products accumulated in
// binary64.  Each workgroup reduces a contiguous slice of the concatenated
// [node | edge] product streams and writes one partial.
__global__ __launch_bounds__(256) void edge_objective_kernel(
    double *__restrict__ partial, const float *__restrict__ pot_node,
    const float *__restrict__ mu_node, long long n_node, const float *__restrict__ pot_edge,
    const float *__restrict__ mu_edge, long long n_edge) {
  __shared__ double s_red[4];
  double acc = 0.0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_node; i += stride) {
    const float pr = pot_node[i] * mu_node[i];
    acc += (double)pr;
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_edge; i += stride) {
    const float pr = pot_edge[i] * mu_edge[i];
    acc += (double)pr;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

}  // namespace admm
