// Natural logarithm in binary64 for the edge solver (Tree.cpp:120-133 take the logarithm of every
// node and edge marginal once per solve: k * k + 2 k of them per edge).  The device library's log
// spends about 110 instructions per value; this one about 25, with an error below 0.75 units in
// the last place (0.53 outside [1 - 2^-7, 1 + 2^-6]; tools/check_edge_log.cpp measures it against
// the long double logarithm; the same source compiles for the host there, and binary64 fused
// multiply-adds give the same bits on both).  The solver rounds every logarithm to binary32 at
// once, so against the C library's log the stored values differ about as rarely as the device
// library's do (none in 1.2e8 random arguments).
//
//   x = 2^k z with z in [0.6875, 1.375); the leading 7 bits of z's pattern past 0.6875 pick a
//   piece i with centre c_i;  r = z / c_i - 1 (|r| < 2^-7);  log x = k ln2 + log c_i + log1p(r).
//
// k ln2_hi + logc_hi is exact (both are multiples of 2^-42), the sum with r is kept as a pair,
// log1p(r) - r is a polynomial of degree 8 (truncation below 2e-18 relative).  The two pieces
// touching 1 have c = 1, so r = z - 1 is exact and small results keep full relative accuracy.
// Any positive finite x, subnormal included when it comes from a binary32 value (those are
// normal in binary64); NaN and +infinity come back as they are; zero and negative arguments do
// not reach it (Tree.cpp:46-57 clamps them to 1e-17 first).
#pragma once

#include "log_table.inc"

#ifdef __HIPCC__
#define ADMM_LOG_FN __device__ __forceinline__
#else
#define ADMM_LOG_FN static inline
#endif

namespace admm {

constexpr int kLogPieces = 128;
// 16 bytes per piece, one read: -log(1 / c) as a binary64 high part and a binary32 low part, and
// 1 / c in binary32 (the logarithm is that of the rounded value, so nothing is lost by it)
struct __attribute__((aligned(16))) LogPiece {
  double logc_hi;
  float invc, logc_lo;
};
static const LogPiece kLogTableHost[kLogPieces] = {ADMM_LOG_TABLE};

ADMM_LOG_FN double edge_log(double x, const LogPiece *tab) {
  const unsigned long long ix = __builtin_bit_cast(unsigned long long, x);
  const unsigned hx = (unsigned)(ix >> 32), lx = (unsigned)ix;
  const unsigned tmp = hx - 0x3fe60000u;
  const int i = (int)((tmp >> 13) & 127u);
  const int k = (int)tmp >> 20;
  const unsigned hz = hx - (tmp & 0xfff00000u);
  const double z = __builtin_bit_cast(double, ((unsigned long long)hz << 32) | lx);
  const LogPiece piece = tab[i];
  const double invc = (double)piece.invc, lchi = piece.logc_hi, lclo = (double)piece.logc_lo;
  const double r = __builtin_fma(z, invc, -1.0);
  const double kd = (double)k;
  const double w = __builtin_fma(kd, ADMM_LOG_LN2_HI, lchi);  // exact
  const double hi = w + r;
  double lo = (w - hi) + r;  // what the sum above rounded away (|w| >= |r| or w == 0)
  lo = lo + __builtin_fma(kd, ADMM_LOG_LN2_LO, lclo);
  const double r2 = r * r;
  double q = __builtin_fma(r, -1.0 / 8.0, 1.0 / 7.0);
  q = __builtin_fma(r, q, -1.0 / 6.0);
  q = __builtin_fma(r, q, 1.0 / 5.0);
  q = __builtin_fma(r, q, -1.0 / 4.0);
  q = __builtin_fma(r, q, 1.0 / 3.0);
  q = __builtin_fma(r, q, -1.0 / 2.0);
  const double y = hi + __builtin_fma(r2, q, lo);
  return hx >= 0x7ff00000u ? x : y;  // NaN, +infinity
}

}  // namespace admm
