// The edge solve divides a row of binary32 values by one binary32 divisor (Tree.cpp:271) as
//   r = 1.0 / (double)den;  q0 = (double)num * r;  q1 = fma(fma(-q0, den, num), r, q0);
//   q = (float)(q1 is NaN ? q0 : q1)
// (q1 is the correctly rounded binary64 quotient; with an infinite operand the residual is NaN
// and the plain product already is the answer).  A numerator of -0 gives +0 where the division
// gives -0; the solve's numerators are products of non-negative values, and the other use
// subtracts the quotient from 0, which drops the sign of a zero.  This program checks
// on the host (IEEE binary64 arithmetic, as on the device) that the result is num / den bit for
// bit: random pairs of every magnitude, pairs built to land next to rounding boundaries,
// quotients around the overflow and the underflow thresholds, and exact ties between two
// subnormal results (which the product q0 alone can miss).
//   g++ -O2 -std=c++17 -ffp-contract=off -o check_recip_div tools/check_recip_div.cpp && ./check_recip_div [pairs]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint64_t s = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

static float fast_div(float num, float den, long *subnormal) {
  const double r = 1.0 / (double)den;
  const double q0 = (double)num * r;
  const double q1 = __builtin_fma(__builtin_fma(-q0, (double)den, (double)num), r, q0);
  const float q = (float)(q1 != q1 ? q0 : q1);
  if (std::fpclassify(q) == FP_SUBNORMAL) ++*subnormal;
  return q;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 200000000;
  long bad = 0, slow = 0, missed_ties = 0, bad_case[5] = {0, 0, 0, 0, 0};
  for (long it = 0; it < n; it++) {
    const uint64_t u = rnd();
    float a, b;
    switch (it % 5) {
      case 0:  // any two positive finite values
        a = from_bits((uint32_t)(u >> 33) % 0x7f800000u);
        b = from_bits((uint32_t)(u & 0x7fffffffu) % 0x7f800000u);
        break;
      case 1: {  // a = b * m rounded, m a 25-bit midpoint: quotients next to a rounding boundary
        b = from_bits(0x3f800000u | (uint32_t)(u & 0x7fffffu));
        const double m = 1.0 + (double)(((u >> 23) & 0xffffffu) * 2 + 1) * 0x1p-25;
        a = (float)((double)b * m);
        a = from_bits(bits(a) + (uint32_t)((u >> 60) & 3) - 1);
        break;
      }
      case 2:  // quotients around the smallest normal value and below it
        a = from_bits(0x00800000u + (uint32_t)(u & 0x3ffffffu));
        b = from_bits(0x3f800000u + (uint32_t)((u >> 26) & 0xfffffffu) % 0x0c000000u);
        break;
      case 4: {  // exact ties between two subnormal results: a = b * (odd * 2^-150)
        const uint32_t A = 0x800u | (uint32_t)(u & 0x7ffu);              // 12-bit significand of b
        const uint32_t m = (uint32_t)((u >> 11) & 0x7ffu) | 1u;          // odd, below 2^11
        const int eb = (int)((u >> 22) % 60) - 30;
        b = ldexpf((float)A, eb);
        a = (float)ldexp((double)A * (double)m, eb - 150);               // exact: 23 bits at most
        break;
      }
      default:  // quotients around the largest finite value
        a = from_bits(0x7e000000u + (uint32_t)(u & 0x17fffffu));
        b = from_bits(0x3d000000u + (uint32_t)((u >> 26) & 0x3ffffffu));
        break;
    }
    if (b == 0.0f) continue;
    if ((u >> 63) && a != 0.0f) a = -a;  // x / penaltyPrime has numerators of either sign
    const float want = a / b;
    long sl = 0;
    const float got = fast_div(a, b, &sl);
    slow += sl;
    if (bits(got) != bits(want)) { bad++; if (bad_case[it % 5]++ < 2) printf("case %ld: %a / %a = %a, division %a\n", it % 5, a, b, got, want); }
    // how often the product alone would have been wrong (subnormal ties: what the correction is for)
    if (bits((float)((double)a * (1.0 / (double)b))) != bits(want)) missed_ties++;
  }
  printf("%ld pairs: %ld results differ from the division; %ld subnormal results; the uncorrected product would have missed %ld\n",
         n, bad, slow, missed_ties);
  const float sp[][2] = {{0.0f, 3.0f}, {1.0f, INFINITY}, {INFINITY, INFINITY}, {INFINITY, 2.0f}, {NAN, 1.0f}, {1.0f, NAN},
                         {3.0e38f, 0.5f}, {1e-45f, 2.0f}, {3e-45f, 2.0f}, {1.1754944e-38f, 1.0000001f}};
  for (auto &p : sp) {
    long sl = 0;
    const float got = fast_div(p[0], p[1], &sl), want = p[0] / p[1];
    const bool same = bits(got) == bits(want) || (got != got && want != want);
    printf("%g / %g = %g (division %g)%s\n", p[0], p[1], got, want, same ? "" : "  DIFFERENT");
    bad += !same;
  }
  return bad != 0;
}
