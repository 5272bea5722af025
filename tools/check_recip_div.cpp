// The edge solve divides a row of binary32 values by one binary32 divisor (Tree.cpp:271) as
//   q = (float)((double)num * (1.0 / (double)den)),  and again as num / den when q is subnormal.
// This program checks on the host (IEEE binary64 multiply and divide, as on the device) that the
// result is num / den bit for bit: random pairs of every magnitude, pairs built to land near
// rounding boundaries, quotients around the overflow and the underflow thresholds.
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

static float fast_div(float num, float den, long *slow) {
  const double r = 1.0 / (double)den;
  float q = (float)((double)num * r);
  if (std::fpclassify(q) == FP_SUBNORMAL) { q = num / den; ++*slow; }
  return q;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 200000000;
  long bad = 0, slow = 0, missed_ties = 0;
  for (long it = 0; it < n; it++) {
    const uint64_t u = rnd();
    float a, b;
    switch (it & 3) {
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
      default:  // quotients around the largest finite value
        a = from_bits(0x7e000000u + (uint32_t)(u & 0x17fffffu));
        b = from_bits(0x3d000000u + (uint32_t)((u >> 26) & 0x3ffffffu));
        break;
    }
    if (b == 0.0f) continue;
    if (u >> 63) a = -a;  // x / penaltyPrime has numerators of either sign
    const float want = a / b;
    long sl = 0;
    const float got = fast_div(a, b, &sl);
    slow += sl;
    if (bits(got) != bits(want)) bad++;
    // how often the product alone would have been wrong (all of them subnormal ties, which is
    // what the second division is for)
    if (sl && bits((float)((double)a * (1.0 / (double)b))) != bits(want)) missed_ties++;
  }
  printf("%ld pairs: %ld results differ from the division; %ld took the second division, of which the product alone missed %ld\n",
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
