// Accuracy of edge_log (admm_amd/csrc/edge_log.h) on the host: the same source as the device
// function, binary64 fused multiply-adds on both.  Compares with the long double logarithm
// (64-bit significand) and prints the largest error in units of the last place per class of
// arguments, and how often the result rounded to binary32 differs from the C library's log
// rounded to binary32 (what the edge solver stores, Tree.cpp:120-133).
//   g++ -O2 -std=c++20 -ffp-contract=off -o check_edge_log tools/check_edge_log.cpp && ./check_edge_log [samples]
//   ./check_edge_log --values in.bin out.bin   (the fixture of tests/golden/make_edge_log.py)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../admm_amd/csrc/edge_log.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static double ulp_err(double x, double y) {
  const long double t = logl((long double)x);
  int e;
  frexpl(t, &e);
  const long double ulp = ldexpl(1.0L, (e < -1021 ? -1021 : e) - 53);
  return (double)(fabsl((long double)y - t) / ulp);
}

// --values in.bin out.bin: edge_log of every binary64 value of in.bin, written to out.bin
static int values(const char *in, const char *out) {
  FILE *fi = fopen(in, "rb"), *fo = fopen(out, "wb");
  if (!fi || !fo) return 2;
  double x;
  while (fread(&x, sizeof x, 1, fi) == 1) {
    const double y = admm::edge_log(x, admm::kLogTableHost);
    fwrite(&y, sizeof y, 1, fo);
  }
  fclose(fi);
  return fclose(fo) ? 2 : 0;
}

int main(int argc, char **argv) {
  if (argc == 4 && !strcmp(argv[1], "--values")) return values(argv[2], argv[3]);
  const long n = argc > 1 ? atol(argv[1]) : 20000000;
  struct Cls { const char *name; double worst; double at; long flips, cnt; } cls[6] = {
      {"binary32 in (0, 1]", 0, 0, 0, 0}, {"binary32 near 1", 0, 0, 0, 0}, {"binary32 any positive", 0, 0, 0, 0},
      {"binary64 in (0, 1]", 0, 0, 0, 0}, {"binary64 near 1", 0, 0, 0, 0}, {"binary64 any positive normal", 0, 0, 0, 0}};
  for (long it = 0; it < n; it++) {
    for (int c = 0; c < 6; c++) {
      double x;
      const uint64_t u = rnd();
      if (c == 0) { float f = (float)((u >> 11) * (1.0 / 9007199254740992.0)); if (f <= 0) f = 1e-17f; x = f; }
      else if (c == 1) { float f = 1.0f + (float)((int)(u & 0xffff) - 32768) * 0x1p-20f * (float)((u >> 16) & 15); x = f; }
      else if (c == 2) { uint32_t b = (uint32_t)(u >> 33); if (b == 0) b = 1; if (b >= 0x7f800000u) b = 0x7f7fffffu; float f; memcpy(&f, &b, 4); x = f; }
      else if (c == 3) { x = (double)(u >> 11) * (1.0 / 9007199254740992.0); if (x <= 0) x = 1e-300; }
      else if (c == 4) { x = 1.0 + ((double)(int64_t)(u >> 20) - 8796093022208.0) * 0x1p-50; }
      else { uint64_t b = u >> 1; if (b < 0x0010000000000000ull) b += 0x0010000000000000ull; if (b >= 0x7ff0000000000000ull) b = 0x7fefffffffffffffull; memcpy(&x, &b, 8); }
      const double y = admm::edge_log(x, admm::kLogTableHost);
      const double e = ulp_err(x, y);
      if (e > cls[c].worst) { cls[c].worst = e; cls[c].at = x; }
      cls[c].flips += (float)y != (float)log(x);
      cls[c].cnt++;
    }
  }
  int bad = 0;
  for (auto &c : cls) {
    printf("%-30s worst %.4f ulp at %a; binary32 result differs from the C library's in %ld of %ld\n", c.name, c.worst, c.at, c.flips, c.cnt);
    if (c.worst > 0.76) bad = 1;
  }
  const double sp[] = {1.0, 1e-17, 0.5, 2.0, 0x1p-149, 0x1.fffffep127, INFINITY, NAN, 0.6875, 1.375, 0x1.fffffffffffffp-1, 0x1.0000000000001p0};
  for (double x : sp) printf("log(%a) = %a (C library %a)\n", x, admm::edge_log(x, admm::kLogTableHost), log(x));
  if (admm::edge_log(1.0, admm::kLogTableHost) != 0.0) bad = 1;
  return bad;
}
