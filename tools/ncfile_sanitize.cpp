// Host-only robustness check of the graph-file reader: build with
//   g++ -std=c++17 -g -fsanitize=address,undefined -Iinclude tools/ncfile_sanitize.cpp \
//       admm_amd/csrc/host/NcFile.cpp admm_amd/csrc/host/graphfile_capi.cpp -o /tmp/ncfile_sanitize
// and run it: every truncation of a valid file and a few thousand corrupted
// copies go through the reader, which must fail cleanly or succeed, never crash.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "admm_graphfile.h"

static std::vector<char> slurp(const std::string &p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string &p, const std::vector<char> &b, size_t n) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write(b.data(), (std::streamsize)n);
}
static void exercise(const std::string &p) {
  admm_graphfile_info gi;
  if (admm_graphfile_read_info(p.c_str(), &gi) != 0) return;
  if (gi.num_edges < 0 || gi.num_edges > 1000 || gi.num_vertices < 0 || gi.num_vertices > 1000 || gi.k < 0 || gi.k > 64 ||
      gi.k2 < 0 || gi.k2 > 4096)
    return;  // a corrupted size: the caller would not allocate for it either
  std::vector<int32_t> el(2 * (size_t)gi.num_edges + 2);
  std::vector<float> un((size_t)gi.num_vertices * gi.k + 1), pw((size_t)gi.num_edges * gi.k2 + 1);
  admm_graphfile_read_edges(p.c_str(), 0, gi.num_edges, el.data());
  admm_graphfile_read_unaries(p.c_str(), 0, gi.num_vertices, un.data(), 1);
  admm_graphfile_read_pairwise(p.c_str(), 0, gi.num_edges, pw.data(), 1);
  int shared = 0;
  std::vector<float> tab((size_t)gi.k2 + 1);
  admm_graphfile_shared_table(p.c_str(), tab.data(), 1, &shared);
}

int main() {
  const std::string good = "/tmp/ncfile_sanitize_good.nc", bad = "/tmp/ncfile_sanitize_bad.nc";
  for (int version : {1, 2, 5}) {
    admm_graphfile_info gi{version, 3, 4, 12, 17, 3, 9};
    std::vector<int32_t> el(34);
    for (size_t i = 0; i < el.size(); i++) el[i] = (int)(i % 12);
    std::vector<float> un(36, 0.5f), pw(17 * 9, 0.25f);
    if (admm_graphfile_write(good.c_str(), &gi, el.data(), un.data(), pw.data(), 1) != 0) { fprintf(stderr, "write failed: %s\n", admm_graphfile_last_error()); return 1; }
    exercise(good);
    const std::vector<char> b = slurp(good);
    const size_t hdr = 600;  // the header and the start of the data
    for (size_t n = 0; n < b.size(); n += (n < hdr ? 1 : 37)) { spit(bad, b, n); exercise(bad); }
    srand(1234 + version);
    for (int it = 0; it < 3000; it++) {
      std::vector<char> c = b;
      const int flips = 1 + rand() % 3;
      for (int f = 0; f < flips; f++) c[(size_t)rand() % std::min(c.size(), hdr)] = (char)rand();
      spit(bad, c, c.size());
      exercise(bad);
    }
  }
  printf("graph-file reader: truncations and corruptions handled without a fault\n");
  return 0;
}
