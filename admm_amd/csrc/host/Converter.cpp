// The reference's converter (Converter.cpp:8-23): raw binary files named by a
// four-line manifest -> a graph file in the reference's schema.
//
//   converter <infile> <outfile>
#include <cstdio>

#include "../../../include/admm_graphfile.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    fprintf(stderr, "./converter <infile> <outfile>\n");  // Converter.cpp:13
    return 2;
  }
  const int rc = admm_graphfile_convert(argv[1], argv[2]);
  if (rc != ADMM_GF_OK) {
    fprintf(stderr, "converter: %s\n", admm_graphfile_last_error());
    return 1;
  }
  admm_graphfile_info info;
  if (admm_graphfile_read_info(argv[2], &info) == ADMM_GF_OK)
    printf("wrote %s: CDF-%d, %lld vertices, %lld edges, %d labels\n", argv[2], info.version,
           (long long)info.num_vertices, (long long)info.num_edges, info.k);
  return 0;
}
