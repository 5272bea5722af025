/*
 * admm_graphfile.h -- C ABI of libadmm_graphfile.so: the graph files of
 * xiaocai00/ADMM without PnetCDF or MPI.
 *
 * The reference stores its input graph in a NetCDF classic-family file through
 * PnetCDF (schema: Grid::Save, Grid.cpp:1066-1097) and reads hyperslabs of it
 * per rank (Grid.cpp:476-543, 654-702).  This library reads and writes that
 * schema directly (CDF-1, CDF-2 and CDF-5 headers, big-endian data), so a file
 * written by the reference can be solved here and the other way round.  It is
 * host-only: no HIP, no device.
 *
 * Schema (Grid.cpp:1069-1096)
 *   dimensions  numVertices, numVerticePotentials, numEdges, numEdgePotentials,
 *               numEndPointsPerEdge (= 2)
 *   variables   edgeList          int   [numEdges][numEndPointsPerEdge]
 *               edgePotentials    float [numEdges][numEdgePotentials]
 *               verticePotentials float [numVertices][numVerticePotentials]
 *   attributes  graphType (text "Grid"), numRows, numCols, numEdges,
 *               numVertices, numVerticePotentials, numEdgePotentials (int)
 * The file holds potentials as scores; the reference's loader negates both
 * arrays after reading (Grid.cpp:547-560), which turns them into the costs the
 * solvers work on.  The read calls below do the same when `negate` is set.
 *
 * Every function returns 0 on success and a negative ADMM_GF_E_* code on
 * failure; admm_graphfile_last_error() gives this thread's last message.
 */
#ifndef ADMM_GRAPHFILE_H_
#define ADMM_GRAPHFILE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMM_GF_OK 0
#define ADMM_GF_E_ARG (-1)    /* bad argument                                  */
#define ADMM_GF_E_IO (-2)     /* open / read / write failed                    */
#define ADMM_GF_E_FORMAT (-3) /* not a CDF file, or not this schema            */

typedef struct admm_graphfile_info {
  int32_t version;       /* 1, 2 or 5 (read); 0 lets the writer choose: 1 when
                          * every offset fits 32 bits, as ncmpi_create without a
                          * format flag does (Grid.cpp:1067), else 5            */
  int32_t rows, cols;    /* numRows, numCols; -1 when the file has none         */
  int64_t num_vertices;  /* numVertices                                        */
  int64_t num_edges;     /* numEdges                                           */
  int32_t k;             /* numVerticePotentials                               */
  int32_t k2;            /* numEdgePotentials                                  */
} admm_graphfile_info;

/* Grid::SetGlobalInfo(fname) (Grid.cpp:654-702): the global sizes.  The
 * attributes are read as the reference reads them; a size whose attribute is
 * missing is taken from the dimension of the same name. */
int admm_graphfile_read_info(const char *path, admm_graphfile_info *info);

/* Hyperslab reads of Grid::PartitionByEdge (Grid.cpp:476-494, 510-523,
 * 529-543): `count` rows starting at row `first` of edgeList (2 ints per row),
 * verticePotentials (k floats per row) and edgePotentials (k2 floats per row),
 * converted to host byte order.  negate != 0 applies the loader's sign flip
 * (Grid.cpp:547-560). */
int admm_graphfile_read_edges(const char *path, int64_t first, int64_t count, int32_t *edge_list);
int admm_graphfile_read_unaries(const char *path, int64_t first, int64_t count, float *out, int negate);
int admm_graphfile_read_pairwise(const char *path, int64_t first, int64_t count, float *out, int negate);

/* A rectangle of a grid's verticePotentials: for each of the `nrows` grid rows
 * from `row0` on, the `ncols` vertices from column `col0` on, out =
 * [nrows][ncols][k], where vertex (r, c) is row r * grid_cols + c of the
 * variable.  What one rank of a row / column strip partition keeps
 * (Grid::PartitionByRow reads its rows the same way, Grid.cpp:886-903). */
int admm_graphfile_read_unaries_block(const char *path, int32_t grid_cols, int32_t row0, int32_t nrows,
                                      int32_t col0, int32_t ncols, float *out, int negate);

/* Whether every edge of the file carries the same k2-entry table (what the
 * chain solver needs).  table (k2 floats, may be NULL) receives the table of
 * edge 0, sign-flipped when negate != 0; *shared is 1 or 0.  Streams the
 * variable in blocks, so the file may be larger than memory. */
int admm_graphfile_shared_table(const char *path, float *table, int negate, int *shared);

/* Grid::Save (Grid.cpp:1049-1131): writes the whole graph in the reference's
 * schema, values as given (no sign flip: pass scores).  pairwise_per_edge = 0:
 * `pairwise` is one k2-entry table, written for every edge. */
int admm_graphfile_write(const char *path, const admm_graphfile_info *info, const int32_t *edge_list,
                         const float *unary, const float *pairwise, int pairwise_per_edge);

/* Grid::Convert + Converter.cpp (Grid.cpp:985-1045, Converter.cpp:16-22): the
 * manifest names four files, one per line -- info (text: vertices, edges, k),
 * edge list (raw int32 pairs), node potentials (raw float32 [V][k]) and edge
 * potentials (raw float32 [E][k*k]), all in host byte order -- which are
 * written out as a graph file.  The reference leaves numRows / numCols unset on
 * this path; they are written as -1. */
int admm_graphfile_convert(const char *manifest_path, const char *out_path);

const char *admm_graphfile_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_GRAPHFILE_H_ */
