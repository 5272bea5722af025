// C ABI of libadmm_graphfile.so (see include/admm_graphfile.h): the
// reference's graph-file schema on top of the CDF reader / writer in NcFile.
#include "../../../include/admm_graphfile.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "NcFile.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

// Sizes as Grid::SetGlobalInfo reads them (Grid.cpp:663-669), each falling
// back to the dimension of the same name.
int load_info(const std::string &path, ncfile::Header *h, admm_graphfile_info *info) {
  std::string err;
  if (!ncfile::read_header(path, h, &err)) {
    const bool io = err.compare(0, 6, "cannot") == 0;
    return fail(io ? ADMM_GF_E_IO : ADMM_GF_E_FORMAT, err);
  }
  auto att = [&](const char *name, int *out) -> bool {
    const ncfile::Att *a = h->find_att(name);
    return a && a->as_int(out);
  };
  auto sized = [&](const char *name, int64_t *out) -> bool {
    int v = 0;
    if (att(name, &v)) { *out = v; return true; }
    const int d = h->find_dim(name);
    if (d < 0) return false;
    *out = h->dims[(size_t)d].len;
    return true;
  };
  info->version = h->version;
  int v = 0;
  info->rows = att("numRows", &v) ? v : -1;
  info->cols = att("numCols", &v) ? v : -1;
  int64_t k = 0, k2 = 0;
  if (!sized("numVertices", &info->num_vertices) || !sized("numEdges", &info->num_edges) ||
      !sized("numVerticePotentials", &k) || !sized("numEdgePotentials", &k2))
    return fail(ADMM_GF_E_FORMAT, "'" + path + "' lacks the sizes numVertices / numEdges / numVerticePotentials / numEdgePotentials");
  info->k = (int32_t)k;
  info->k2 = (int32_t)k2;
  return ADMM_GF_OK;
}

// A caller sizes its buffers from the sizes read_info reports: a variable
// whose own dimensions say otherwise must not be read into them.
bool shape_ok(const ncfile::Header &h, const admm_graphfile_info &info, const ncfile::Var &v) {
  const bool node = v.name == "verticePotentials", list = v.name == "edgeList";
  const int64_t want_rows = node ? info.num_vertices : info.num_edges;
  const int64_t want_width = list ? 2 : (node ? info.k : info.k2);
  if (v.dimids.size() != 2) return false;
  for (int64_t id : v.dimids) if (id < 0 || id >= (int64_t)h.dims.size()) return false;
  return h.dims[(size_t)v.dimids[0]].len == want_rows && h.dims[(size_t)v.dimids[1]].len == want_width;
}

int read_var(const char *path, const char *var, int64_t first, int64_t count, void *out, bool negate) {
  if (!path || (!out && count > 0)) return fail(ADMM_GF_E_ARG, "null argument");
  ncfile::Header h;
  std::string err;
  admm_graphfile_info info;
  const int rc = load_info(path, &h, &info);
  if (rc) return rc;
  const ncfile::Var *v = h.find_var(var);
  if (!v) return fail(ADMM_GF_E_FORMAT, std::string("'") + path + "' has no variable '" + var + "'");
  const bool is_list = strcmp(var, "edgeList") == 0;
  const bool is_float = v->type == ncfile::NC_FLOAT;
  if (is_list == is_float || (is_list && v->type != ncfile::NC_INT))
    return fail(ADMM_GF_E_FORMAT, std::string("variable '") + var + "' has the wrong type");
  if (!shape_ok(h, info, *v)) return fail(ADMM_GF_E_FORMAT, std::string("variable '") + var + "' of '" + path + "' does not have the shape the file's sizes state");
  if (!ncfile::read_rows(path, h, var, first, count, out, &err)) return fail(ADMM_GF_E_IO, err);
  if (negate && is_float) {
    const int64_t n = count * h.dims[(size_t)v->dimids[1]].len;
    float *f = static_cast<float *>(out);
    for (int64_t i = 0; i < n; i++) f[i] = -f[i];  // Grid.cpp:547-560
  }
  return ADMM_GF_OK;
}

int write_graph(const char *path, const admm_graphfile_info &info, const int32_t *edges, const float *unary,
                const float *pairwise, bool per_edge) {
  ncfile::Writer w(info.version);
  // order of definition as in Grid::Save (Grid.cpp:1070-1096)
  const int dV = w.def_dim("numVertices", info.num_vertices);
  const int dK = w.def_dim("numVerticePotentials", info.k);
  const int dE = w.def_dim("numEdges", info.num_edges);
  const int dK2 = w.def_dim("numEdgePotentials", info.k2);
  const int d2 = w.def_dim("numEndPointsPerEdge", 2);
  const int vList = w.def_var("edgeList", ncfile::NC_INT, {dE, d2});
  const int vEdge = w.def_var("edgePotentials", ncfile::NC_FLOAT, {dE, dK2});
  const int vNode = w.def_var("verticePotentials", ncfile::NC_FLOAT, {dV, dK});
  w.put_att_text("graphType", "Grid", true);  // 5 characters with the NUL, Grid.cpp:1087
  w.put_att_int("numRows", info.rows);
  w.put_att_int("numCols", info.cols);
  w.put_att_int("numEdges", (int)info.num_edges);
  w.put_att_int("numVertices", (int)info.num_vertices);
  w.put_att_int("numVerticePotentials", info.k);
  w.put_att_int("numEdgePotentials", info.k2);
  std::string err;
  if (!w.begin_data(path, &err)) return fail(ADMM_GF_E_IO, err);
  if (!w.write_rows(vList, 0, info.num_edges, edges, &err)) return fail(ADMM_GF_E_IO, err);
  if (!w.write_rows(vNode, 0, info.num_vertices, unary, &err)) return fail(ADMM_GF_E_IO, err);
  if (per_edge) {
    if (!w.write_rows(vEdge, 0, info.num_edges, pairwise, &err)) return fail(ADMM_GF_E_IO, err);
  } else {
    // one table for every edge: replicate it block by block
    const int64_t blk = std::max<int64_t>(1, (4 << 20) / std::max(1, info.k2));
    std::vector<float> rep((size_t)(std::min<int64_t>(blk, info.num_edges) * info.k2));
    for (size_t i = 0; i < rep.size(); i++) rep[i] = pairwise[i % (size_t)info.k2];
    for (int64_t e = 0; e < info.num_edges; e += blk)
      if (!w.write_rows(vEdge, e, std::min<int64_t>(blk, info.num_edges - e), rep.data(), &err)) return fail(ADMM_GF_E_IO, err);
  }
  if (!w.close(&err)) return fail(ADMM_GF_E_IO, err);
  return ADMM_GF_OK;
}

template <typename T>
bool read_raw(const std::string &path, std::vector<T> *out, size_t n, std::string *err) {
  out->assign(n, T());
  std::ifstream f(path.c_str(), std::ios::in | std::ios::binary);
  if (!f.is_open()) { *err = "cannot open '" + path + "'"; return false; }
  f.read(reinterpret_cast<char *>(out->data()), (std::streamsize)(n * sizeof(T)));
  if ((size_t)f.gcount() != n * sizeof(T)) {
    *err = "'" + path + "' holds " + std::to_string((size_t)f.gcount()) + " bytes, " + std::to_string(n * sizeof(T)) + " expected";
    return false;
  }
  return true;
}

}  // namespace

extern "C" {

const char *admm_graphfile_last_error(void) { return g_err.c_str(); }

int admm_graphfile_read_info(const char *path, admm_graphfile_info *info) {
  if (!path || !info) return fail(ADMM_GF_E_ARG, "null argument");
  ncfile::Header h;
  return load_info(path, &h, info);
}

int admm_graphfile_read_edges(const char *path, int64_t first, int64_t count, int32_t *edge_list) {
  return read_var(path, "edgeList", first, count, edge_list, false);
}

int admm_graphfile_read_unaries(const char *path, int64_t first, int64_t count, float *out, int negate) {
  return read_var(path, "verticePotentials", first, count, out, negate != 0);
}

int admm_graphfile_read_pairwise(const char *path, int64_t first, int64_t count, float *out, int negate) {
  return read_var(path, "edgePotentials", first, count, out, negate != 0);
}

int admm_graphfile_read_unaries_block(const char *path, int32_t grid_cols, int32_t row0, int32_t nrows,
                                      int32_t col0, int32_t ncols, float *out, int negate) {
  if (!path || (!out && nrows > 0 && ncols > 0)) return fail(ADMM_GF_E_ARG, "null argument");
  if (grid_cols < 1 || row0 < 0 || nrows < 0 || col0 < 0 || ncols < 0 || (int64_t)col0 + ncols > grid_cols)
    return fail(ADMM_GF_E_ARG, "bad rectangle");
  ncfile::Header h;
  std::string err;
  admm_graphfile_info info;
  const int rc = load_info(path, &h, &info);
  if (rc) return rc;
  const ncfile::Var *v = h.find_var("verticePotentials");
  if (!v || v->type != ncfile::NC_FLOAT || !shape_ok(h, info, *v))
    return fail(ADMM_GF_E_FORMAT, std::string("variable 'verticePotentials' of '") + path + "' does not have the shape the file's sizes state");
  if (((int64_t)row0 + nrows) * grid_cols > info.num_vertices) return fail(ADMM_GF_E_ARG, "the rectangle reaches past the last vertex");
  const int64_t w = (int64_t)ncols * info.k;
  if (ncols == grid_cols) {  // whole rows are contiguous
    if (!ncfile::read_rows(path, h, "verticePotentials", (int64_t)row0 * grid_cols, (int64_t)nrows * grid_cols, out, &err))
      return fail(ADMM_GF_E_IO, err);
  } else {
    for (int32_t r = 0; r < nrows; r++)
      if (ncols > 0 && !ncfile::read_rows(path, h, "verticePotentials", (int64_t)(row0 + r) * grid_cols + col0, ncols, out + r * w, &err))
        return fail(ADMM_GF_E_IO, err);
  }
  if (negate)
    for (int64_t i = 0; i < w * nrows; i++) out[i] = -out[i];  // Grid.cpp:547-560
  return ADMM_GF_OK;
}

int admm_graphfile_shared_table(const char *path, float *table, int negate, int *shared) {
  if (!path || !shared) return fail(ADMM_GF_E_ARG, "null argument");
  ncfile::Header h;
  admm_graphfile_info info;
  int rc = load_info(path, &h, &info);
  if (rc) return rc;
  *shared = 0;
  if (info.num_edges < 1 || info.k2 < 1) return fail(ADMM_GF_E_FORMAT, std::string("'") + path + "' has no edges");
  const ncfile::Var *ev = h.find_var("edgePotentials");
  if (!ev || ev->type != ncfile::NC_FLOAT || !shape_ok(h, info, *ev))
    return fail(ADMM_GF_E_FORMAT, std::string("variable 'edgePotentials' of '") + path + "' does not have the shape the file's sizes state");
  std::string err;
  std::vector<float> first((size_t)info.k2);
  if (!ncfile::read_rows(path, h, "edgePotentials", 0, 1, first.data(), &err)) return fail(ADMM_GF_E_IO, err);
  const int64_t blk = std::max<int64_t>(1, (16 << 20) / info.k2);
  std::vector<float> buf((size_t)(std::min<int64_t>(blk, info.num_edges) * info.k2));
  bool same = true;
  for (int64_t e = 0; e < info.num_edges && same; e += blk) {
    const int64_t n = std::min<int64_t>(blk, info.num_edges - e);
    if (!ncfile::read_rows(path, h, "edgePotentials", e, n, buf.data(), &err)) return fail(ADMM_GF_E_IO, err);
    for (int64_t i = 0; i < n && same; i++)
      same = memcmp(buf.data() + i * info.k2, first.data(), (size_t)info.k2 * 4) == 0;
  }
  *shared = same ? 1 : 0;
  if (table)
    for (int i = 0; i < info.k2; i++) table[i] = negate ? -first[(size_t)i] : first[(size_t)i];
  return ADMM_GF_OK;
}

int admm_graphfile_write(const char *path, const admm_graphfile_info *info, const int32_t *edge_list,
                         const float *unary, const float *pairwise, int pairwise_per_edge) {
  if (!path || !info || !pairwise || (!edge_list && info->num_edges > 0) || (!unary && info->num_vertices > 0))
    return fail(ADMM_GF_E_ARG, "null argument");
  if (info->num_vertices < 0 || info->num_edges < 0 || info->k < 1 || info->k2 < 1 ||
      info->num_vertices > 0x7FFFFFFFll || info->num_edges > 0x7FFFFFFFll)
    return fail(ADMM_GF_E_ARG, "bad sizes");
  if (info->version != 0 && info->version != 1 && info->version != 2 && info->version != 5)
    return fail(ADMM_GF_E_ARG, "version must be 0, 1, 2 or 5");
  return write_graph(path, *info, edge_list, unary, pairwise, pairwise_per_edge != 0);
}

int admm_graphfile_convert(const char *manifest_path, const char *out_path) {
  if (!manifest_path || !out_path) return fail(ADMM_GF_E_ARG, "null argument");
  // the manifest: four file names, one per line (Grid.cpp:993-1007)
  std::ifstream mf(manifest_path);
  if (!mf.is_open()) return fail(ADMM_GF_E_IO, std::string("cannot open '") + manifest_path + "'");
  std::string names[4];
  for (int i = 0; i < 4; i++) {
    if (!std::getline(mf, names[i])) return fail(ADMM_GF_E_FORMAT, std::string("'") + manifest_path + "' must name four files, one per line");
    while (!names[i].empty() && (names[i].back() == '\r' || names[i].back() == ' ')) names[i].pop_back();
  }
  // info: vertices, edges, labels (Grid.cpp:1009-1013)
  std::ifstream inf(names[0].c_str());
  if (!inf.is_open()) return fail(ADMM_GF_E_IO, "cannot open '" + names[0] + "'");
  long long V = -1, E = -1, k = -1;
  inf >> V >> E >> k;
  if (!inf || V < 0 || E < 0 || k < 1 || V > 0x7FFFFFFFll || E > 0x7FFFFFFFll || k > 46340)
    return fail(ADMM_GF_E_FORMAT, "'" + names[0] + "' must hold three numbers: vertices, edges, labels");
  std::string err;
  std::vector<int32_t> edges;
  std::vector<float> node, edge;
  if (!read_raw(names[1], &edges, (size_t)(2 * E), &err) || !read_raw(names[2], &node, (size_t)(V * k), &err) ||
      !read_raw(names[3], &edge, (size_t)(E * k * k), &err))
    return fail(ADMM_GF_E_IO, err);
  admm_graphfile_info info;
  info.version = 0;
  info.rows = info.cols = -1;  // Grid::Grid() leaves them at -1 (Grid.cpp:85-86)
  info.num_vertices = V;
  info.num_edges = E;
  info.k = (int32_t)k;
  info.k2 = (int32_t)(k * k);  // Grid.cpp:1015
  return write_graph(out_path, info, edges.data(), node.data(), edge.data(), true);
}

}  // extern "C"
