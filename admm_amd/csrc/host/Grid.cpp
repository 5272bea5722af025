#include "Grid.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

void Grid::check(int rc, const char *what) const {
  if (rc == ADMM_OK) return;
  const char *msg = admm_last_error(ctx);
  throw AdmmError(rc, std::string(what) + ": " + (msg ? msg : "unknown error"));
}

Grid::Grid(int r, int n, int dev) : rank(r), numProcs(n), device(dev) {}

Grid::~Grid() {
  if (ctx) admm_destroy(ctx);
}

void Grid::SetGlobalInfo(int rows, int cols, int vps, int eps) {
  if (eps != vps * vps) throw AdmmError(ADMM_E_ARG, "numEdgePotentials must be numVertPotentials squared");
  numGlobalRows = rows;
  numGlobalCols = cols;
  numVertPotentials = vps;
  numEdgePotentials = eps;
  numGlobalVerts = rows * cols;
  numGlobalEdges = rows * (cols - 1) + cols * (rows - 1);
}

static void gf_check(int rc, const char *what) {
  if (rc != ADMM_GF_OK) throw AdmmError(ADMM_E_ARG, std::string(what) + ": " + admm_graphfile_last_error());
}

void Grid::SetGlobalInfo(const std::string &fname) {
  admm_graphfile_info info;
  gf_check(admm_graphfile_read_info(fname.c_str(), &info), "SetGlobalInfo");
  numGlobalRows = info.rows;
  numGlobalCols = info.cols;
  numVertPotentials = info.k;
  numEdgePotentials = info.k2;
  numGlobalVerts = (int)info.num_vertices;  // Grid.cpp:680-681: from the file, not rows * cols
  numGlobalEdges = (int)info.num_edges;
}

void Grid::Load(const std::string &fname) {
  if (numVertPotentials <= 0) SetGlobalInfo(fname);
  const int k = numVertPotentials;
  if (numEdgePotentials != k * k) throw AdmmError(ADMM_E_ARG, "numEdgePotentials must be numVertPotentials squared");
  edgeList.assign((size_t)2 * numGlobalEdges, 0);
  gf_check(admm_graphfile_read_edges(fname.c_str(), 0, numGlobalEdges, edgeList.data()), "Load");
  unaryFile.clear();
  if (part == ROW && numProcs > 1) {
    unaryFile = fname;  // this strip's share is read at InitOptimization
    vertPotentials.clear();
  } else {
    vertPotentials.assign((size_t)numGlobalVerts * k, 0.0f);
    gf_check(admm_graphfile_read_unaries(fname.c_str(), 0, numGlobalVerts, vertPotentials.data(), 1), "Load");
  }
  if (part == EDGE) {
    edgePotentials.assign((size_t)numGlobalEdges * k * k, 0.0f);
    gf_check(admm_graphfile_read_pairwise(fname.c_str(), 0, numGlobalEdges, edgePotentials.data(), 1), "Load");
    sharedTable = false;
  } else {
    // chains: the grid's own edge list, one table for all edges
    if (numGlobalRows < 1 || numGlobalCols < 1 || (long long)numGlobalRows * numGlobalCols != numGlobalVerts)
      throw AdmmError(ADMM_E_ARG, "'" + fname + "' does not describe a rows x cols grid; use the EDGE partitioner");
    std::vector<int> want((size_t)2 * std::max(1, numGlobalRows * (numGlobalCols - 1) + numGlobalCols * (numGlobalRows - 1)));
    const int E = admm_grid_edges(numGlobalRows, numGlobalCols, want.data());
    want.resize((size_t)2 * E);
    if (E != numGlobalEdges || want != edgeList)
      throw AdmmError(ADMM_E_ARG, "the edge list of '" + fname + "' is not the grid's (Grid::Generate order); use the EDGE partitioner");
    edgePotentials.assign((size_t)k * k, 0.0f);
    int shared = 0;
    gf_check(admm_graphfile_shared_table(fname.c_str(), edgePotentials.data(), 1, &shared), "Load");
    if (!shared) throw AdmmError(ADMM_E_ARG, "the edges of '" + fname + "' carry different pairwise tables; the chain solver needs one shared table");
    sharedTable = true;
  }
  batch = 1;
  fromFile = true;
}

void Grid::Save(const std::string &fname) {
  if (vertPotentials.empty() && unaryFile.empty()) throw AdmmError(ADMM_E_STATE, "Generate, Load or Convert before Save");
  if (batch != 1) throw AdmmError(ADMM_E_ARG, "a graph file holds one graph (batch == 1)");
  const int k = numVertPotentials;
  // A strip of several that loaded from a file keeps no host copy of the unaries (it reads its
  // share when it uploads it): they are read back from that file here, as every rank of the
  // reference can save what it loaded (Grid.cpp:1049-1131).
  std::vector<float> fromStripFile;
  if (vertPotentials.empty()) {
    fromStripFile.assign((size_t)numGlobalVerts * k, 0.0f);
    gf_check(admm_graphfile_read_unaries(unaryFile.c_str(), 0, numGlobalVerts, fromStripFile.data(), 1), "Save");
  }
  const std::vector<float> &vertPotentials = fromStripFile.empty() ? this->vertPotentials : fromStripFile;
  std::vector<int> el = edgeList;
  if (el.empty()) {
    el.resize((size_t)2 * std::max(1, numGlobalEdges));
    el.resize((size_t)2 * admm_grid_edges(numGlobalRows, numGlobalCols, el.data()));
  }
  // the file holds scores: the negated costs (Grid.cpp:547-560 undone)
  std::vector<float> un(vertPotentials.size()), pw(edgePotentials.size());
  for (size_t i = 0; i < un.size(); i++) un[i] = -vertPotentials[i];
  for (size_t i = 0; i < pw.size(); i++) pw[i] = -edgePotentials[i];
  admm_graphfile_info info;
  info.version = 0;
  info.rows = numGlobalRows;
  info.cols = numGlobalCols;
  info.num_vertices = numGlobalVerts;
  info.num_edges = numGlobalEdges;
  info.k = k;
  info.k2 = k * k;
  gf_check(admm_graphfile_write(fname.c_str(), &info, el.data(), un.data(), pw.data(), sharedTable ? 0 : 1), "Save");
}

void Grid::Convert(const std::string &manifest) {
  // through a temporary graph file: the converter of the library is the one implementation
  const std::string tmp = manifest + ".converted.nc";
  gf_check(admm_graphfile_convert(manifest.c_str(), tmp.c_str()), "Convert");
  const Partitioner keep = part;
  part = EDGE;
  numVertPotentials = 0;
  try {
    Load(tmp);
  } catch (...) {
    part = keep;
    remove(tmp.c_str());
    throw;
  }
  part = keep;
  remove(tmp.c_str());
  // The raw files hold what a graph file holds (scores): this object now has
  // their negation, as after any Load, and Save negates back, so that
  // Convert + Save writes the raw values unchanged (Grid.cpp:1032-1044, 1110-1126).
}

void Grid::SetPartitioner(Partitioner aPart) {
  if (aPart != ROW && aPart != EDGE) throw AdmmError(ADMM_E_ARG, "no implementation for COL / BLOCK");
  part = aPart;
}

void Grid::Partition() {
  int r0 = 0, nr = numGlobalRows;
  if (part == ROW) admm_partition(numGlobalRows, numProcs, rank, &r0, &nr);
  bbox.setBBox(r0, 0, nr, numGlobalCols);
}

void Grid::Generate(unsigned seed, const std::string &pairwise) {
  const int k = numVertPotentials;
  const size_t n = (size_t)numGlobalVerts * k;
  vertPotentials.resize(n * batch);
  uint32_t first_state = 0;
  for (int g = 0; g < batch; g++) {
    const uint32_t st = admm_lcg_fill(seed + g, (int64_t)n, vertPotentials.data() + n * g);
    if (g == 0) first_state = st;
  }
  edgePotentials.assign((size_t)k * k, 0.0f);
  for (int a = 0; a < k; a++)
    for (int b = 0; b < k; b++) {
      float w;
      if (pairwise == "potts") w = a == b ? 0.0f : 0.5f;
      else if (pairwise == "trunc_linear") w = std::min(std::abs(a - b), 4) * 0.25f;
      else if (pairwise == "dense") w = 0.0f;
      else throw AdmmError(ADMM_E_ARG, "unknown pairwise kind " + pairwise);
      edgePotentials[(size_t)a * k + b] = w;
    }
  if (pairwise == "dense") admm_lcg_fill(first_state, (int64_t)k * k, edgePotentials.data());
  sharedTable = true;
  fromFile = false;
  unaryFile.clear();
  edgeList.clear();
}

void Grid::SetPotentials(const std::vector<float> &unary, const std::vector<float> &pairwise, bool perEdge) {
  vertPotentials = unary;
  edgePotentials = pairwise;
  sharedTable = !perEdge;
  unaryFile.clear();
}

void Grid::UploadSharesFromFile() {
  int r0, nr, c0, nc;
  check(admm_strip_info(ctx, &r0, &nr, &c0, &nc), "admm_strip_info");
  const int k = numVertPotentials, cols = numGlobalCols, rows = numGlobalRows;
  const size_t budget = (size_t)64 << 20;  // bytes of host staging
  std::vector<float> buf;
  int step = (int)std::max<size_t>(1, budget / (sizeof(float) * k * std::max(cols, 1)));
  for (int r = r0; r < r0 + nr; r += step) {
    const int n = std::min(step, r0 + nr - r);
    buf.resize((size_t)n * cols * k);
    gf_check(admm_graphfile_read_unaries_block(unaryFile.c_str(), cols, r, n, 0, cols, buf.data(), 1), "Load");
    check(admm_upload_unaries_rows(ctx, r, n, buf.data()), "admm_upload_unaries_rows");
  }
  step = (int)std::max<size_t>(1, budget / (sizeof(float) * k * std::max(nc, 1)));
  for (int r = 0; r < rows; r += step) {
    const int n = std::min(step, rows - r);
    buf.resize((size_t)n * nc * k);
    gf_check(admm_graphfile_read_unaries_block(unaryFile.c_str(), cols, r, n, c0, nc, buf.data(), 1), "Load");
    check(admm_upload_unaries_cols(ctx, r, n, buf.data()), "admm_upload_unaries_cols");
  }
}

void Grid::InitOptimization() {
  if (vertPotentials.empty() && unaryFile.empty()) throw AdmmError(ADMM_E_STATE, "Generate or SetPotentials before InitOptimization");
  if (bbox.getBoxRowSize() == 0 && bbox.getBoxColSize() == 0) Partition();
  admm_desc d{};
  d.rows = numGlobalRows; d.cols = numGlobalCols; d.k = numVertPotentials; d.batch = batch;
  d.solver = part == EDGE ? ADMM_SOLVER_EDGE_BETHE : ADMM_SOLVER_CHAIN_MINSUM;
  d.penalty = penalty; d.penalty_prime = penaltyPrime; d.rho = rhoEdge;
  d.device = device; d.rank = rank; d.nranks = numProcs;
  d.exchange = ADMM_EXCHANGE_LABELS;
  if (part == EDGE && fromFile) {  // any graph: hand the file's edge list over (Grid.cpp:476-494)
    d.edge_list = edgeList.data();
    d.num_vertices = numGlobalVerts;
    d.num_edges = numGlobalEdges;
    if (d.rows < 1) d.rows = 1;
    if (d.cols < 1) d.cols = 1;
  }
  if (ctx) { admm_destroy(ctx); ctx = nullptr; }
  const int rc = admm_create(&d, &ctx);
  if (rc != ADMM_OK) throw AdmmError(rc, std::string("admm_create: ") + admm_last_error(nullptr));
  if (!unaryFile.empty()) UploadSharesFromFile();
  else check(admm_upload_unaries(ctx, vertPotentials.data()), "admm_upload_unaries");
  check(admm_upload_pairwise(ctx, edgePotentials.data(), sharedTable ? 0 : 1), "admm_upload_pairwise");
  check(admm_init(ctx), "admm_init");
}

void Grid::Compute() { check(admm_compute(ctx), "admm_compute"); }
void Grid::Communicate() { check(admm_communicate(ctx), "admm_communicate"); }
void Grid::Update() { check(admm_update(ctx), "admm_update"); }
void Grid::Synchronize() { check(admm_synchronize(ctx), "admm_synchronize"); }

double Grid::ComputeLinearObj() {
  double v = 0.0;
  check(admm_linear_objective(ctx, &v), "admm_linear_objective");
  return v;
}

void Grid::FinalizeOptimization() {
  if (ctx) check(admm_finalize(ctx), "admm_finalize");
}

void Grid::Objective(double *primal, double *dual) { check(admm_objective(ctx, primal, dual), "admm_objective"); }

std::vector<int> Grid::Labels() {
  int r0, nr, c0, nc;
  check(admm_strip_info(ctx, &r0, &nr, &c0, &nc), "admm_strip_info");
  std::vector<int> out((size_t)batch * nr * numGlobalCols);
  check(admm_download_labels(ctx, out.data()), "admm_download_labels");
  return out;
}

void Grid::Trees(TreeSnapshot &out) {
  if (!ctx) throw AdmmError(ADMM_E_STATE, "InitOptimization before Trees");
  out.Download(ctx, numVertPotentials);
}

void Grid::CommunicateGroup(std::vector<Grid *> &strips) {
  std::vector<admm_ctx *> h;
  for (Grid *g : strips) h.push_back(g->ctx);
  const int rc = admm_communicate_group(h.data(), (int)h.size());
  if (rc != ADMM_OK) throw AdmmError(rc, std::string("admm_communicate_group: ") + admm_last_error(h.empty() ? nullptr : h[0]));
}
