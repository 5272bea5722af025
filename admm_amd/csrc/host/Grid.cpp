#include "Grid.h"

#include <algorithm>
#include <cmath>

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

void Grid::SetGlobalInfo(const std::string &fname) {
  throw AdmmError(ADMM_E_ARG, "reading '" + fname + "': the CDF-5 reader is not built yet; use SetGlobalInfo(rows, cols, k, k*k) and Generate");
}

void Grid::Load(const std::string &fname) {
  throw AdmmError(ADMM_E_ARG, "reading '" + fname + "': the CDF-5 reader is not built yet; use Generate or SetPotentials");
}

void Grid::Save(const std::string &fname) {
  throw AdmmError(ADMM_E_ARG, "writing '" + fname + "': the CDF-5 writer is not built yet");
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
}

void Grid::SetPotentials(const std::vector<float> &unary, const std::vector<float> &pairwise, bool perEdge) {
  vertPotentials = unary;
  edgePotentials = pairwise;
  sharedTable = !perEdge;
}

void Grid::InitOptimization() {
  if (vertPotentials.empty()) throw AdmmError(ADMM_E_STATE, "Generate or SetPotentials before InitOptimization");
  if (bbox.getBoxRowSize() == 0 && bbox.getBoxColSize() == 0) Partition();
  admm_desc d{};
  d.rows = numGlobalRows; d.cols = numGlobalCols; d.k = numVertPotentials; d.batch = batch;
  d.solver = part == EDGE ? ADMM_SOLVER_EDGE_BETHE : ADMM_SOLVER_CHAIN_MINSUM;
  d.penalty = penalty; d.penalty_prime = penaltyPrime; d.rho = rhoEdge;
  d.device = device; d.rank = part == EDGE ? 0 : rank; d.nranks = part == EDGE ? 1 : numProcs;
  d.exchange = ADMM_EXCHANGE_LABELS;
  if (ctx) { admm_destroy(ctx); ctx = nullptr; }
  const int rc = admm_create(&d, &ctx);
  if (rc != ADMM_OK) throw AdmmError(rc, std::string("admm_create: ") + admm_last_error(nullptr));
  check(admm_upload_unaries(ctx, vertPotentials.data()), "admm_upload_unaries");
  check(admm_upload_pairwise(ctx, edgePotentials.data(), sharedTable ? 0 : 1), "admm_upload_pairwise");
  check(admm_init(ctx), "admm_init");
}

void Grid::Compute() { check(admm_compute(ctx), "admm_compute"); }
void Grid::Communicate() { check(admm_communicate(ctx), "admm_communicate"); }
void Grid::Update() { check(admm_update(ctx), "admm_update"); }

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

void Grid::CommunicateGroup(std::vector<Grid *> &strips) {
  std::vector<admm_ctx *> h;
  for (Grid *g : strips) h.push_back(g->ctx);
  const int rc = admm_communicate_group(h.data(), (int)h.size());
  if (rc != ADMM_OK) throw AdmmError(rc, std::string("admm_communicate_group: ") + admm_last_error(h.empty() ? nullptr : h[0]));
}
