// Host-side trees with the reference's class and member names (class Edge,
// Tree.h:20-51).  In the reference every tree owns six heap arrays and
// Grid::Compute calls its methods one tree after the other
// (Grid.cpp:1400-1410); here the trees' variables live in device arrays
// (structure of arrays over all trees) and one kernel launch runs
// UpdateLambda, AdjustPotentials and OptimizeEdge (Tree.cpp:64-274) for all of
// them.  These classes are what a caller of the reference's public members gets
// instead: a snapshot of the solver state, downloaded through
// admm_download_state, addressed tree by tree under the reference's names.
// They hold no arithmetic; there is no host implementation of the solvers.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/admm_hip.h"

#define NUM_NODE 2  // Tree.h:13
#define NUM_EDGE 1  // Tree.h:14

// One tree of the edge solver: the members of the reference's class Edge that
// carry state (Tree.h:22-35), as read-only views into a TreeSnapshot.
class Edge {
 public:
  int k = 0;        // labels per node
  int ksquare = 0;  // entries of the pairwise table
  int nedges = NUM_EDGE;
  // node: endpoint 1 then endpoint 2, k entries each
  const float *x_mu_node = nullptr;         // marginals of the tree's own solution
  const float *x_potential_node = nullptr;  // the tree's share of the unaries
  const float *edge_mu_node = nullptr;      // consensus copied to the tree by Update
  // edge: [a * k + b], a = label of endpoint 1
  const float *x_mu_edge = nullptr;
  const float *x_potential_edge = nullptr;
  // lambda
  const float *lambda_node = nullptr;
};

// One tree of the chain solver (a row chain or a column chain of the grid): its
// nodes' labels, multipliers and, for row chains, shares of the unaries.
class Chain {
 public:
  int k = 0;
  int length = 0;        // nodes
  bool column = false;   // a column chain of the strip's columns, else a row chain of its rows
  int index = 0;         // global row or column
  long long stride = 1;  // elements between consecutive nodes in the arrays below (in nodes)
  const int *x = nullptr;               // labels, node t at x[t * stride]; -1 before the first Compute
  const float *lambda_node = nullptr;   // node t at lambda_node[t * stride * k .. + k)
  const float *x_potential_node = nullptr;  // row chains only (the column chains' shares are the same numbers)
};

// Snapshot of every tree of one context.
class TreeSnapshot {
  std::vector<float> f_[6];
  std::vector<int> xi_[2];
  std::vector<float> lam_[2], th_;
  std::vector<Edge> edges_;
  std::vector<Chain> chains_;

  static void check(admm_ctx *ctx, int rc, const char *what) {
    if (rc != ADMM_OK) throw std::runtime_error(std::string(what) + ": " + admm_last_error(ctx));
  }

 public:
  // Downloads the state of ctx's trees (waits for the context's stream).
  void Download(admm_ctx *ctx, int k) {
    edges_.clear();
    chains_.clear();
    if (admm_state_size(ctx, 0) > 0 || admm_state_size(ctx, 4) > 0) {
      for (int w = 0; w < 6; w++) {
        f_[w].assign(admm_state_size(ctx, w), 0.0f);
        check(ctx, admm_download_state(ctx, w, f_[w].data()), "admm_download_state");
      }
      const size_t E = f_[0].size() / (2 * (size_t)k);
      edges_.resize(E);
      for (size_t e = 0; e < E; e++) {
        Edge &t = edges_[e];
        t.k = k;
        t.ksquare = k * k;
        t.x_mu_node = f_[0].data() + e * 2 * k;
        t.x_potential_node = f_[1].data() + e * 2 * k;
        t.edge_mu_node = f_[2].data() + e * 2 * k;
        t.lambda_node = f_[3].data() + e * 2 * k;
        t.x_mu_edge = f_[4].data() + e * (size_t)k * k;
        t.x_potential_edge = f_[5].data() + e * (size_t)k * k;
      }
      return;
    }
    int r0 = 0, nr = 0, c0 = 0, nc = 0;
    check(ctx, admm_strip_info(ctx, &r0, &nr, &c0, &nc), "admm_strip_info");
    const size_t nrow = admm_state_size(ctx, 13), ncol = admm_state_size(ctx, 14);
    if (nrow == 0 && ncol == 0) return;
    const int cols = nr > 0 ? (int)(nrow / (size_t)nr) : 0, rows = nc > 0 ? (int)(ncol / (size_t)nc) : 0;
    lam_[0].assign(admm_state_size(ctx, 10), 0.0f);
    lam_[1].assign(admm_state_size(ctx, 11), 0.0f);
    th_.assign(admm_state_size(ctx, 12), 0.0f);
    xi_[0].assign(nrow, 0);
    xi_[1].assign(ncol, 0);
    if (nrow) {
      check(ctx, admm_download_state(ctx, 10, lam_[0].data()), "admm_download_state");
      check(ctx, admm_download_state(ctx, 12, th_.data()), "admm_download_state");
      check(ctx, admm_download_state(ctx, 13, reinterpret_cast<float *>(xi_[0].data())), "admm_download_state");
    }
    if (ncol) {
      check(ctx, admm_download_state(ctx, 11, lam_[1].data()), "admm_download_state");
      check(ctx, admm_download_state(ctx, 14, reinterpret_cast<float *>(xi_[1].data())), "admm_download_state");
    }
    for (int r = 0; r < nr; r++) {  // [strip rows][cols]
      Chain t;
      t.k = k; t.length = cols; t.column = false; t.index = r0 + r; t.stride = 1;
      t.x = xi_[0].data() + (size_t)r * cols;
      t.lambda_node = lam_[0].data() + (size_t)r * cols * k;
      t.x_potential_node = th_.data() + (size_t)r * cols * k;
      chains_.push_back(t);
    }
    for (int c = 0; c < nc; c++) {  // [rows][strip cols]
      Chain t;
      t.k = k; t.length = rows; t.column = true; t.index = c0 + c; t.stride = nc;
      t.x = xi_[1].data() + c;
      t.lambda_node = lam_[1].data() + (size_t)c * k;
      chains_.push_back(t);
    }
  }
  // edgeVal of the reference (Grid.h:66): this rank's edge trees in edge-list order.
  const std::vector<Edge> &edgeVal() const { return edges_; }
  // The chain solver's trees: the strip's row chains, then its column chains.
  const std::vector<Chain> &chains() const { return chains_; }
};
