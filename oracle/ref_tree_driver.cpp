/*
 * ref_tree_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points around the reference's own Edge class.  This file is linked
 * with Tree.cpp compiled where it lies under /root/reference (see Makefile,
 * target _ref); the result, oracle/_ref/libref_tree.so, is used to bit-compare
 * the restatement in admm_oracle.c and to generate tests/golden.  Tree.cpp is
 * the only reference source that builds here: Grid.cpp needs pnetcdf.h, which
 * this image lacks, so the loop around the Edge calls below is this file's
 * own (it follows Grid.cpp:1394-1412, 163-193, 934-980, 1474-1491, 1665-1680
 * and is itself pinned by the totalObj trace the unmodified reference printed).
 */
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Tree.h" /* -I/root/reference */

namespace {

struct RefGrid {
  int V, E, k, k2;
  float penalty, penaltyPrime, rho;
  std::vector<int> edgeList, vertCnt;
  std::vector<float> vertPot, edgePot;
  std::vector<Edge *> edgeVal;
  std::vector<float *> vertMuArr;
  APGMem *aux;
  EdgeOp *edgeOp;
};

void average(RefGrid *g) { /* Grid.cpp:176-180 */
  for (int i = 0; i < g->V; i++)
    for (int j = 0; j < g->k; j++) g->vertMuArr[i][j] /= g->vertCnt[i];
}

}  // namespace

extern "C" {

void *ref_create(int V, int E, int k, const int *edgeList, const float *vertPot,
                 const float *edgePot, float penalty, float penaltyPrime, float rho) {
  RefGrid *g = new RefGrid;
  g->V = V; g->E = E; g->k = k; g->k2 = k * k;
  g->penalty = penalty; g->penaltyPrime = penaltyPrime; g->rho = rho;
  g->edgeList.assign(edgeList, edgeList + 2 * (size_t)E);
  g->vertPot.assign(vertPot, vertPot + (size_t)V * k);
  g->edgePot.assign(edgePot, edgePot + (size_t)E * k * k);
  g->vertCnt.assign(V, 0);
  g->edgeVal.resize(E);
  for (int i = 0; i < E; i++) g->edgeVal[i] = new Edge(k);
  g->vertMuArr.resize(V);
  for (int i = 0; i < V; i++) g->vertMuArr[i] = new float[k];
  g->aux = new APGMem(k);
  g->edgeOp = new EdgeOp(k);
  return g;
}

void ref_destroy(void *h) {
  RefGrid *g = (RefGrid *)h;
  for (size_t i = 0; i < g->edgeVal.size(); i++) delete g->edgeVal[i];
  for (size_t i = 0; i < g->vertMuArr.size(); i++) delete[] g->vertMuArr[i];
  delete g->aux;
  /* EdgeOp's destructor pairs new[] with delete (Tree.cpp:284-285); leak it. */
  delete g;
}

void ref_init(void *h) {
  RefGrid *g = (RefGrid *)h;
  const int k = g->k;
  for (int i = 0; i < 2 * g->E; i++) g->vertCnt[g->edgeList[i]]++;
  for (int i = 0; i < g->V; i++) memset(g->vertMuArr[i], 0, sizeof(float) * k);
  for (int i = 0; i < 2 * g->E; i++) { /* Grid.cpp:1479-1484 */
    int vid = g->edgeList[i];
    for (int j = 0; j < k; j++) g->vertMuArr[vid][j] += g->vertPot[(size_t)vid * k + j];
  }
  average(g);
  for (int i = 0; i < g->E; i++) { /* Grid.cpp:944-979 */
    int l1 = g->edgeList[2 * i], l2 = g->edgeList[2 * i + 1];
    Edge *e = g->edgeVal[i];
    for (int j = 0; j < k; j++) {
      e->x_potential_node[j] = g->vertMuArr[l1][j] / (g->rho * g->vertCnt[l1]);
      e->x_potential_node[k + j] = g->vertMuArr[l2][j] / (g->rho * g->vertCnt[l2]);
      e->x_mu_node[j] = 1.0 / (float)k;
      e->edge_mu_node[j] = 1.0 / (float)k;
    }
    for (int j = 0; j < g->k2; j++) {
      e->x_potential_edge[j] = g->edgePot[(size_t)g->k2 * i + j] / g->rho;
      e->x_mu_edge[j] = 1.0 / (float)g->k2;
    }
  }
}

void ref_compute(void *h) {
  RefGrid *g = (RefGrid *)h;
  for (int i = 0; i < g->V; i++) memset(g->vertMuArr[i], 0, sizeof(float) * g->k);
  for (int i = 0; i < g->E; i++) { /* Grid.cpp:1400-1410 */
    g->edgeVal[i]->UpdateLambda(g->penalty);
    g->edgeVal[i]->AdjustPotentials(g->edgeOp, g->penalty, g->rho);
    g->edgeVal[i]->OptimizeEdge(g->edgeOp, g->aux, g->vertMuArr, g->edgeList[2 * i],
                                g->edgeList[2 * i + 1], g->penaltyPrime);
  }
}

void ref_update(void *h) {
  RefGrid *g = (RefGrid *)h;
  average(g);
  for (int i = 0; i < 2 * g->E; i++) /* Grid.cpp:188-192 */
    memcpy(g->edgeVal[i / 2]->edge_mu_node + (i % 2) * g->k, g->vertMuArr[g->edgeList[i]],
           g->k * sizeof(float));
}

double ref_objective(void *h) {
  RefGrid *g = (RefGrid *)h;
  double obj = 0.0; /* Grid.cpp:1667-1679 */
  for (int i = 0; i < g->E; i++) {
    for (int j = 0; j < g->k * 2; j++)
      obj += g->edgeVal[i]->x_potential_node[j] * g->edgeVal[i]->x_mu_node[j];
    for (int j = 0; j < g->k2; j++)
      obj += g->edgeVal[i]->x_potential_edge[j] * g->edgeVal[i]->x_mu_edge[j];
  }
  return obj * g->rho;
}

/* which: 0 x_mu_node, 1 x_potential_node, 2 edge_mu_node, 3 lambda_node (E*2k);
 * 4 x_mu_edge, 5 x_potential_edge (E*k2); 6 vertMuArr (V*k). */
void ref_get(void *h, int which, float *out) {
  RefGrid *g = (RefGrid *)h;
  const int k = g->k, k2 = g->k2;
  if (which == 6) {
    for (int i = 0; i < g->V; i++) memcpy(out + (size_t)i * k, g->vertMuArr[i], sizeof(float) * k);
    return;
  }
  for (int i = 0; i < g->E; i++) {
    Edge *e = g->edgeVal[i];
    switch (which) {
      case 0: memcpy(out + (size_t)i * 2 * k, e->x_mu_node, sizeof(float) * 2 * k); break;
      case 1: memcpy(out + (size_t)i * 2 * k, e->x_potential_node, sizeof(float) * 2 * k); break;
      case 2: memcpy(out + (size_t)i * 2 * k, e->edge_mu_node, sizeof(float) * 2 * k); break;
      case 3: memcpy(out + (size_t)i * 2 * k, e->lambda_node, sizeof(float) * 2 * k); break;
      case 4: memcpy(out + (size_t)i * k2, e->x_mu_edge, sizeof(float) * k2); break;
      case 5: memcpy(out + (size_t)i * k2, e->x_potential_edge, sizeof(float) * k2); break;
    }
  }
}

void ref_get_vertcnt(void *h, int *out) {
  RefGrid *g = (RefGrid *)h;
  memcpy(out, g->vertCnt.data(), sizeof(int) * g->V);
}

/* One subproblem through the reference's Edge methods on caller data. */
void ref_solve_one(int k, float *mu, const float *pot, const float *m, float *lam,
                   float *mue, const float *pote, float penalty, float penaltyPrime,
                   float rho, float *acc_l, float *acc_r) {
  Edge e(k);
  APGMem aux(k);
  EdgeOp *op = new EdgeOp(k);
  memcpy(e.x_mu_node, mu, sizeof(float) * 2 * k);
  memcpy(e.x_potential_node, pot, sizeof(float) * 2 * k);
  memcpy(e.edge_mu_node, m, sizeof(float) * 2 * k);
  memcpy(e.lambda_node, lam, sizeof(float) * 2 * k);
  memcpy(e.x_mu_edge, mue, sizeof(float) * k * k);
  memcpy(e.x_potential_edge, pote, sizeof(float) * k * k);
  std::vector<float *> acc(2);
  acc[0] = acc_l; acc[1] = acc_r;
  e.UpdateLambda(penalty);
  e.AdjustPotentials(op, penalty, rho);
  e.OptimizeEdge(op, &aux, acc, 0, 1, penaltyPrime);
  memcpy(mu, e.x_mu_node, sizeof(float) * 2 * k);
  memcpy(lam, e.lambda_node, sizeof(float) * 2 * k);
  memcpy(mue, e.x_mu_edge, sizeof(float) * k * k);
}

}  // extern "C"
