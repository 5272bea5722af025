/*
 * admm_oracle.c -- TEST INFRASTRUCTURE ONLY (see admm_oracle.h).
 *
 * Scalar restatement of the reference's ADMM hot path.  Every function cites
 * the reference lines it follows.  The arithmetic types are the reference's:
 * state is binary32, log/exp are the double libm functions with the result
 * rounded to float where the reference stores into a float, and no operation
 * is fused (build with -ffp-contract=off; the reference's x86-64 build emits
 * no FMA).
 */
#include "admm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_EPSILON 0.00000000000000001 /* Tree.h:15 */

/* ------------------------------------------------------------------ */
/* synthetic inputs                                                     */
/* ------------------------------------------------------------------ */

unsigned oracle_lcg_fill(unsigned state, long long n, float *out) {
  for (long long i = 0; i < n; i++) {
    state = state * 1664525u + 1013904223u;
    out[i] = (float)(state >> 8) / 16777216.0f;
  }
  return state;
}

int oracle_grid_edges(int R, int C, int *edgeList) {
  int e = 0;
  for (int r = 0; r < R; r++) {
    for (int c = 0; c + 1 < C; c++) { /* Grid.cpp:585-596 */
      edgeList[2 * e] = r * C + c;
      edgeList[2 * e + 1] = r * C + c + 1;
      e++;
    }
    if (r == R - 1) break; /* Grid.cpp:598-600 */
    for (int c = 0; c < C; c++) { /* Grid.cpp:605-614 */
      edgeList[2 * e] = r * C + c;
      edgeList[2 * e + 1] = (r + 1) * C + c;
      e++;
    }
  }
  return e;
}

/* ------------------------------------------------------------------ */
/* edge solver                                                          */
/* ------------------------------------------------------------------ */

struct oracle_edge {
  int V, E, k, k2;
  float penalty, penaltyPrime, rho;
  int *edgeList;
  int *vertCnt;
  float *vertPot, *edgePot;
  float *mu_node, *pot_node, *m_node, *lambda_node; /* E*2k */
  float *mu_edge, *pot_edge;                        /* E*k2 */
  float *vertMu;                                    /* V*k  */
};

static float *falloc(long long n) { return (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float)); }

oracle_edge *oracle_edge_create(int V, int E, int k, const int *edgeList,
                                const float *vertPot, const float *edgePot,
                                float penalty, float penaltyPrime, float rho) {
  oracle_edge *o = (oracle_edge *)calloc(1, sizeof(*o));
  o->V = V; o->E = E; o->k = k; o->k2 = k * k;
  o->penalty = penalty; o->penaltyPrime = penaltyPrime; o->rho = rho;
  o->edgeList = (int *)malloc(sizeof(int) * 2 * (size_t)(E > 0 ? E : 1));
  memcpy(o->edgeList, edgeList, sizeof(int) * 2 * (size_t)E);
  o->vertCnt = (int *)calloc((size_t)(V > 0 ? V : 1), sizeof(int));
  o->vertPot = falloc((long long)V * k);
  memcpy(o->vertPot, vertPot, sizeof(float) * (size_t)V * k);
  o->edgePot = falloc((long long)E * o->k2);
  memcpy(o->edgePot, edgePot, sizeof(float) * (size_t)E * o->k2);
  o->mu_node = falloc((long long)E * 2 * k);
  o->pot_node = falloc((long long)E * 2 * k);
  o->m_node = falloc((long long)E * 2 * k);
  o->lambda_node = falloc((long long)E * 2 * k);
  o->mu_edge = falloc((long long)E * o->k2);
  o->pot_edge = falloc((long long)E * o->k2);
  o->vertMu = falloc((long long)V * k);
  return o;
}

void oracle_edge_destroy(oracle_edge *o) {
  if (!o) return;
  free(o->edgeList); free(o->vertCnt); free(o->vertPot); free(o->edgePot);
  free(o->mu_node); free(o->pot_node); free(o->m_node); free(o->lambda_node);
  free(o->mu_edge); free(o->pot_edge); free(o->vertMu);
  free(o);
}

/* Consensus division, Grid.cpp:176-180: float /= int. */
static void edge_average(oracle_edge *o) {
  for (int v = 0; v < o->V; v++)
    for (int j = 0; j < o->k; j++)
      o->vertMu[(size_t)v * o->k + j] /= o->vertCnt[v];
}

void oracle_edge_init(oracle_edge *o) {
  const int k = o->k, k2 = o->k2;
  /* copy count per vertex: GetVerts, Grid.cpp:815-856 (one rank: no remote
   * contribution from Grid.cpp:309-319). */
  memset(o->vertCnt, 0, sizeof(int) * (size_t)o->V);
  for (int i = 0; i < 2 * o->E; i++) o->vertCnt[o->edgeList[i]]++;
  /* Grid.cpp:1478-1484: one add of theta_v per incident endpoint. */
  memset(o->vertMu, 0, sizeof(float) * (size_t)o->V * k);
  for (int i = 0; i < 2 * o->E; i++) {
    int v = o->edgeList[i];
    for (int j = 0; j < k; j++)
      o->vertMu[(size_t)v * k + j] += o->vertPot[(size_t)v * k + j];
  }
  edge_average(o); /* Grid.cpp:1486-1487 */
  /* InitTree, Grid.cpp:944-979; initial marginals Tree.cpp:24-33. */
  const float inv_k = 1.0 / (float)k;
  const float inv_k2 = 1.0 / (float)k2;
  for (int e = 0; e < o->E; e++) {
    int l = o->edgeList[2 * e], r = o->edgeList[2 * e + 1];
    float *pn = o->pot_node + (size_t)e * 2 * k;
    for (int j = 0; j < k; j++) {
      pn[j] = o->vertMu[(size_t)l * k + j] / (o->rho * o->vertCnt[l]);
      pn[k + j] = o->vertMu[(size_t)r * k + j] / (o->rho * o->vertCnt[r]);
    }
    for (int j = 0; j < 2 * k; j++) {
      o->mu_node[(size_t)e * 2 * k + j] = inv_k;
      o->m_node[(size_t)e * 2 * k + j] = inv_k;
      o->lambda_node[(size_t)e * 2 * k + j] = 0.0f;
    }
    for (int j = 0; j < k2; j++) {
      o->pot_edge[(size_t)e * k2 + j] = o->edgePot[(size_t)e * k2 + j] / o->rho;
      o->mu_edge[(size_t)e * k2 + j] = inv_k2;
    }
  }
}

void oracle_edge_solve_one(int k, float *mu, const float *pot, const float *m,
                           float *lam, float *mue, const float *pote,
                           float penalty, float penaltyPrime, float rho,
                           float *acc_l, float *acc_r) {
  const int k2 = k * k;
  float *xnode = (float *)malloc(sizeof(float) * (size_t)(9 * k + 4 * k2));
  float *nderiv = xnode + 2 * k, *npot = nderiv + 2 * k;
  float *buf = npot + 2 * k, *m12 = buf + k, *m21 = m12 + k;
  float *xedge = m21 + k, *ederiv = xedge + k2, *epot = ederiv + k2, *bsq = epot + k2;

  /* UpdateLambda, Tree.cpp:64-73 */
  for (int j = 0; j < 2 * k; j++) lam[j] += penalty * (mu[j] - m[j]);
  /* AdjustPotentials, Tree.cpp:78-107 */
  for (int i = 0; i < 2 * k; i++)
    xnode[i] = rho * (pot[i] + lam[i]) - penalty * (m[i] - mu[i]);
  for (int i = 0; i < k2; i++) xedge[i] = rho * pote[i];
  /* UpdateMu, Tree.cpp:46-57 */
  for (int i = 0; i < 2 * k; i++) if (mu[i] <= 0.0) mu[i] = ORACLE_EPSILON;
  for (int i = 0; i < k2; i++) if (mue[i] <= 0.0) mue[i] = ORACLE_EPSILON;
  /* derivatives, Tree.cpp:118-151 */
  for (int a = 0; a < k; a++) {
    float value = log(mu[a]);
    nderiv[a] = -1.0 - value;
    for (int b = 0; b < k; b++)
      ederiv[a * k + b] = value - log(mue[a * k + b]) + 1;
  }
  for (int b = 0; b < k; b++) {
    float value = log(mu[k + b]);
    nderiv[k + b] = -1.0 - value;
    for (int a = 0; a < k; a++) ederiv[a * k + b] += value;
  }
  /* potentials, Tree.cpp:157-176 */
  for (int i = 0; i < 2 * k; i++) {
    float value = xnode[i] + penaltyPrime * nderiv[i];
    value /= penaltyPrime;
    npot[i] = 0.0 - value;
  }
  for (int i = 0; i < k2; i++) {
    float value = xedge[i] + penaltyPrime * ederiv[i];
    value /= penaltyPrime;
    epot[i] = 0.0 - value;
  }
  /* sumProductEdge, Tree.cpp:186-274 */
  float sum = 0.0;
  memcpy(bsq, epot, sizeof(float) * (size_t)k2);
  for (int b = 0; b < k; b++) { /* message 1 -> 2, Tree.cpp:191-205 */
    buf[b] = 0.0;
    for (int a = 0; a < k; a++) {
      buf[b] += exp(epot[a * k + b] + npot[a]);
      bsq[a * k + b] += npot[a];
    }
    sum += buf[b];
  }
  for (int b = 0; b < k; b++) m12[b] = buf[b] / sum;
  sum = 0.0;
  for (int a = 0; a < k; a++) { /* message 2 -> 1, Tree.cpp:212-229 */
    buf[a] = 0.0;
    for (int b = 0; b < k; b++) {
      buf[a] += exp(epot[a * k + b] + npot[k + b]);
      bsq[a * k + b] += npot[k + b];
    }
    sum += buf[a];
  }
  for (int a = 0; a < k; a++) m21[a] = buf[a] / sum;
  sum = 0.0; /* node marginal 1, Tree.cpp:236-245 */
  for (int i = 0; i < k; i++) { buf[i] = exp(npot[i]) * m21[i]; sum += buf[i]; }
  for (int i = 0; i < k; i++) { mu[i] = buf[i] / sum; acc_l[i] += mu[i]; }
  sum = 0.0; /* node marginal 2, Tree.cpp:246-254 */
  for (int i = 0; i < k; i++) { buf[i] = exp(npot[i + k]) * m12[i]; sum += buf[i]; }
  for (int i = 0; i < k; i++) { mu[i + k] = buf[i] / sum; acc_r[i] += mu[i + k]; }
  for (int a = 0; a < k; a++) { /* edge marginal, Tree.cpp:256-273 */
    buf[a] = 0.0;
    for (int b = 0; b < k; b++) {
      bsq[a * k + b] = exp(bsq[a * k + b]);
      buf[a] += bsq[a * k + b];
    }
    for (int b = 0; b < k; b++) {
      if (buf[a] != 0.0) mue[a * k + b] = mu[a] * bsq[a * k + b] / buf[a];
      else mue[a * k + b] = 0.0;
    }
  }
  free(xnode);
}

void oracle_edge_compute(oracle_edge *o) {
  const int k = o->k, k2 = o->k2;
  memset(o->vertMu, 0, sizeof(float) * (size_t)o->V * k); /* Grid.cpp:144-148 */
  for (int e = 0; e < o->E; e++) {                         /* Grid.cpp:1400-1410 */
    int l = o->edgeList[2 * e], r = o->edgeList[2 * e + 1];
    oracle_edge_solve_one(k, o->mu_node + (size_t)e * 2 * k, o->pot_node + (size_t)e * 2 * k,
                          o->m_node + (size_t)e * 2 * k, o->lambda_node + (size_t)e * 2 * k,
                          o->mu_edge + (size_t)e * k2, o->pot_edge + (size_t)e * k2,
                          o->penalty, o->penaltyPrime, o->rho,
                          o->vertMu + (size_t)l * k, o->vertMu + (size_t)r * k);
  }
}

void oracle_edge_update(oracle_edge *o) {
  const int k = o->k;
  edge_average(o); /* Grid.cpp:163-181, no neighbours on one rank */
  for (int i = 0; i < 2 * o->E; i++) /* Grid.cpp:188-192 */
    memcpy(o->m_node + (size_t)(i / 2) * 2 * k + (size_t)(i % 2) * k,
           o->vertMu + (size_t)o->edgeList[i] * k, sizeof(float) * (size_t)k);
}

double oracle_edge_objective(const oracle_edge *o) {
  double obj = 0.0; /* Grid.cpp:1665-1680 */
  for (int e = 0; e < o->E; e++) {
    for (int j = 0; j < 2 * o->k; j++)
      obj += o->pot_node[(size_t)e * 2 * o->k + j] * o->mu_node[(size_t)e * 2 * o->k + j];
    for (int j = 0; j < o->k2; j++)
      obj += o->pot_edge[(size_t)e * o->k2 + j] * o->mu_edge[(size_t)e * o->k2 + j];
  }
  return obj * o->rho;
}

const int *oracle_edge_vertcnt(const oracle_edge *o) { return o->vertCnt; }
float *oracle_edge_mu_node(oracle_edge *o) { return o->mu_node; }
float *oracle_edge_pot_node(oracle_edge *o) { return o->pot_node; }
float *oracle_edge_m_node(oracle_edge *o) { return o->m_node; }
float *oracle_edge_lambda_node(oracle_edge *o) { return o->lambda_node; }
float *oracle_edge_mu_edge(oracle_edge *o) { return o->mu_edge; }
float *oracle_edge_pot_edge(oracle_edge *o) { return o->pot_edge; }
float *oracle_edge_vertmu(oracle_edge *o) { return o->vertMu; }

/* ------------------------------------------------------------------ */
/* edge solver on P ranks (EDGE partition), simulated in one process      */
/* ------------------------------------------------------------------ */
/*
 * What the reference does under mpiexec -n P: rank r owns a contiguous slice
 * of the edge list (Grid.cpp:446-454), keeps a partial consensus sum for every
 * vertex its edges touch (Grid.cpp:815-856), and per iteration sends every
 * neighbouring rank its partial sums of the vertices they share
 * (Grid.cpp:1639-1663), adds what it receives in ascending order of the
 * neighbours' ranks and divides by the global copy count (Grid.cpp:163-181).
 * The per-edge state lives in one oracle_edge; only the consensus arithmetic
 * depends on the partition.
 */
struct oracle_edge_ranks {
  oracle_edge *g;
  int P;
  int *eoff;     /* P+1: rank r owns edges [eoff[r], eoff[r+1]) */
  int *nv;       /* distinct vertices per rank */
  int **verts;   /* vSelfVec: sorted distinct global ids */
  int **cnt;     /* vertCnt: local copies, plus the neighbours' after init */
  float **vmu;   /* vertMuArr [nv][k] */
  int *nn;       /* neighbours per rank */
  int **nb;      /* their ranks, ascending (Grid.cpp:237-252) */
  int **ns;      /* shared vertices per neighbour */
  int ***sh;     /* outNeighborMapEdge: shared global ids, ascending */
  float ***recv; /* recvBuffer [neighbour][ns*k] */
};

static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return x < y ? -1 : x > y;
}

static int lvid(const oracle_edge_ranks *o, int r, int gv) { /* vSelfMapG2L */
  int lo = 0, hi = o->nv[r] - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    if (o->verts[r][mid] == gv) return mid;
    if (o->verts[r][mid] < gv) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

oracle_edge_ranks *oracle_edge_ranks_create(int V, int E, int k, const int *edgeList,
                                            const float *vertPot, const float *edgePot,
                                            float penalty, float penaltyPrime, float rho, int P) {
  oracle_edge_ranks *o = (oracle_edge_ranks *)calloc(1, sizeof(*o));
  o->g = oracle_edge_create(V, E, k, edgeList, vertPot, edgePot, penalty, penaltyPrime, rho);
  o->P = P;
  o->eoff = (int *)calloc((size_t)P + 1, sizeof(int));
  for (int r = 0; r < P; r++) { /* Grid.cpp:446-454 */
    int n = E / P, div = E % P, off;
    if (r < div) { n += 1; off = n * r; } else { off = (n + 1) * div + n * (r - div); }
    o->eoff[r] = off;
    o->eoff[r + 1] = off + n;
  }
  o->nv = (int *)calloc((size_t)P, sizeof(int));
  o->verts = (int **)calloc((size_t)P, sizeof(int *));
  o->cnt = (int **)calloc((size_t)P, sizeof(int *));
  o->vmu = (float **)calloc((size_t)P, sizeof(float *));
  for (int r = 0; r < P; r++) { /* GetVerts, Grid.cpp:815-856 */
    int n2 = 2 * (o->eoff[r + 1] - o->eoff[r]);
    int *v = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    memcpy(v, edgeList + 2 * (size_t)o->eoff[r], sizeof(int) * (size_t)n2);
    qsort(v, (size_t)n2, sizeof(int), cmp_int);
    o->verts[r] = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    o->cnt[r] = (int *)calloc((size_t)(n2 > 0 ? n2 : 1), sizeof(int));
    int nv = 0;
    for (int i = 0; i < n2; i++) {
      if (nv > 0 && o->verts[r][nv - 1] == v[i]) o->cnt[r][nv - 1]++;
      else { o->verts[r][nv] = v[i]; o->cnt[r][nv] = 1; nv++; }
    }
    o->nv[r] = nv;
    o->vmu[r] = falloc((long long)nv * k);
    free(v);
  }
  /* InitNeighborsByEdge, Grid.cpp:198-331 */
  o->nn = (int *)calloc((size_t)P, sizeof(int));
  o->nb = (int **)calloc((size_t)P, sizeof(int *));
  o->ns = (int **)calloc((size_t)P, sizeof(int *));
  o->sh = (int ***)calloc((size_t)P, sizeof(int **));
  o->recv = (float ***)calloc((size_t)P, sizeof(float **));
  for (int r = 0; r < P; r++) {
    o->nb[r] = (int *)calloc((size_t)P, sizeof(int));
    o->ns[r] = (int *)calloc((size_t)P, sizeof(int));
    o->sh[r] = (int **)calloc((size_t)P, sizeof(int *));
    o->recv[r] = (float **)calloc((size_t)P, sizeof(float *));
    if (o->nv[r] == 0) continue;
    const int lo = o->verts[r][0], hi = o->verts[r][o->nv[r] - 1];
    for (int i = 0; i < P; i++) {
      if (i == r || o->nv[i] == 0) continue;
      /* neighbours are the ranks whose vertex range overlaps mine (Grid.cpp:240-251) */
      if (o->verts[i][0] > hi || lo > o->verts[i][o->nv[i] - 1]) continue;
      const int q = o->nn[r]++;
      o->nb[r][q] = i;
      /* shared vertices: the runs of both sides intersected, ascending (Grid.cpp:264-279) */
      o->sh[r][q] = (int *)malloc(sizeof(int) * (size_t)o->nv[r]);
      int n = 0, a = 0, b = 0;
      while (a < o->nv[r] && b < o->nv[i]) {
        if (o->verts[r][a] == o->verts[i][b]) { o->sh[r][q][n++] = o->verts[r][a]; a++; b++; }
        else if (o->verts[r][a] < o->verts[i][b]) a++; else b++;
      }
      o->ns[r][q] = n;
      o->recv[r][q] = falloc((long long)n * k);
    }
  }
  /* global copy counts: mine plus what every neighbour reports (Grid.cpp:288-319);
   * the reports are the neighbours' LOCAL counts, so take them before any is raised */
  int **local = (int **)calloc((size_t)P, sizeof(int *));
  for (int r = 0; r < P; r++) {
    local[r] = (int *)malloc(sizeof(int) * (size_t)(o->nv[r] > 0 ? o->nv[r] : 1));
    memcpy(local[r], o->cnt[r], sizeof(int) * (size_t)o->nv[r]);
  }
  for (int r = 0; r < P; r++)
    for (int q = 0; q < o->nn[r]; q++)
      for (int j = 0; j < o->ns[r][q]; j++) {
        const int gv = o->sh[r][q][j];
        o->cnt[r][lvid(o, r, gv)] += local[o->nb[r][q]][lvid(o, o->nb[r][q], gv)];
      }
  for (int r = 0; r < P; r++) free(local[r]);
  free(local);
  return o;
}

void oracle_edge_ranks_destroy(oracle_edge_ranks *o) {
  if (!o) return;
  for (int r = 0; r < o->P; r++) {
    for (int q = 0; q < o->nn[r]; q++) { free(o->sh[r][q]); free(o->recv[r][q]); }
    free(o->verts[r]); free(o->cnt[r]); free(o->vmu[r]); free(o->nb[r]); free(o->ns[r]);
    free(o->sh[r]); free(o->recv[r]);
  }
  free(o->eoff); free(o->nv); free(o->verts); free(o->cnt); free(o->vmu); free(o->nn);
  free(o->nb); free(o->ns); free(o->sh); free(o->recv);
  oracle_edge_destroy(o->g);
  free(o);
}

/* CommunicateByEdge (Grid.cpp:1639-1663) for all ranks at once, then
 * UpdateMuArrAfterComm (Grid.cpp:163-181). */
static void ranks_exchange_and_average(oracle_edge_ranks *o) {
  const int k = o->g->k;
  for (int r = 0; r < o->P; r++) /* every message leaves before any sum changes */
    for (int q = 0; q < o->nn[r]; q++) {
      const int i = o->nb[r][q];
      for (int j = 0; j < o->ns[r][q]; j++)
        memcpy(o->recv[r][q] + (size_t)j * k, o->vmu[i] + (size_t)lvid(o, i, o->sh[r][q][j]) * k,
               sizeof(float) * (size_t)k);
    }
  for (int r = 0; r < o->P; r++) {
    for (int q = 0; q < o->nn[r]; q++)
      for (int j = 0; j < o->ns[r][q]; j++) {
        float *dst = o->vmu[r] + (size_t)lvid(o, r, o->sh[r][q][j]) * k;
        for (int l = 0; l < k; l++) dst[l] += o->recv[r][q][(size_t)j * k + l];
      }
    for (int v = 0; v < o->nv[r]; v++)
      for (int l = 0; l < k; l++) o->vmu[r][(size_t)v * k + l] /= o->cnt[r][v];
  }
}

void oracle_edge_ranks_init(oracle_edge_ranks *o) {
  oracle_edge *g = o->g;
  const int k = g->k, k2 = g->k2;
  /* Grid.cpp:1478-1487.  The reference indexes the unaries it read (a
   * contiguous range of vertices) by local id (Grid.cpp:510-523, 1482), which
   * is the vertex's own row as long as a rank's vertex set has no gaps; the
   * vertex's own row is used here. */
  for (int r = 0; r < o->P; r++) {
    memset(o->vmu[r], 0, sizeof(float) * (size_t)o->nv[r] * k);
    for (int i = 2 * o->eoff[r]; i < 2 * o->eoff[r + 1]; i++) {
      const int gv = g->edgeList[i], lv = lvid(o, r, gv);
      for (int j = 0; j < k; j++) o->vmu[r][(size_t)lv * k + j] += g->vertPot[(size_t)gv * k + j];
    }
  }
  ranks_exchange_and_average(o);
  const float inv_k = 1.0 / (float)k;
  const float inv_k2 = 1.0 / (float)k2;
  for (int r = 0; r < o->P; r++)
    for (int e = o->eoff[r]; e < o->eoff[r + 1]; e++) { /* InitTree, Grid.cpp:944-979 */
      const int l = lvid(o, r, g->edgeList[2 * e]), rr = lvid(o, r, g->edgeList[2 * e + 1]);
      float *pn = g->pot_node + (size_t)e * 2 * k;
      for (int j = 0; j < k; j++) {
        pn[j] = o->vmu[r][(size_t)l * k + j] / (g->rho * o->cnt[r][l]);
        pn[k + j] = o->vmu[r][(size_t)rr * k + j] / (g->rho * o->cnt[r][rr]);
      }
      for (int j = 0; j < 2 * k; j++) {
        g->mu_node[(size_t)e * 2 * k + j] = inv_k;
        g->m_node[(size_t)e * 2 * k + j] = inv_k;
        g->lambda_node[(size_t)e * 2 * k + j] = 0.0f;
      }
      for (int j = 0; j < k2; j++) {
        g->pot_edge[(size_t)e * k2 + j] = g->edgePot[(size_t)e * k2 + j] / g->rho;
        g->mu_edge[(size_t)e * k2 + j] = inv_k2;
      }
    }
}

void oracle_edge_ranks_compute(oracle_edge_ranks *o) {
  oracle_edge *g = o->g;
  const int k = g->k, k2 = g->k2;
  for (int r = 0; r < o->P; r++) {
    memset(o->vmu[r], 0, sizeof(float) * (size_t)o->nv[r] * k); /* Grid.cpp:144-148 */
    for (int e = o->eoff[r]; e < o->eoff[r + 1]; e++) {         /* Grid.cpp:1400-1410 */
      const int l = lvid(o, r, g->edgeList[2 * e]), rr = lvid(o, r, g->edgeList[2 * e + 1]);
      oracle_edge_solve_one(k, g->mu_node + (size_t)e * 2 * k, g->pot_node + (size_t)e * 2 * k,
                            g->m_node + (size_t)e * 2 * k, g->lambda_node + (size_t)e * 2 * k,
                            g->mu_edge + (size_t)e * k2, g->pot_edge + (size_t)e * k2,
                            g->penalty, g->penaltyPrime, g->rho,
                            o->vmu[r] + (size_t)l * k, o->vmu[r] + (size_t)rr * k);
    }
  }
}

void oracle_edge_ranks_update(oracle_edge_ranks *o) {
  oracle_edge *g = o->g;
  const int k = g->k;
  ranks_exchange_and_average(o); /* Communicate + UpdateMuArrAfterComm */
  for (int r = 0; r < o->P; r++)
    for (int i = 2 * o->eoff[r]; i < 2 * o->eoff[r + 1]; i++) /* Grid.cpp:188-192 */
      memcpy(g->m_node + (size_t)(i / 2) * 2 * k + (size_t)(i % 2) * k,
             o->vmu[r] + (size_t)lvid(o, r, g->edgeList[i]) * k, sizeof(float) * (size_t)k);
}

double oracle_edge_ranks_objective(const oracle_edge_ranks *o) {
  const oracle_edge *g = o->g;
  double total = 0.0; /* MPI_Reduce SUM of the ranks' values, admm.cpp:94-95 */
  for (int r = 0; r < o->P; r++) {
    double obj = 0.0; /* Grid.cpp:1665-1680 */
    for (int e = o->eoff[r]; e < o->eoff[r + 1]; e++) {
      for (int j = 0; j < 2 * g->k; j++)
        obj += g->pot_node[(size_t)e * 2 * g->k + j] * g->mu_node[(size_t)e * 2 * g->k + j];
      for (int j = 0; j < g->k2; j++)
        obj += g->pot_edge[(size_t)e * g->k2 + j] * g->mu_edge[(size_t)e * g->k2 + j];
    }
    total += obj * g->rho;
  }
  return total;
}

oracle_edge *oracle_edge_ranks_state(oracle_edge_ranks *o) { return o->g; }
int oracle_edge_ranks_nverts(const oracle_edge_ranks *o, int r) { return o->nv[r]; }
const int *oracle_edge_ranks_verts(const oracle_edge_ranks *o, int r) { return o->verts[r]; }
const int *oracle_edge_ranks_vertcnt(const oracle_edge_ranks *o, int r) { return o->cnt[r]; }
float *oracle_edge_ranks_vertmu(oracle_edge_ranks *o, int r) { return o->vmu[r]; }

/* ------------------------------------------------------------------ */
/* chain solver                                                         */
/* ------------------------------------------------------------------ */
/*
 * Trees are the R row chains and the C column chains, so every vertex has
 * exactly two copies (vertCnt == 2).  The shell is the edge solver's:
 *   split     theta_tau = theta_v / (rho * 2), W_tau = W / rho   (Grid.cpp:952-977)
 *   dual      lambda += penalty * (mu - m)                        (Tree.cpp:70)
 *   adjust    t = rho*(theta_tau + lambda) - penalty*(m - mu)     (Tree.cpp:85-86)
 *             pairwise = rho * W_tau                              (Tree.cpp:99)
 *   consensus m_v = (mu_row + mu_col) / 2                         (Grid.cpp:176-180)
 * and the subproblem is min-sum on a chain of T nodes, swept from both ends
 * towards the middle (the two halves are independent until they meet, which
 * is what lets two wavefronts share a chain).  With L = (T + 1) / 2:
 *   left   A_0 = t_0,          A_i[b] = t_i[b] + min_a (A_{i-1}[a] + W[a][b]),  i < L
 *   right  B_{T-1} = t_{T-1},  B_j[a] = t_j[a] + min_b (B_{j+1}[b] + W[a][b]),  j >= L
 *   join   J[a] = A_{L-1}[a] + min_b (B_L[b] + W[a][b])
 * Every minimum remembers the lowest index attaining it.  x_{L-1} = lowest
 * argmin of J, x_L = its remembered b; the left half is traced back from
 * x_{L-1}, the right half forward from x_L.  T = 1 has no right half and
 * x_0 = lowest argmin of A_0.  Each candidate is one rounded add and each
 * message one more rounded add, in exactly this association.
 * mu is the one-hot vector of the chain's labels (before the first solve it
 * is the uniform 1/k of Tree.cpp:24-33, kept as the label -1).
 */
struct oracle_chain {
  int R, C, k;
  float penalty, rho;
  float *theta;    /* V*k, per-copy share */
  float *Wt;       /* k*k, rho * (W / rho) */
  float *Wsplit;   /* k*k, W / rho */
  float *lam_row, *lam_col, *m; /* V*k */
  int *xrow, *xcol;             /* V */
  float inv_k;
};

oracle_chain *oracle_chain_create(int R, int C, int k, const float *unary,
                                  const float *W, float penalty, float rho) {
  oracle_chain *o = (oracle_chain *)calloc(1, sizeof(*o));
  const long long V = (long long)R * C;
  o->R = R; o->C = C; o->k = k; o->penalty = penalty; o->rho = rho;
  o->theta = falloc(V * k);
  memcpy(o->theta, unary, sizeof(float) * (size_t)(V * k));
  o->Wt = falloc((long long)k * k);
  o->Wsplit = falloc((long long)k * k);
  memcpy(o->Wsplit, W, sizeof(float) * (size_t)k * k);
  o->lam_row = falloc(V * k);
  o->lam_col = falloc(V * k);
  o->m = falloc(V * k);
  o->xrow = (int *)malloc(sizeof(int) * (size_t)(V > 0 ? V : 1));
  o->xcol = (int *)malloc(sizeof(int) * (size_t)(V > 0 ? V : 1));
  return o;
}

void oracle_chain_destroy(oracle_chain *o) {
  if (!o) return;
  free(o->theta); free(o->Wt); free(o->Wsplit); free(o->lam_row); free(o->lam_col);
  free(o->m); free(o->xrow); free(o->xcol); free(o);
}

void oracle_chain_init(oracle_chain *o) {
  const long long V = (long long)o->R * o->C;
  const int k = o->k;
  const int cnt = 2;
  for (long long i = 0; i < V * k; i++) {
    /* Grid.cpp:1478-1487 with two copies, then Grid.cpp:953-956 */
    float s = 0.0f;
    s += o->theta[i];
    s += o->theta[i];
    s /= cnt;
    o->theta[i] = s / (o->rho * cnt);
    o->lam_row[i] = 0.0f;
    o->lam_col[i] = 0.0f;
  }
  o->inv_k = 1.0 / (float)k;
  for (long long i = 0; i < V * k; i++) o->m[i] = o->inv_k;
  for (long long i = 0; i < V; i++) { o->xrow[i] = -1; o->xcol[i] = -1; }
  for (int i = 0; i < k * k; i++) {
    o->Wsplit[i] = o->Wsplit[i] / o->rho; /* Grid.cpp:975-976 */
    o->Wt[i] = o->rho * o->Wsplit[i];     /* Tree.cpp:99 */
  }
}

static inline float chain_mu(int x, int l, float inv_k) {
  return x < 0 ? inv_k : (x == l ? 1.0f : 0.0f);
}

static int g_threads = 1;
static int g_threads_fwd(void) { return g_threads; }

/* min over src of (M[src] + W[...]) with the lowest src attaining it.
 * transposed = 0: W[src][tgt] (left half); 1: W[tgt][src] (right half, join). */
static inline float minplus(const float *M, const float *W, int k, int tgt, int transposed, int *arg) {
  float best = M[0] + (transposed ? W[tgt * k] : W[tgt]);
  int at = 0;
  for (int s = 1; s < k; s++) {
    float c = M[s] + (transposed ? W[tgt * k + s] : W[s * k + tgt]);
    if (c < best) { best = c; at = s; }
  }
  *arg = at;
  return best;
}

/* Min-sum over a chain whose adjusted unaries are th[T][k], pairwise W[a][b]
 * (a = label of the earlier node).  M: 3k floats of scratch.  ptr (T*k bytes)
 * and x (T labels, node t at x[t * xstep]) may both be NULL when only the
 * minimum is wanted.  ptr[t][.] holds, for t < L, the back-pointers of node t
 * and, for t >= L, for every label of node t-1 the best label of node t.
 * Returns min_a J[a]. */
static float chain_minsum(const float *th, const float *W, int k, int T, float *M,
                          unsigned char *ptr, int *x, long long xstep) {
  const int L = (T + 1) / 2;
  float *A = M, *B = M + k, *N = M + 2 * k;
  int arg;
  for (int b = 0; b < k; b++) A[b] = th[b];
  for (int t = 1; t < L; t++) {
    for (int b = 0; b < k; b++) {
      float best = minplus(A, W, k, b, 0, &arg);
      N[b] = th[(size_t)t * k + b] + best;
      if (ptr) ptr[(size_t)t * k + b] = (unsigned char)arg;
    }
    memcpy(A, N, sizeof(float) * (size_t)k);
  }
  int xl = 0, xr = 0;
  float value;
  if (T - L > 0) {
    for (int a = 0; a < k; a++) B[a] = th[(size_t)(T - 1) * k + a];
    for (int t = T - 2; t >= L; t--) {
      for (int a = 0; a < k; a++) {
        float best = minplus(B, W, k, a, 1, &arg);
        N[a] = th[(size_t)t * k + a] + best;
        if (ptr) ptr[(size_t)(t + 1) * k + a] = (unsigned char)arg;
      }
      memcpy(B, N, sizeof(float) * (size_t)k);
    }
    /* join: N[a] = A[a] + min_b (B[b] + W[a][b]) */
    int g0 = 0;
    for (int a = 0; a < k; a++) {
      float best = minplus(B, W, k, a, 1, &arg);
      N[a] = A[a] + best;
      if (ptr) ptr[(size_t)L * k + a] = (unsigned char)arg;
      if (a == 0 || N[a] < N[xl]) { xl = a; g0 = arg; }
    }
    xr = g0;
    value = N[xl];
  } else {
    for (int b = 1; b < k; b++) if (A[b] < A[xl]) xl = b;
    value = A[xl];
  }
  if (x && ptr) {
    int cur = xl;
    for (int t = L - 1; t >= 0; t--) {
      x[(long long)t * xstep] = cur;
      if (t > 0) cur = ptr[(size_t)t * k + cur];
    }
    cur = xr;
    for (int t = L; t < T; t++) {
      x[(long long)t * xstep] = cur;
      if (t + 1 < T) cur = ptr[(size_t)(t + 1) * k + cur];
    }
  }
  return value;
}

/* The subproblem alone, for tests that check it against an independent
 * solver: n chains of T nodes with adjusted unaries th[n][T][k] and the table
 * W[a][b]; x[n][T] receives the labels, val[n] the chain minima. */
void oracle_chain_minsum_batch(const float *th, const float *W, int k, int T, int n, int *x, float *val) {
#ifdef _OPENMP
#pragma omp parallel num_threads(g_threads_fwd())
#endif
  {
    float *M = falloc(3LL * k);
    unsigned char *bp = (unsigned char *)malloc((size_t)T * k + 1);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int c = 0; c < n; c++)
      val[c] = chain_minsum(th + (size_t)c * T * k, W, k, T, M, bp, x + (size_t)c * T, 1);
    free(M); free(bp);
  }
}

/* One chain of T nodes; node t is first + t*step. */
static void chain_solve(oracle_chain *o, float *lam, int *x, long long first,
                        long long step, int T, float *th, float *M, unsigned char *bp) {
  const int k = o->k;
  for (int t = 0; t < T; t++) {
    long long n = first + (long long)t * step;
    for (int l = 0; l < k; l++) {
      float mu = chain_mu(x[n], l, o->inv_k);
      float mv = o->m[n * k + l];
      float la = lam[n * k + l];
      la += o->penalty * (mu - mv);
      lam[n * k + l] = la;
      th[(size_t)t * k + l] = o->rho * (o->theta[n * k + l] + la) - o->penalty * (mv - mu);
    }
  }
  chain_minsum(th, o->Wt, k, T, M, bp, x + first, step);
}

void oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }

/* Chains are independent inside one Compute, so the baseline may run one chain
 * per thread; results do not depend on the thread count. */
void oracle_chain_compute_range(oracle_chain *o, int r0, int r1, int c0, int c1) {
  const int R = o->R, C = o->C, k = o->k;
  const int T = R > C ? R : C;
#ifdef _OPENMP
#pragma omp parallel num_threads(g_threads)
#endif
  {
    float *th = falloc((long long)T * k);
    float *M = falloc(3LL * k);
    unsigned char *bp = (unsigned char *)malloc((size_t)T * k + 1);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4) nowait
#endif
    for (int r = r0; r < r1; r++) chain_solve(o, o->lam_row, o->xrow, (long long)r * C, 1, C, th, M, bp);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int c = c0; c < c1; c++) chain_solve(o, o->lam_col, o->xcol, c, C, R, th, M, bp);
    free(th); free(M); free(bp);
  }
}

void oracle_chain_compute(oracle_chain *o) { oracle_chain_compute_range(o, 0, o->R, 0, o->C); }

void oracle_chain_update_range(oracle_chain *o, int r0, int r1, int c0, int c1) {
  const int k = o->k;
  for (int r = r0; r < r1; r++)
    for (int c = c0; c < c1; c++) {
      long long n = (long long)r * o->C + c;
      for (int l = 0; l < k; l++) {
        float s = 0.0f;
        s += chain_mu(o->xrow[n], l, o->inv_k);
        s += chain_mu(o->xcol[n], l, o->inv_k);
        s /= 2; /* Grid.cpp:178 */
        o->m[n * k + l] = s;
      }
    }
}

void oracle_chain_update(oracle_chain *o) { oracle_chain_update_range(o, 0, o->R, 0, o->C); }

static inline int chain_decode(int xr, int xc) {
  /* argmax of m_v with the lowest index winning ties */
  if (xr < 0 || xc < 0) return 0;
  return xr < xc ? xr : xc;
}

void oracle_chain_labels(const oracle_chain *o, int *labels) {
  const long long V = (long long)o->R * o->C;
  for (long long n = 0; n < V; n++) labels[n] = chain_decode(o->xrow[n], o->xcol[n]);
}

void oracle_chain_objective(const oracle_chain *o, double *primal, double *dual, double *linear) {
  const int R = o->R, C = o->C, k = o->k;
  if (primal) {
    double p = 0.0;
    for (int r = 0; r < R; r++)
      for (int c = 0; c < C; c++) {
        long long n = (long long)r * C + c;
        int x = chain_decode(o->xrow[n], o->xcol[n]);
        p += (double)o->theta[n * k + x];
        p += (double)o->theta[n * k + x];
        if (c + 1 < C) p += (double)o->Wsplit[x * k + chain_decode(o->xrow[n + 1], o->xcol[n + 1])];
        if (r + 1 < R) p += (double)o->Wsplit[x * k + chain_decode(o->xrow[n + C], o->xcol[n + C])];
      }
    *primal = p * o->rho;
  }
  if (linear) {
    double p = 0.0;
    for (int r = 0; r < R; r++)
      for (int c = 0; c < C; c++) {
        long long n = (long long)r * C + c;
        int xr = o->xrow[n] < 0 ? 0 : o->xrow[n], xc = o->xcol[n] < 0 ? 0 : o->xcol[n];
        p += (double)o->theta[n * k + xr];
        p += (double)o->theta[n * k + xc];
        if (c + 1 < C) p += (double)o->Wsplit[xr * k + (o->xrow[n + 1] < 0 ? 0 : o->xrow[n + 1])];
        if (r + 1 < R) p += (double)o->Wsplit[xc * k + (o->xcol[n + C] < 0 ? 0 : o->xcol[n + C])];
      }
    *linear = p * o->rho;
  }
  if (dual) {
    /* sum over chains of the chain minimum under theta_tau + lambda_tau */
    double d = 0.0;
    const int Tmax = R > C ? R : C;
    float *M = falloc(3LL * k);
    float *th = falloc((long long)Tmax * k);
    for (int pass = 0; pass < 2; pass++) {
      const float *lam = pass == 0 ? o->lam_row : o->lam_col;
      int nch = pass == 0 ? R : C, T = pass == 0 ? C : R;
      for (int ch = 0; ch < nch; ch++) {
        long long first = pass == 0 ? (long long)ch * C : ch, step = pass == 0 ? 1 : C;
        for (int t = 0; t < T; t++) {
          long long n = first + (long long)t * step;
          for (int b = 0; b < k; b++) th[(size_t)t * k + b] = o->theta[n * k + b] + lam[n * k + b];
        }
        d += (double)chain_minsum(th, o->Wsplit, k, T, M, NULL, NULL, 0);
      }
    }
    free(M); free(th);
    *dual = d * o->rho;
  }
}

int *oracle_chain_xrow(oracle_chain *o) { return o->xrow; }
int *oracle_chain_xcol(oracle_chain *o) { return o->xcol; }
float *oracle_chain_consensus(oracle_chain *o) { return o->m; }
float *oracle_chain_lambda_row(oracle_chain *o) { return o->lam_row; }
float *oracle_chain_lambda_col(oracle_chain *o) { return o->lam_col; }
float *oracle_chain_theta(oracle_chain *o) { return o->theta; }
