/*
 * admm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the ADMM hot path of xiaocai00/ADMM.  It exists so
 * that the HIP kernels can be checked against it; nothing that ships may call
 * it.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py load the library built from this file.
 *
 * Two subproblem solvers sit behind the same ADMM shell:
 *
 *   edge  : the reference's solver.  Every tree is one edge; the subproblem is
 *           an entropy-proximal sum-product (Tree.cpp:112-274).  Pinned: the
 *           arithmetic is bit-compared against the reference's own Tree.cpp
 *           compiled from /root/reference (oracle/_ref), and the whole loop
 *           against the totalObj trace the unmodified reference printed.
 *
 *   chain : row/column chain min-sum (Viterbi).  The reference has no chain
 *           code, so this solver is "parity unpinned" by the reference; its
 *           shell (dual update, potential adjust, consensus, split of the
 *           unaries) is the edge solver's, which is pinned.
 */
#ifndef ADMM_ORACLE_H_
#define ADMM_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- synthetic inputs ---------- */

/* 32-bit LCG: s = s*1664525 + 1013904223; u = (float)(s >> 8) / 16777216.0f.
 * Fills out[0..n) and returns the advanced state. */
unsigned oracle_lcg_fill(unsigned state, long long n, float *out);

/* Grid edge list in the order of Grid::Generate (Grid.cpp:581-615): for each
 * row the C-1 horizontal edges, then (except after the last row) the C
 * vertical edges.  edgeList has 2*E ints.  Returns E. */
int oracle_grid_edges(int R, int C, int *edgeList);

/* ---------- edge solver (reference algorithm) ---------- */

typedef struct oracle_edge oracle_edge;

/* vertPot: V*k unary costs (the values the reference holds AFTER the loader's
 * sign flip, Grid.cpp:547-553).  edgePot: E*k*k per-edge tables, likewise
 * after the flip (Grid.cpp:554-560).  Arbitrary edge lists are accepted. */
oracle_edge *oracle_edge_create(int V, int E, int k, const int *edgeList,
                                const float *vertPot, const float *edgePot,
                                float penalty, float penaltyPrime, float rho);
void oracle_edge_destroy(oracle_edge *o);
void oracle_edge_init(oracle_edge *o);        /* Grid.cpp:1474-1491, 934-980 */
void oracle_edge_compute(oracle_edge *o);     /* Grid.cpp:1394-1412 */
void oracle_edge_update(oracle_edge *o);      /* Grid.cpp:186-193, 163-181 */
double oracle_edge_objective(const oracle_edge *o); /* Grid.cpp:1665-1680 */

/* One edge subproblem on caller-owned arrays (Tree.cpp:64-274): updates
 * lambda, mu_node, mu_edge in place and adds the new node marginals into
 * acc_l[k], acc_r[k]. */
void oracle_edge_solve_one(int k, float *mu_node, const float *pot_node,
                           const float *m_node, float *lambda_node,
                           float *mu_edge, const float *pot_edge,
                           float penalty, float penaltyPrime, float rho,
                           float *acc_l, float *acc_r);

/* state access (pointers into the oracle, valid until destroy) */
const int *oracle_edge_vertcnt(const oracle_edge *o);      /* V       */
float *oracle_edge_mu_node(oracle_edge *o);          /* E*2k    */
float *oracle_edge_pot_node(oracle_edge *o);         /* E*2k    */
float *oracle_edge_m_node(oracle_edge *o);           /* E*2k    */
float *oracle_edge_lambda_node(oracle_edge *o);      /* E*2k    */
float *oracle_edge_mu_edge(oracle_edge *o);          /* E*k*k   */
float *oracle_edge_pot_edge(oracle_edge *o);         /* E*k*k   */
float *oracle_edge_vertmu(oracle_edge *o);           /* V*k     */

/* The same solver as the reference runs it on P MPI ranks (EDGE partition:
 * Grid.cpp:445-454, 198-331, 1639-1663, 163-181), all ranks simulated in one
 * process.  The per-edge state is that of an oracle_edge; only the order in
 * which a vertex's copies are summed depends on P. */
typedef struct oracle_edge_ranks oracle_edge_ranks;
oracle_edge_ranks *oracle_edge_ranks_create(int V, int E, int k, const int *edgeList,
                                            const float *vertPot, const float *edgePot,
                                            float penalty, float penaltyPrime, float rho, int P);
void oracle_edge_ranks_destroy(oracle_edge_ranks *o);
void oracle_edge_ranks_init(oracle_edge_ranks *o);
void oracle_edge_ranks_compute(oracle_edge_ranks *o);
void oracle_edge_ranks_update(oracle_edge_ranks *o);   /* Communicate + Update */
double oracle_edge_ranks_objective(const oracle_edge_ranks *o);
oracle_edge *oracle_edge_ranks_state(oracle_edge_ranks *o);      /* per-edge arrays */
int oracle_edge_ranks_nverts(const oracle_edge_ranks *o, int r);
const int *oracle_edge_ranks_verts(const oracle_edge_ranks *o, int r);   /* sorted global ids */
const int *oracle_edge_ranks_vertcnt(const oracle_edge_ranks *o, int r); /* global copy counts */
float *oracle_edge_ranks_vertmu(oracle_edge_ranks *o, int r);            /* nverts*k */

/* Threads used by oracle_chain_compute* (one chain per thread; default 1). */
void oracle_set_threads(int n);

/* ---------- chain solver (parity unpinned by the reference) ---------- */

typedef struct oracle_chain oracle_chain;

/* unary: R*C*k costs, node-major [r][c][label].  W: k*k, W[a*k+b] is the cost
 * of label a on the earlier node (left / upper) and b on the later one. */
oracle_chain *oracle_chain_create(int R, int C, int k, const float *unary,
                                  const float *W, float penalty, float rho);
void oracle_chain_destroy(oracle_chain *o);
void oracle_chain_init(oracle_chain *o);
/* Solve the row chains [r0,r1) and the column chains [c0,c1). */
void oracle_chain_compute_range(oracle_chain *o, int r0, int r1, int c0, int c1);
void oracle_chain_compute(oracle_chain *o);
/* Consensus for rows [r0,r1) x cols [c0,c1) from the current label arrays. */
void oracle_chain_update_range(oracle_chain *o, int r0, int r1, int c0, int c1);
void oracle_chain_update(oracle_chain *o);
/* primal: energy of the decoded labels; dual: sum of chain minima under
 * theta_tau + lambda_tau; linear: energy of every chain's own labelling under
 * its share of the potentials (the analogue of Grid.cpp:1665-1680).  Any
 * pointer may be NULL. */
void oracle_chain_objective(const oracle_chain *o, double *primal, double *dual,
                            double *linear);
void oracle_chain_labels(const oracle_chain *o, int *labels);  /* R*C decoded */
/* The chain subproblem alone (two-ended min-sum, lowest index on ties): n
 * chains of T nodes, adjusted unaries th[n][T][k], table W[a][b]; labels to
 * x[n][T], chain minima to val[n]. */
void oracle_chain_minsum_batch(const float *th, const float *W, int k, int T, int n, int *x, float *val);
int *oracle_chain_xrow(oracle_chain *o);       /* R*C, -1 before first solve */
int *oracle_chain_xcol(oracle_chain *o);       /* R*C */
float *oracle_chain_consensus(oracle_chain *o);   /* R*C*k */
float *oracle_chain_lambda_row(oracle_chain *o);  /* R*C*k */
float *oracle_chain_lambda_col(oracle_chain *o);  /* R*C*k */
float *oracle_chain_theta(oracle_chain *o);       /* R*C*k (per-copy share) */

#ifdef __cplusplus
}
#endif
#endif
