/*
 * admm_hip.h -- C ABI of libadmm_hip.so, the MI355X implementation of the ADMM
 * MAP-inference hot path of xiaocai00/ADMM.
 *
 * The reference has no FFI layer: its driver (admm.cpp:41-101) calls methods
 * of class Grid, and Grid::Compute hands every tree to class Edge
 * (Grid.cpp:1400-1410, Tree.h:47-50).  The entry points below sit on that
 * seam.  Each one names the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ADMM_E_* code on
 *     failure; admm_last_error() gives the message (the reference has no
 *     error returns at all, Grid.cpp:26-30 only prints);
 *   - the caller owns every host buffer, the context owns all device memory;
 *   - one context per GPU; calls on one context are not thread-safe, calls on
 *     different contexts are independent;
 *   - compute / communicate / update are stream-ordered on the context's own
 *     HIP stream and return without waiting; admm_synchronize(), the
 *     objective calls and the download calls wait for the stream;
 *   - all arithmetic that the reference does in binary32 is done in binary32
 *     without fused multiply-add.
 */
#ifndef ADMM_HIP_H_
#define ADMM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMM_OK 0
#define ADMM_E_ARG (-1)     /* bad argument or unsupported size            */
#define ADMM_E_STATE (-2)   /* call out of order (e.g. compute before init) */
#define ADMM_E_HIP (-3)     /* a HIP runtime call failed                    */
#define ADMM_E_COMM (-4)    /* an RCCL call failed / communicator missing   */
#define ADMM_E_NOMEM (-5)

/* Subproblem solver. */
#define ADMM_SOLVER_EDGE_BETHE 0   /* one tree per edge, Tree.cpp:112-274      */
#define ADMM_SOLVER_CHAIN_MINSUM 1 /* row + column chains, min-sum (Viterbi)   */

/* What crosses GPUs between the row and the column chains (chain solver). */
#define ADMM_EXCHANGE_LABELS 0 /* int32 labels; consensus rebuilt on both sides */
#define ADMM_EXCHANGE_DENSE 1  /* the marginals as float k-vectors, the form the
                                * reference sends (Grid.cpp:1646-1656); k times
                                * the bytes, bit-identical results             */

typedef struct admm_ctx admm_ctx;

/* Problem and placement.  Replaces Grid::SetGlobalInfo (Grid.cpp:96-116),
 * Grid::SetPenalties (Grid.h:98), Grid::SetRhoEdge (Grid.h:111) and
 * Grid::SetPartitioner (Grid.h:110). */
typedef struct admm_desc {
  int32_t rows;          /* numGlobalRows                                     */
  int32_t cols;          /* numGlobalCols                                     */
  int32_t k;             /* numVertPotentials.  Edge solver: 2..1024 (the
                          * reference allocates for any k, Tree.cpp:15-22;
                          * beyond 64 labels an edge's table stays in HBM).
                          * Chain solver: 2..64, a label vector is held across
                          * the lanes of one wavefront.  admm_create refuses
                          * anything else with ADMM_E_ARG                     */
  int32_t batch;         /* independent grids of this shape (>= 1)            */
  int32_t solver;        /* ADMM_SOLVER_*                                     */
  float penalty;         /* admm.cpp:44, first argument                       */
  float penalty_prime;   /* admm.cpp:44, second argument (edge solver only)   */
  float rho;             /* rhoEdge, always 1.0 in the reference              */
  int32_t device;        /* HIP device ordinal                                */
  int32_t rank;          /* this context's strip, 0 <= rank < nranks          */
  int32_t nranks;        /* strips (ranks); > 1 needs batch == 1.  Chain
                          * solver: row / column strips.  Edge solver: the
                          * reference's EDGE partition, contiguous slices of
                          * the edge list (Grid.cpp:445-454)                 */
  int32_t exchange;      /* ADMM_EXCHANGE_*                                   */
  /* Edge solver on an arbitrary graph (the reference accepts any edge list,
   * Grid.cpp:476-494).  NULL: the grid's own edge list in the order of
   * Grid::Generate (Grid.cpp:581-615). */
  const int32_t *edge_list;  /* 2 * num_edges vertex ids                      */
  int32_t num_vertices;
  int32_t num_edges;
} admm_desc;

/* Grid::Grid + Grid::Load's allocations (Grid.cpp:456-460, 1607-1613). */
int admm_create(const admm_desc *desc, admm_ctx **out);
/* Grid::~Grid (Grid.cpp:121-142). */
int admm_destroy(admm_ctx *ctx);
/* Message of the last failure on ctx (or, with NULL, of the last failed
 * admm_create on this thread). */
const char *admm_last_error(const admm_ctx *ctx);

/* Unary costs, node-major [batch][rows*cols][k] (graph mode: [V][k]): the
 * values Grid holds after the loader's sign flip (Grid.cpp:510-523, 547-553).
 * Always the whole problem; a strip context keeps its own share. */
int admm_upload_unaries(admm_ctx *ctx, const float *unary);
/* The same, a share at a time, for a caller that stages only what this strip
 * keeps: in the reference every rank reads its own hyperslab of the file and
 * nothing else (Grid.cpp:446-454, 510-523).  Chain solver, batch == 1.
 * _rows: whole grid rows [nrows][cols][k], global rows row0 .. row0+nrows-1,
 * inside this context's row strip (admm_strip_info).  _cols: this context's
 * column strip of global rows row0 .. row0+nrows-1, [nrows][strip cols][k]; any
 * rows of the grid.  With one strip the rows are the whole problem and _cols
 * copies nothing.  admm_init needs every row of both shares; the first share
 * after an admm_init starts a new upload (everything has to arrive again).
 * Each call returns when the caller's buffer may be reused. */
int admm_upload_unaries_rows(admm_ctx *ctx, int row0, int nrows, const float *rows);
int admm_upload_unaries_cols(admm_ctx *ctx, int row0, int nrows, const float *cols);
/* Pairwise costs after the sign flip (Grid.cpp:529-543, 554-560).
 * per_edge = 1: [num_edges][k*k] in edge-list order (edge solver);
 * per_edge = 0: one k*k table W[a*k+b], a = label of the edge's first vertex,
 * shared by every edge (required by the chain solver). */
int admm_upload_pairwise(admm_ctx *ctx, const float *table, int per_edge);

/* Grid::InitOptimization + InitTree (Grid.cpp:1474-1491, 934-980) and the
 * copy counts of GetVerts / InitNeighborsByEdge (Grid.cpp:815-856, 309-319). */
int admm_init(admm_ctx *ctx);
/* Grid::Compute (Grid.cpp:1394-1412): dual update, potential adjust and the
 * subproblem of every tree (Tree.cpp:64-274). */
int admm_compute(admm_ctx *ctx);
/* Grid::Communicate (Grid.cpp:1509-1526, 1639-1663).  One strip: nothing to
 * exchange.  Several strips, chain solver: RCCL all-to-all (grouped send /
 * receive) of the chain labels, or of their marginals with
 * ADMM_EXCHANGE_DENSE.  Several ranks, edge solver: CommunicateByEdge, every
 * rank sends each neighbouring rank its partial consensus sums of the vertices
 * they share (grouped RCCL send / receive); admm_update then adds what arrived
 * in ascending order of the neighbours' ranks, as Grid.cpp:163-173 does, so it
 * must follow an admm_communicate. */
int admm_communicate(admm_ctx *ctx);
/* The same exchange when all n strips of a problem live in this process (one
 * context per strip, rank order, on one GPU or on several): device-to-device
 * copies ordered with events between the contexts' streams, no RCCL and no
 * host synchronisation.  The reference has no such mode (it is one process per
 * rank); it serves single-process multi-GPU drivers and lets the strip logic be
 * tested on one GPU. */
int admm_communicate_group(admm_ctx **ctxs, int n);
/* The same exchange staged through host memory, for a caller that brings its
 * own transport (the reference's is MPI_Isend / MPI_Irecv on packed host
 * buffers, Grid.cpp:1648-1662).  admm_exchange_counts gives the 4-byte
 * elements (chain solver: labels, or k floats per label with
 * ADMM_EXCHANGE_DENSE; edge solver: k floats per shared vertex) sent to and
 * received from every rank; admm_exchange_pack_host fills host_send,
 * peer-major, and waits for it; the caller moves block h of every rank to
 * rank h; admm_exchange_unpack_host takes the received blocks, peer-major. */
int admm_exchange_counts(admm_ctx *ctx, int64_t *send_counts, int64_t *recv_counts);
int admm_exchange_pack_host(admm_ctx *ctx, int32_t *host_send);
int admm_exchange_unpack_host(admm_ctx *ctx, const int32_t *host_recv);
/* Grid::Update + UpdateMuArrAfterComm (Grid.cpp:186-193, 163-181).  Chain
 * solver: the consensus of a node is a function of the two labels it received,
 * so Update only makes the newest labels of every set of chains the ones the
 * crossing chains see from the next admm_compute on (a transposition of both
 * label arrays, or, on one strip where the sweep can read them in place, a
 * change of two pointers and no launch). */
int admm_update(admm_ctx *ctx);
/* Grid::ComputeLinearObj (Grid.cpp:1665-1680): this strip's share of
 * rho * sum over trees of <theta, mu>; the caller sums over strips as
 * admm.cpp:95 does. */
int admm_linear_objective(admm_ctx *ctx, double *obj);
/* Chain solver: this strip's share of the energy of the decoded labelling
 * (primal) and of the sum of chain minima under theta + lambda (dual).
 * Either pointer may be NULL.  Edge solver: primal = the linear objective,
 * dual = NaN.  No counterpart in the reference, which decodes no labels. */
int admm_objective(admm_ctx *ctx, double *primal, double *dual);
/* Chain solver: the linear objective without stopping the stream.  Compute
 * leaves every chain's term on the device; admm_objective_enqueue queues their
 * sum into device slot `slot` (1 <= slot < 4096) and admm_objective_read waits
 * for the stream and copies slots [slot, slot+n) out.  The reference prints
 * its objective every iteration (admm.cpp:94-99); a driver that wants the
 * same trace without a host round trip per iteration queues one slot per
 * iteration and reads them at the end.  A queued slot is complete after the
 * second admm_compute that follows (its sums ride along with those launches),
 * or as soon as admm_objective_read asks for it, whichever comes first (the
 * two ways add the same binary64 terms, the unary ones in a different order). */
int admm_objective_enqueue(admm_ctx *ctx, int slot);
int admm_objective_read(admm_ctx *ctx, int slot, int n, double *linear);
/* Waits for everything queued on the context's stream. */
int admm_synchronize(admm_ctx *ctx);
/* Without waiting: *busy = 1 while work queued on the context's stream is
 * still running, 0 once it has all completed.  Lets a driver put a deadline on
 * an exchange whose partner may never answer (the reference's MPI_Wait,
 * Grid.cpp:1660-1662, has none). */
int admm_query(admm_ctx *ctx, int *busy);
/* Grid::FinalizeOptimization (Grid.cpp:1414-1416): releases the communicator. */
int admm_finalize(admm_ctx *ctx);

/* Results.  Labels: decoded labelling of this strip's rows,
 * [batch][strip rows][cols] (chain solver).  Consensus: vertMuArr after
 * Update, [batch][strip rows * cols][k]; edge solver: [local vertices][k],
 * which is [V][k] on one rank. */
int admm_download_labels(admm_ctx *ctx, int32_t *labels);
int admm_download_consensus(admm_ctx *ctx, float *consensus);
/* admm_download_labels for callers that cannot allocate device memory (a
 * serving loop that reads the labelling every iteration, a harness that
 * forbids hipMalloc): it has no counterpart in the reference.  Same output,
 * same kernel; the device staging between the decode kernel and the copy to
 * the host is a fixed array inside the library's code object (16 MiB per GPU,
 * held until the process ends) instead of context memory, and the call makes
 * no hipMalloc / hipFree.  A strip of more than ADMM_STATIC_LABELS_MAX labels
 * (batch * strip rows * cols) is refused with ADMM_E_ARG.  All contexts on one
 * GPU share that array: calls that overlap in time on one GPU (two threads, two
 * contexts) race; calls one after the other do not, since each waits for its
 * copy before it returns. */
#define ADMM_STATIC_LABELS_MAX (4 << 20)
int admm_download_labels_static(admm_ctx *ctx, int32_t *labels);

/* Solver state for parity tests.  Edge solver (members of class Edge,
 * Tree.h:26-35), this rank's edges: 0 x_mu_node, 1 x_potential_node,
 * 2 edge_mu_node, 3 lambda_node ([E][2k]); 4 x_mu_edge, 5 x_potential_edge
 * ([E][k*k]).
 * Chain solver: 10 lambda of the row chains [strip rows][cols][k],
 * 11 lambda of the column chains [rows][strip cols][k], 12 per-copy unary
 * share [strip rows][cols][k]; 13 row-chain labels [strip rows][cols],
 * 14 column-chain labels [rows][strip cols] (int32, -1 before the first
 * Compute; pass an int32 buffer cast to float*). */
int admm_download_state(admm_ctx *ctx, int which, float *out);
/* Number of elements admm_download_state writes for `which` (0 if unknown). */
size_t admm_state_size(const admm_ctx *ctx, int which);

/* Strip partition (Grid::PartitionByRow, Grid.cpp:377-390): n items over
 * nranks, the remainder going to the lowest ranks.  Pure host arithmetic. */
void admm_partition(int n, int nranks, int rank, int *start, int *count);
/* This context's strips. */
int admm_strip_info(const admm_ctx *ctx, int *row0, int *nrows, int *col0, int *ncols);
/* Edge solver: this rank's slice of the edge list (Grid::PartitionByEdge,
 * Grid.cpp:446-454) and the number of distinct vertices its edges touch
 * (GetVerts, Grid.cpp:815-856); their global ids in ascending order, which is
 * the row order of admm_download_consensus (vSelfVec). */
int admm_edge_local_info(const admm_ctx *ctx, int *edge0, int *nedges, int *nverts);
int admm_edge_local_vertices(const admm_ctx *ctx, int32_t *ids);
/* The exchange plan admm_create builds for rank `rank` of `nranks` from the
 * whole edge list (GetVerts + InitNeighborsByEdge, Grid.cpp:815-856, 198-331):
 * pure host arithmetic, no device.  info[5] = first edge, edges, distinct
 * vertices, neighbouring ranks, shared entries in total.  The arrays may be
 * NULL (call once for the sizes): verts / counts = global id and global copy
 * count (vertCnt after Grid.cpp:309-319) per local vertex; neigh_ranks
 * ascending, neigh_off[q] .. neigh_off[q+1] = neighbour q's entries of
 * shared_local (local vertex ids, ascending: outNeighborMapEdge) and
 * shared_counts (that neighbour's own copy count of the vertex). */
int admm_edge_plan(const int32_t *edge_list, int num_edges, int nranks, int rank, int32_t *info,
                   int32_t *verts, int32_t *counts, int32_t *neigh_ranks, int32_t *neigh_off,
                   int32_t *shared_local, int32_t *shared_counts);

/* Multi-GPU bootstrap.  Rank 0 fills a 128-byte RCCL id, the caller ships it
 * to every rank by its own means (the reference's is MPI_Init), and every
 * rank calls admm_comm_init before the first admm_communicate. */
#define ADMM_COMM_ID_BYTES 128
int admm_comm_get_unique_id(void *id);
int admm_comm_init(admm_ctx *ctx, const void *id);

/* Synthetic inputs (Grid::Generate, Grid.cpp:631-645, with a reproducible
 * generator instead of rand()): 32-bit LCG s = s*1664525 + 1013904223,
 * u = (float)(s >> 8) / 2^24.  Fills out[0..n) and returns the new state. */
uint32_t admm_lcg_fill(uint32_t state, int64_t n, float *out);
/* Edge list of a rows x cols grid in Grid::Generate's order; returns E. */
int admm_grid_edges(int rows, int cols, int32_t *edge_list);

/* Device-side timing of the solver kernels: average duration in milliseconds
 * of the Compute launches since the last reset, measured with HIP events on
 * the context's stream, and how many launches that covers.  enable = 0: off;
 * 1: every launch is timed; n > 1: every n-th launch (two event records around
 * a launch cost several microseconds of stream time, which a short iteration
 * feels; a sample of the launches gives the same average). */
int admm_profile_reset(admm_ctx *ctx, int enable);
int admm_profile_read(admm_ctx *ctx, double *compute_ms_avg, int *launches);

/* Which sweep kernel admm_compute launches for this context (chain solver,
 * after admm_init): lanes a label vector is padded to, lanes that share one
 * target label, and whether the kernel specialised for Potts tables (zero
 * diagonal, one non-negative value everywhere else) is the one chosen.  A
 * diagnostic without a counterpart in the reference. */
int admm_sweep_kernel_info(admm_ctx *ctx, int *label_lanes, int *lanes_per_label, int *potts);
/* The same as the name a kernel trace shows for it, e.g.
 * "chain_sweep_kernel<32, 1, 8, 4, false, true, true, true>" (labels padded to,
 * lanes per label, prefetch groups, wavefronts per SIMD, dual evaluation, Potts
 * step, int16 multiplier counts, path compression for the trace-back), so that a measurement can say which instance it
 * timed.  buf receives at most len bytes, NUL-terminated. */
int admm_sweep_kernel_name(admm_ctx *ctx, char *buf, int len);

/* The launch options the last admm_compute ran with, as "name=value" pairs
 * separated by blanks (chain solver: wavefronts, mixed workgroups, pairs of
 * column wavefronts per die, die remap, wavefronts per half chain, direct
 * labels, deferred sums, width of the multipliers; edge solver: whether the
 * solve summed the objective).  Several of them follow tuning knobs in the
 * environment, so a measurement records them.  Empty before the first
 * admm_compute.  A diagnostic without a counterpart in the reference. */
int admm_launch_info(admm_ctx *ctx, char *buf, int len);

/* The binary64 logarithm the edge solver takes of every marginal
 * (Tree.cpp:120-133), applied on the device to n host values: y[i] = log x[i].
 * For accuracy tests of that function; no counterpart in the reference. */
int admm_edge_log_values(int device, const double *x, double *y, int64_t n);

/* The exponential the edge solver takes three times per table entry
 * (Tree.cpp:196, 219, 261), applied on the device to n binary32 arguments:
 * own[i] = its value, lib[i] = the device library's exp of the same argument,
 * was_nan[i] = 1 where the routine flagged a NaN argument (its value is then
 * not the reference's; the solver turns the sums it enters into NaN).  For
 * equality tests; no counterpart in the reference. */
int admm_edge_exp_values(int device, const float *x, double *own, double *lib, unsigned char *was_nan, int64_t n);

/* Number of HIP devices visible to this process (0 when there is none). */
int admm_device_count(void);
/* Free and total device memory of `device` in bytes as the HIP runtime reports
 * them (hipMemGetInfo): lets a test see that a call allocated nothing.  A
 * diagnostic without a counterpart in the reference. */
int admm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* Library version string. */
const char *admm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H_ */
