// C ABI of libadmm_hip.so (see include/admm_hip.h): context management, data
// movement and kernel launches for the two subproblem solvers.
#include "../../include/admm_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "chain_kernels.hpp"
#include "edge_kernels.hpp"

using namespace admm;

// Staging of admm_download_labels_static: device memory that is part of the
// loaded code object, so the call that uses it never allocates.  Every GPU the
// library runs on holds its own copy (16 MiB) until the process ends.  Host code
// never uses this name as an address: scratch_setup() asks the runtime where the
// current device keeps its copy and only that pointer is handed to kernels and copies.
static __device__ __align__(256) int32_t scratch[ADMM_STATIC_LABELS_MAX];

namespace {

thread_local std::string g_create_error;

// Where each device keeps its copy of `scratch` (null: not asked yet, or the last
// query failed, see g_scratch_err).  Read and written under g_scratch_mu only.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
int32_t *g_scratch_ptr[kMaxDevices] = {};
std::string g_scratch_err[kMaxDevices];

// The address of `device`'s copy, asked for once per device with that device current
// (a query that failed is made again by the next caller).  Null with *why set on failure.
int32_t *scratch_for(int device, std::string *why) {
  if (device < 0 || device >= kMaxDevices) {
    if (why) *why = "device ordinal out of range";
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) return g_scratch_ptr[device];
  int32_t *ptr = nullptr;
  const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !ptr) {
    g_scratch_err[device] = std::string("hipGetSymbolAddress(scratch) failed: ") + hipGetErrorString(e);
    if (why) *why = g_scratch_err[device];
    return nullptr;
  }
  g_scratch_ptr[device] = ptr;
  return ptr;
}

struct Strip {
  int row0 = 0, nrows = 0, col0 = 0, ncols = 0;
};

// Chain solver on several strips.  Offsets (in labels) of the per-peer send blocks inside sendbuf:
//   a-blocks: my row-chain labels on peer h's columns, nr x nc_h each, then
//   b-blocks: my column-chain labels on peer h's rows, nc x nr_h each.
// Host arithmetic on the problem's sizes, worked out once per context.
struct ExchangePlan {
  std::vector<long long> off_a, off_b;  // size G + 1
  std::vector<int> r0, nr, c0, nc;      // every rank's strips
};

}  // namespace

struct admm_ctx {
  admm_desc d{};
  std::string err;
  hipStream_t stream = nullptr;
  bool have_unary = false, have_pair = false, inited = false;
  // chain solver: th_r / th_c hold the per-copy shares already (admm_init scales
  // the uploaded unaries in place), so an admm_init that follows a new pairwise
  // table alone must not scale them a second time
  bool theta_scaled = false;
  // chain solver, unaries uploaded a share at a time: which rows of the row
  // strip / of the column strip have arrived since the last restart
  std::vector<uint8_t> seen_r, seen_c;
  std::vector<void *> allocs;

  // ---- edge solver ----
  int V = 0, E = 0;
  std::vector<int> h_edges, h_inc_off, h_inc;
  int *d_edges = nullptr, *d_inc_off = nullptr, *d_inc = nullptr;
  float *d_vert_pot = nullptr, *d_edge_pot = nullptr;
  int pair_per_edge = 0;
  float *d_mu_node = nullptr, *d_pot_node = nullptr, *d_lam_node = nullptr;
  float *d_mu_edge = nullptr, *d_pot_edge = nullptr, *d_m_vert = nullptr;
  double *d_partial = nullptr;
  int n_partial = 0;
  double *d_edge_obj = nullptr;  // edge solver: one objective term per workgroup of the last solve
  int edge_obj_n = 0;            // terms the last solve wrote; 0 = none since admm_init
  double *d_obj_slots = nullptr;  // [n_obj_slots][2] (primal, linear) queued objective values
  int n_obj_slots = 0;
  // edge solver on several ranks (EDGE partition, Grid.cpp:445-454): V and E
  // above are this rank's vertices and edges, vertices renumbered 0..V-1 in
  // ascending global order (vSelfMapG2L)
  int Vglobal = 0, Eglobal = 0, e0 = 0;
  std::vector<int> h_lverts;         // global id per local vertex
  std::vector<int> h_cnt;            // global copy count per local vertex (vertCnt after Grid.cpp:309-319)
  std::vector<int> nb_rank, nb_off;  // neighbouring ranks, ascending, and where each one's shared block starts (+ end)
  std::vector<int> h_send_idx;       // local vertex per shared entry, neighbour-major (outNeighborMapEdge)
  std::vector<int> h_nb_cnt;         // the neighbour's own copy count per shared entry
  std::vector<float> h_unary_local;  // unaries of the local vertices, kept for admm_init
  int *d_cnt = nullptr, *d_send_idx = nullptr, *d_rcv_off = nullptr, *d_rcv_pos = nullptr;
  float *d_m_part = nullptr, *d_sendv = nullptr, *d_recvv = nullptr;

  // ---- chain solver ----
  Strip st;
  int KP = 0;
  float inv_k = 0.f;
  // row strip: [batch][nrows][cols][k]; column strip: [batch][rows][ncols][k].
  // With one strip the column-side pointers alias the row-side ones.
  float *th_r = nullptr, *lam_r = nullptr;
  float *th_c = nullptr, *lam_c = nullptr;
  // One strip (the whole grid on this device): the multipliers of the column
  // copies are those of the row copies negated, bit for bit (opposite
  // increments from a common zero), so only lam_r is kept; the column chains
  // read it with the sign flipped and store nothing, and the row chains write
  // their update into lam_c, which is swapped with lam_r after the sweep.
  bool lam_shared() const { return d.nranks == 1; }
  // The multiplier arrays hold int16 counts of penalty / 2 instead of binary32
  // values (the LAM16 kernels): decided by admm_init, dropped for good once the
  // counts could leave int16.
  bool lam16 = false;
  int lam16_sweeps = 0, lam16_limit = 32000;
  // ... and as int8 while every count is known to be far from +-127.  A count moves by at
  // most one per sweep, so a scan of the arrays that found every |n| below lam8_thr after c
  // sweeps covers the sweeps up to c + 126 - lam8_thr.  Scans are queued every lam8_every
  // sweeps; their results travel to pinned host words and are looked at, without waiting,
  // when a later sweep is queued (a wait happens only when the host has run a cover's whole
  // length ahead of the device).  The first scan that reports a count at the threshold turns
  // the arrays into int16, for good.
  bool lam8 = false;
  int lam8_thr = 64, lam8_every = 16;
  int lam8_clean = 0;  // sweeps after which the last clean scan ran
  struct CountScan { hipEvent_t ev = nullptr; int at = 0; bool pending = false; };
  static constexpr int kScans = 8;
  CountScan scans[kScans];
  int scan_next = 0;
  int *d_scan_flag = nullptr;  // [kScans], device
  int *h_scan_flag = nullptr;  // [kScans], pinned host
  size_t lam_slack = 0;                          // floats of slack on either side of a multiplier array
  float *lamf_r = nullptr, *lamf_c = nullptr;    // several strips: binary32 copies, allocated when first needed
  // Labels the sweep reads and writes are bytes (label_t); labels that arrive
  // from other strips are int32, the exchange's wire format.
  label_t *xrow_r = nullptr;   // [batch][nrows][cols]   labels of my row chains
  label_t *xcolT_c = nullptr;  // [batch][ncols][rows]   labels of my column chains
  int *xcolT_r = nullptr;      // [cols][nrows]          column-chain labels on my rows   } several strips only: with one
  int *xrow_c = nullptr;       // [rows][ncols]          row-chain labels on my columns   } strip they are the two above
  label_t *xo_r = nullptr;     // [batch][nrows][cols]   column-chain labels on my rows, row-major: what the row sweep reads
  label_t *xo_c = nullptr;     // [batch][ncols][rows]   row-chain labels on my columns, column-major: what the column sweep reads
  bool one_strip() const { return d.nranks == 1; }
  // One strip, direct labels: the sweep reads the crossing chains' labels straight from the
  // array those chains wrote (one byte per step, a grid row or column apart) instead of from a
  // transposed copy, so admm_update launches nothing.  The sweep of one set must not write
  // where the other set reads in the same launch, hence two buffers per set: the published one
  // (what the crossing chains read: the labels as of the last admm_update) and the other, which
  // the sweep writes; xrow_r / xcolT_c always point at the newest labels.  The transposed copies
  // xo_r / xo_c are then made only when a kernel other than the sweep asks for them.
  bool direct_labels = false;
  label_t *xrow_buf[2] = {nullptr, nullptr}, *xcolT_buf[2] = {nullptr, nullptr};
  label_t *xrow_pub = nullptr, *xcolT_pub = nullptr;
  bool xo_stale = false;
  uint8_t *bp_r = nullptr, *bp_c = nullptr;
  float *d_W = nullptr, *d_Wsplit = nullptr, *d_Wt = nullptr;
  float *d_WsplitT = nullptr, *d_WtT = nullptr;  // transposes, for the half chains swept backwards
  // the table the sweep runs on (Wt) has a zero diagonal and one non-negative
  // value everywhere else: the sweep kernel specialised for that form applies
  bool potts = false;
  // ... or is a function g(|a - b|) of the label distance that is constant from distance 4 on and
  // nowhere above that constant (truncated linear / quadratic ...): the sweep specialised for it
  bool trunc3 = false;
  float *d_chain_min = nullptr;
  // The linear objective of a solve has two per-chain terms: the pairwise one
  // is written by that solve's sweep, the unary one by the NEXT sweep (which
  // has the operands in registers) or, when somebody asks before that, by a
  // gather kernel.
  double *d_chain_lin = nullptr;    // per chain: pairwise energy of its new labels (written by Compute)
  double *d_chain_unary = nullptr;  // per chain: unary energy of the labels the last sweep started from
  // (both exist twice and every sweep writes the pair the one before it did not, so that sums
  // over the last sweep's terms can run inside the next sweep's launch)
  double *d_chain_lin_alt = nullptr, *d_chain_unary_alt = nullptr;
  // Sums of per-half-chain terms into objective slots that are queued but not launched yet:
  // they ride along with the next sweep (one spare workgroup of its launch), or are launched
  // on their own when somebody reads a slot first.
  struct RedJob { double *dst; const double *part; int n; };
  std::vector<RedJob> red_q;
  bool lin_valid = false;
  int pending_slot = 0;             // queued objective whose unary term is still to come, or 0
  int swept_slot = 0;               // queued objective whose unary term the last sweep left, unsummed, in d_chain_unary
  int *sendbuf = nullptr;
  float *sendbuf_mu = nullptr, *recvbuf_mu = nullptr;  // ADMM_EXCHANGE_DENSE: the same blocks as float k-vectors
  std::vector<float> h_unary_stage;

  // staging of the download calls (decoded labels, consensus vectors, widened
  // state): grown on demand, kept between calls
  void *d_stage = nullptr;
  size_t stage_bytes = 0;

  ExchangePlan plan;
  bool plan_ready = false;
  ncclComm_t comm = nullptr;
  hipEvent_t ev_packed = nullptr, ev_pulled = nullptr;  // admm_communicate_group
  // admm_communicate_group pulled the labels straight into the arrays the sweep reads
  // (pull_labels_kernel): the admm_update that follows has nothing left to transpose
  bool pulled = false;
  // (rank 0's context remembers whether its group can pull: the group's size, its last context, the answer)
  int pull_group_n = 0;
  const admm_ctx *pull_group_last = nullptr;
  bool pull_group_ok = false;
  label_t *xo_r_alt = nullptr, *xo_c_alt = nullptr;  // where a pull writes; admm_update swaps them in

  // the options the last Compute launch ran with (admm_launch_info)
  std::string launch_info;

  // ---- profiling ----
  bool prof = false;
  int prof_every = 1, prof_seen = 0;  // every prof_every-th Compute is timed
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
};

namespace {

int fail(admm_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(c, call)                                                              \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess)                                                            \
      return fail((c), ADMM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                  __FILE__, __LINE__);                                                \
  } while (0)

#define NCCL_TRY(c, call)                                                             \
  do {                                                                                \
    ncclResult_t r__ = (call);                                                        \
    if (r__ != ncclSuccess)                                                           \
      return fail((c), ADMM_E_COMM, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__), \
                  __FILE__, __LINE__);                                                \
  } while (0)

// A group of RCCL point-to-point calls.  ncclGroupStart is always matched by
// ncclGroupEnd, also when a call inside fails (the thread must not be left
// inside an open group): the first error is remembered and end() reports it.
class NcclGroup {
  admm_ctx *c_;
  ncclResult_t first_ = ncclSuccess;
  const char *what_ = "";
  bool open_ = false;

 public:
  explicit NcclGroup(admm_ctx *c) : c_(c) {
    first_ = ncclGroupStart();
    open_ = first_ == ncclSuccess;
    if (!open_) what_ = "ncclGroupStart";
  }
  NcclGroup(const NcclGroup &) = delete;
  NcclGroup &operator=(const NcclGroup &) = delete;
  bool ok() const { return first_ == ncclSuccess; }
  void add(ncclResult_t r, const char *what) {
    if (first_ == ncclSuccess && r != ncclSuccess) { first_ = r; what_ = what; }
  }
  int end() {
    if (open_) {
      open_ = false;
      const ncclResult_t r = ncclGroupEnd();
      if (first_ == ncclSuccess && r != ncclSuccess) { first_ = r; what_ = "ncclGroupEnd"; }
    }
    return first_ == ncclSuccess ? ADMM_OK : fail(c_, ADMM_E_COMM, "%s failed: %s", what_, ncclGetErrorString(first_));
  }
  ~NcclGroup() { if (open_) (void)ncclGroupEnd(); }
};

template <typename T>
int dev_alloc(admm_ctx *c, T **p, size_t n) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess)
    return fail(c, ADMM_E_NOMEM, "hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
  c->allocs.push_back(q);
  *p = (T *)q;
  // ADMM_POISON=<byte> (debugging knob): every allocation starts filled with that byte, so a read
  // of memory nothing initialised shows up on every run and not only when the memory was dirty
  const char *poison = getenv("ADMM_POISON");
  if (poison) {
    (void)hipMemset(q, atoi(poison) & 0xFF, std::max<size_t>(n, 1) * sizeof(T));
    (void)hipDeviceSynchronize();
  }
  return ADMM_OK;
}

// Device staging of at least `bytes` for a download call; up to kStageKeepBytes it is
// kept by the context, so only a call that needs more than the calls before it allocates.
int stage_buf(admm_ctx *c, void **p, size_t bytes) {
  if (bytes > c->stage_bytes) {
    if (c->d_stage) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(c->d_stage);
      c->d_stage = nullptr;
      c->stage_bytes = 0;
    }
    const hipError_t e = hipMalloc(&c->d_stage, bytes);
    if (e != hipSuccess) {
      c->d_stage = nullptr;
      return fail(c, ADMM_E_NOMEM, "hipMalloc of %zu bytes of staging failed: %s", bytes, hipGetErrorString(e));
    }
    c->stage_bytes = bytes;
  }
  *p = c->d_stage;
  return ADMM_OK;
}

// The staging is kept between calls only while it is small: a download of the consensus
// vectors or of widened state of a large grid takes hundreds of megabytes, which go back
// to the device when the call that asked for them returns (every download call waits for
// its copy, so nothing is in flight then).
constexpr size_t kStageKeepBytes = 64u << 20;
struct StageRelease {
  admm_ctx *c;
  explicit StageRelease(admm_ctx *ctx) : c(ctx) {}
  StageRelease(const StageRelease &) = delete;
  StageRelease &operator=(const StageRelease &) = delete;
  ~StageRelease() {
    if (c && c->d_stage && c->stage_bytes > kStageKeepBytes) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(c->d_stage);
      c->d_stage = nullptr;
      c->stage_bytes = 0;
    }
  }
};

// Host -> device on the context's own stream, complete on return.  (The stream does
// not synchronise with the null stream, so nothing the kernels read goes through that one.)
hipError_t h2d_sync(admm_ctx *c, void *dst, const void *src, size_t bytes) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}

int grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  return (int)std::max<long long>(1, std::min<long long>(g, 256 * 32));
}

int next_pow2(int k) {
  int p = 2;
  while (p < k) p <<= 1;
  return p;
}

bool is_chain(const admm_ctx *c) { return c->d.solver == ADMM_SOLVER_CHAIN_MINSUM; }

// ------------------------------------------------------------------ chain launches

// ADMM_SWEEP_LDS_PAD=<bytes> (tuning knob): dynamic LDS a sweep launch asks for on top of what
// it uses, which lowers the number of workgroups a compute unit holds.
inline unsigned sweep_pad_lds() {
  const char *e = getenv("ADMM_SWEEP_LDS_PAD");
  return e ? (unsigned)std::max(0, atoi(e)) : 0u;
}

struct SweepShape {
  int KP, NS;
};

// Lane layout: when there are few chains, several lanes share a target label
// (NS > 1) so that a wavefront's 64 lanes are all busy on one chain; when
// chains are plentiful, narrow labels pack several chains into a wavefront.
SweepShape pick_shape(int KP, long long nchains) {
  nchains *= 2;  // every chain is swept as two half chains, one wavefront slot each
  const long long simd_slots = 256LL * 4 * 2;  // two wavefronts per SIMD
  int NS = std::min(64 / KP, KP);
  if (KP <= 16) {
    // packing 64/KP chains per wavefront still fills the machine?
    if (nchains / (64 / KP) >= simd_slots) NS = 1;
  }
  if (KP == 64) NS = 1;
  // 32 labels: two chains per wavefront (each lane scans all 32 sources, no
  // cross-lane combine) once that still gives every SIMD two wavefronts
  if (KP == 32 && nchains / 2 >= simd_slots) NS = 1;
  // tuning knob: ADMM_SWEEP_NS=<n> forces the split when a kernel for it exists
  if (const char *e = getenv("ADMM_SWEEP_NS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= std::min(64 / KP, KP) && (v & (v - 1)) == 0 && (KP != 32 || v <= 2) && (v == 1 || v == std::min(64 / KP, KP))) NS = v;
  }
  return {KP, NS};
}

// Template arguments the general kernel is built with for a lane layout.
// GPB: prefetch distance in load groups of NS steps; 64 source labels per lane
// need the registers of a 2-wavefront-per-SIMD allocation.  MINW: arg-min of
// 32+ candidates per lane, or a split chain (scarce chains, so occupancy is not
// the limit): room for 256 registers (8 or 16 labels packed several chains to a
// wavefront spill at 128 registers).
// 32 source labels per lane with one lane per label (two chains per wavefront, the dense 32-label
// table at full occupancy): four wavefronts per SIMD in 128 registers, with operands fetched four
// steps ahead and half of a step's messages read from LDS only after the arg-min work (MVLATE in
// the kernel) - at 160 registers three wavefronts fitted, and the 4096 wavefronts of a 2048 x 2048
// grid ran as one round of 3072 and a tail of 1024 alone on their SIMDs.
constexpr int sweep_gpb(int KP, int NS) { return (NS >= 4 || KP / NS >= 64 || (KP == 32 && NS == 1)) ? 4 : 8; }
// (2 and 4 labels with one lane per label: 32 and 16 chains per wavefront, whose label staging takes
// the registers of one and of three wavefronts per SIMD)
constexpr int sweep_minw(int KP, int NS) { return (KP == 32 && NS == 1) ? 4 : (KP == 64 && NS == 1) ? 3 : ((KP / NS >= 32 || NS > 1) ? 2 : (KP >= 4 ? 3 : 1)); }

template <int KP, int NS, bool DUAL>
void launch_sweep_t(const ChainArgs &a, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((chain_sweep_kernel<KP, NS, sweep_gpb(KP, NS), sweep_minw(KP, NS), DUAL>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
}

// Two wavefronts per half chain (the kernels built with SPLIT), for launches with at most one
// wavefront per SIMD (one strip of eight or of four of the 2048 x 2048 grid: 512 or 1024
// wavefronts).  A wavefront alone on its SIMD issues one instruction every 4.8 clock ticks
// whatever its kind, so halving the instructions per wavefront and step is what shortens the
// sweep: 0.178 -> 0.116 ms per sweep for a strip of eight, 0.159 -> 0.143 ms for one of four
// (there the second wavefront shares a SIMD with another chain's); with two wavefronts per SIMD
// to begin with (a strip of two) it changes nothing.  (Since then 0.110 and 0.132 ms: blocks of
// 16 steps, operands in one burst for a lone shell wavefront, issue priority for a shell that
// shares its SIMD - see the kernel.)  ADMM_SWEEP_TWOWAVE=0/1 (tuning and test
// knob) overrides.
bool twowave_chosen(int KP, int waves, bool trunc = false) {
  if (KP != 32 && !trunc) return false;
  if (const char *e = getenv("ADMM_SWEEP_TWOWAVE")) return atoi(e) != 0;
  return waves <= 1024;
}
// (the shell fetches its operands sixteen steps ahead there: 3 % over eight, 0.118 -> 0.115 ms per
// sweep for a strip of eight, 0.145 -> 0.139 for one of four)
constexpr int kSplitGpb = 16;
// MINW: 2 where every wavefront of the launch has a SIMD to itself (that instance fetches a block's
// operands in one burst, see the kernel), 3 otherwise (two wavefronts per SIMD is all such a launch has).
template <int KP, int MINW>
void launch_sweep_potts_split_t(const ChainArgs &a, int blocks, hipStream_t s, bool lam16, bool lam8) {
  if (lam8) hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, kSplitGpb, MINW, false, true, true, false, true, true>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
  else if (lam16) hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, kSplitGpb, MINW, false, true, true, false, false, true>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
  else hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, kSplitGpb, MINW, false, true, false, false, false, true>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
}
// (the split doubles the wavefronts; the chip has 1024 SIMDs)
inline bool split_lone(int waves) { return 2 * waves <= 1024; }

// Steps per unrolled block of the Potts kernels (= how far ahead the operands are fetched, in one
// burst per block): 16 where several steps of a chain share a 128-byte line (32: spills), 8 with
// 32 and 64 labels (the records of the path compression are one per block of eight steps).
constexpr int potts_gpb(int KP) { return KP <= 16 ? 16 : 8; }
template <int KP, bool PCOMP = false>
void launch_sweep_potts_t(const ChainArgs &a, int blocks, hipStream_t s) {
  // a step needs few registers: four wavefronts per SIMD, operands fetched a block ahead (potts_gpb)
  hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, potts_gpb(KP), 4, false, true, false, PCOMP>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
}
template <int KP, bool PCOMP = false>
void launch_sweep_potts16_t(const ChainArgs &a, int blocks, hipStream_t s, bool lam8) {
  // (32 and 64 labels with a load per step: operands 16 steps ahead instead of 8 were 1 to 2 % slower)
  if (lam8) hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, potts_gpb(KP), 4, false, true, true, PCOMP, true>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
  else hipLaunchKernelGGL((chain_sweep_kernel<KP, 1, potts_gpb(KP), 4, false, true, true, PCOMP>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
}

// Path compression in the Potts kernel (the forward pass tracks, per block of eight
// steps, where every label's path comes from; the trace-back hops over blocks and fills
// them in with one lane per block): pays when the trace-back is a visible part of the
// launch and the forward pass has issue slots to spare, i.e. long chains of 32 or 64
// labels with four wavefronts per SIMD (2048 x 2048, k = 32: trace-back 121 k -> 42 k clock
// ticks per wavefront of 900 k, forward pass + 2 %: 0.454 -> 0.425 ms per sweep; 2048 x 2048,
// k = 64: 0.927 -> 0.880 ms; with two wavefronts per SIMD + 3 % to - 3 %).  A lone wavefront per
// SIMD is bound by the instructions it issues and loses more in the forward pass than
// the trace-back gives back; eight-label chains are short and already walk eight chains
// at once.  ADMM_SWEEP_PCOMP=0/1 (tuning and test knob) overrides the choice.
bool pcomp_chosen(int KP, int waves) {
  if (KP < 32) return false;
  if (const char *e = getenv("ADMM_SWEEP_PCOMP")) return atoi(e) != 0;
  return waves >= 4 * 1024;
}

// ADMM_SWEEP_POTTS=0 (tuning and test knob) keeps the general kernel on a Potts table.
bool potts_enabled() {
  const char *e = getenv("ADMM_SWEEP_POTTS");
  return !(e && atoi(e) == 0);
}

// 8 labels or more: the Potts kernel (with four or more chains per wavefront its trace-back
// runs one lane per chain, otherwise on the scalar unit).  2 or 4 labels: the general
// kernel's scan is as short.
bool potts_chosen(const admm_ctx *c, SweepShape sh) { return c->potts && sh.KP >= 8 && potts_enabled(); }

// The step specialised for truncated tables: 16 label lanes, and only where the Potts kernel
// (which is the same idea with no neighbours) does not apply.
bool trunc_chosen(const admm_ctx *c, SweepShape sh) { return c->trunc3 && sh.KP == 16 && !potts_chosen(c, sh); }

// One strip: does the sweep read the crossing chains' labels where those chains wrote them (one
// byte per step, a grid row or column apart, fetched a chunk of 64 steps ahead), so that
// admm_update launches nothing?  The strided bytes cost the sweep extra memory requests: free
// while it is bound by latency or by the vector ALU (512 x 512, k = 16: 19.3 k -> 21.3 k it/s;
// 64 x 64, k = 4: 36.7 k -> 43.2 k; 256 grids of 256 x 256, k = 8: 2090 -> 2250), 5 % of a sweep
// that runs at the rate of the memory system (2048 x 2048, k = 32 Potts with four wavefronts per
// SIMD: 0.350 -> 0.369 ms, more than the 10 us the two transpositions take), which therefore
// keeps the transposed copies.  ADMM_DIRECT_LABELS=0/1 (tuning and test knob) overrides.
bool direct_labels_chosen(const admm_ctx *c) {
  if (c->d.nranks != 1) return false;
  if (const char *e = getenv("ADMM_DIRECT_LABELS")) return atoi(e) != 0;
  const long long nr = (long long)c->d.batch * c->st.nrows, nc = (long long)c->d.batch * c->st.ncols;
  SweepShape sh = pick_shape(c->KP, nr + nc);
  const bool potts = potts_chosen(c, sh);
  if (potts || trunc_chosen(c, sh)) sh.NS = 1;
  const int CPW = 64 / (sh.KP * sh.NS);
  const long long waves = 2 * ((nr + CPW - 1) / CPW) + 2 * ((nc + CPW - 1) / CPW);
  return !(potts && sh.KP >= 32 && waves >= 4096);
}

bool red_deferred(const admm_ctx *c);

template <bool DUAL>
int launch_sweep(admm_ctx *c, ChainArgs &a, SweepShape sh) {
  const bool potts = !DUAL && potts_chosen(c, sh);
  const bool trunc = !DUAL && trunc_chosen(c, sh);
  if (potts || trunc) sh.NS = 1;  // a label's message stays in its one lane
  const int CPW = 64 / (sh.KP * sh.NS);
  int w = 0;
  for (int i = 0; i < a.nseg; i++) {
    a.wave_begin[i] = w;
    w += 2 * ((a.seg[i].nchains + CPW - 1) / CPW);  // left and right half of every group of chains
  }
  a.wave_begin[a.nseg] = w;
  if (w == 0) return ADMM_OK;
  int blocks = (w + kChainBlock / 64 - 1) / (kChainBlock / 64);
  a.mix_blocks = 0;
  a.mix_big = 0;
  if (a.nseg == 2 && !getenv("ADMM_SWEEP_NOMIX")) {  // (the variable: diagnostic, workgroups filled set after set)
    // rows and columns side by side in every workgroup (see ChainArgs)
    const int np0 = (a.wave_begin[1] - a.wave_begin[0]) / 2, np1 = (a.wave_begin[2] - a.wave_begin[1]) / 2;
    const int m = std::min(np0, np1);
    if (m > 0) {
      a.mix_blocks = m;
      a.mix_big = np1 > np0 ? 1 : 0;
      blocks = m + (std::max(np0, np1) - m + 1) / 2;
    }
  }
  hipStream_t s = c->stream;
  // A batch of grids: every die works on whole grids, so that the two reads of a node's
  // unaries (by its row chain and by its column chain) meet in that die's cache when they
  // are close in time (256 grids of 256 x 256, k = 8: 0.448 -> 0.437 ms per sweep).  One large
  // grid loses by it (2048 x 2048, k = 32: 0.350 -> 0.362 ms).  ADMM_SWEEP_XCD=0/1 overrides.
  a.xcd_remap = c->d.batch >= 8;
  if (const char *ex = getenv("ADMM_SWEEP_XCD")) a.xcd_remap = atoi(ex) != 0;
  // Column chains that share a wavefront read adjacent nodes: CPW * k entries of the
  // multipliers per step.  Below a 128-byte line (int8 counts of 2 x 32 labels) the rest of the
  // line is the next wavefront pair's; those two pairs then run on one die (see ChainArgs).
  // ADMM_SWEEP_PAIRXCD=0/1 (tuning and test knob) overrides.
  {
    const int entry = c->lam16 ? (c->lam8 ? 1 : 2) : 4;
    a.pair_xcd = !DUAL && a.mix_blocks > 0 && !a.xcd_remap && CPW * c->d.k * entry < 128;
    if (const char *ep = getenv("ADMM_SWEEP_PAIRXCD")) a.pair_xcd = atoi(ep) != 0 && a.mix_blocks > 0;
  }
  if (!DUAL) {
    const bool two = (potts && twowave_chosen(sh.KP, w)) || (trunc && twowave_chosen(sh.KP, w, true));
    char info[256];
    snprintf(info, sizeof(info), "wavefronts=%d mixed_workgroups=%d pair_xcd=%d xcd_remap=%d two_wavefronts_per_half_chain=%d "
             "direct_labels=%d deferred_sums=%d multipliers=%s",
             w, (a.mix_blocks > 0 && !two) ? 1 : 0, (a.pair_xcd && !two) ? 1 : 0, a.xcd_remap, two ? 1 : 0, c->direct_labels ? 1 : 0,
             red_deferred(c) ? 1 : 0, c->lam16 ? (c->lam8 ? "int8" : "int16") : "binary32");
    c->launch_info = info;
  }
  // queued sums ride along as one more workgroup, behind the sweep's own
  const int w_split = w / 2 + (a.red_jobs > 0 ? 1 : 0);
  a.red_block = blocks;
  if (a.red_jobs > 0) blocks++;
  if (potts && twowave_chosen(sh.KP, w)) {
    a.mix_blocks = 0;
    a.red_block = w / 2;
    // a workgroup per group of chains
    if (split_lone(w)) launch_sweep_potts_split_t<32, 2>(a, w_split, s, c->lam16, c->lam16 && c->lam8);
    else launch_sweep_potts_split_t<32, 3>(a, w_split, s, c->lam16, c->lam16 && c->lam8);
    HIP_TRY(c, hipGetLastError());
    return ADMM_OK;
  }
  const bool pcomp = potts && pcomp_chosen(sh.KP, w);
  if (potts && c->lam16) {
    switch (sh.KP) {
      case 8: launch_sweep_potts16_t<8>(a, blocks, s, c->lam8); break;
      case 16: launch_sweep_potts16_t<16>(a, blocks, s, c->lam8); break;
      case 32: if (pcomp) launch_sweep_potts16_t<32, true>(a, blocks, s, c->lam8); else launch_sweep_potts16_t<32>(a, blocks, s, c->lam8); break;
      case 64: if (pcomp) launch_sweep_potts16_t<64, true>(a, blocks, s, c->lam8); else launch_sweep_potts16_t<64>(a, blocks, s, c->lam8); break;
      default: return fail(c, ADMM_E_ARG, "no sweep kernel for KP=%d", sh.KP);
    }
    HIP_TRY(c, hipGetLastError());
    return ADMM_OK;
  }
  if (potts) {
    switch (sh.KP) {
      case 8: launch_sweep_potts_t<8>(a, blocks, s); break;
      case 16: launch_sweep_potts_t<16>(a, blocks, s); break;
      case 32: if (pcomp) launch_sweep_potts_t<32, true>(a, blocks, s); else launch_sweep_potts_t<32>(a, blocks, s); break;
      case 64: if (pcomp) launch_sweep_potts_t<64, true>(a, blocks, s); else launch_sweep_potts_t<64>(a, blocks, s); break;
      default: return fail(c, ADMM_E_ARG, "no sweep kernel for KP=%d", sh.KP);
    }
    HIP_TRY(c, hipGetLastError());
    return ADMM_OK;
  }
  if (trunc && twowave_chosen(sh.KP, w, true)) {
    a.mix_blocks = 0;
    a.red_block = w / 2;
    if (c->lam16 && c->lam8) hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, true, false, true, true, 3>), dim3(w_split), dim3(kChainBlock), sweep_pad_lds(), s, a);
    else if (c->lam16) hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, true, false, false, true, 3>), dim3(w_split), dim3(kChainBlock), sweep_pad_lds(), s, a);
    else hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, false, false, false, true, 3>), dim3(w_split), dim3(kChainBlock), sweep_pad_lds(), s, a);
    goto launched;
  }
  if (trunc) {
    if (c->lam16 && c->lam8) hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, true, false, true, false, 3>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
    else if (c->lam16) hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, true, false, false, false, 3>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
    else hipLaunchKernelGGL((chain_sweep_kernel<16, 1, 8, 4, false, false, false, false, false, false, 3>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a);
    goto launched;
  }
  if (!DUAL && c->lam16) {
    // the general step on integer counts (one lane per label; admm_init chose the counts only then)
#define SWEEP_COUNTS(kp) \
  if (sh.KP == kp && sh.NS == 1) { \
    if (c->lam8) hipLaunchKernelGGL((chain_sweep_kernel<kp, 1, sweep_gpb(kp, 1), sweep_minw(kp, 1), false, false, true, false, true>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a); \
    else hipLaunchKernelGGL((chain_sweep_kernel<kp, 1, sweep_gpb(kp, 1), sweep_minw(kp, 1), false, false, true, false, false>), dim3(blocks), dim3(kChainBlock), sweep_pad_lds(), s, a); \
    goto launched; \
  }
    SWEEP_COUNTS(16) SWEEP_COUNTS(32)
#undef SWEEP_COUNTS
    return fail(c, ADMM_E_STATE, "no sweep kernel on integer counts for KP=%d NS=%d", sh.KP, sh.NS);
  }
#define SWEEP_CASE(kp, ns) \
  if (sh.KP == kp && sh.NS == ns) { launch_sweep_t<kp, ns, DUAL>(a, blocks, s); goto launched; }
  SWEEP_CASE(2, 1) SWEEP_CASE(2, 2)
  SWEEP_CASE(4, 1) SWEEP_CASE(4, 4)
  SWEEP_CASE(8, 1) SWEEP_CASE(8, 8)
  SWEEP_CASE(16, 1) SWEEP_CASE(16, 4)
  SWEEP_CASE(32, 1) SWEEP_CASE(32, 2)
  SWEEP_CASE(64, 1)
#undef SWEEP_CASE
  return fail(c, ADMM_E_ARG, "no sweep kernel for KP=%d NS=%d", sh.KP, sh.NS);
launched:
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

void fill_seg_rows(const admm_ctx *c, ChainSeg &s, bool dual) {
  const admm_desc &d = c->d;
  s.theta = c->th_r; s.lam = c->lam_r; s.x = c->xrow_r; s.xin = c->xrow_r;
  s.lam_out = dual ? nullptr : (c->lam_shared() ? c->lam_c : c->lam_r);
  s.lam_sign = 0; s.xo = c->xo_r; s.bp = c->bp_r;
  s.xo_gstride = (long long)std::max(1, c->st.nrows) * d.cols; s.xo_cstride = d.cols; s.xo_sstride = 1;
  s.chain_min = dual ? c->d_chain_min : nullptr;
  s.chain_lin = c->d_chain_lin;
  s.chain_unary = c->d_chain_unary;
  s.nchains = d.batch * c->st.nrows;
  s.T = d.cols;
  s.cpg = std::max(1, c->st.nrows);
  s.gstride = (long long)c->st.nrows * d.cols;
  s.cstride = d.cols;
  s.sstride = 1;
}

void fill_seg_cols(const admm_ctx *c, ChainSeg &s, bool dual) {
  const admm_desc &d = c->d;
  s.theta = c->th_c; s.lam = c->lam_shared() ? c->lam_r : c->lam_c; s.x = c->xcolT_c; s.xin = c->xcolT_c;
  s.lam_out = (dual || c->lam_shared()) ? nullptr : c->lam_c;
  s.lam_sign = c->lam_shared() ? 0x80000000u : 0u; s.xo = c->xo_c; s.bp = c->bp_c;
  s.xo_gstride = (long long)std::max(1, c->st.ncols) * d.rows; s.xo_cstride = d.rows; s.xo_sstride = 1;
  s.chain_min = dual ? c->d_chain_min + (long long)d.batch * c->st.nrows : nullptr;
  s.chain_lin = c->d_chain_lin + 2LL * d.batch * c->st.nrows;
  s.chain_unary = c->d_chain_unary + 2LL * d.batch * c->st.nrows;
  s.nchains = d.batch * c->st.ncols;
  s.T = d.rows;
  s.cpg = std::max(1, c->st.ncols);
  s.gstride = (long long)d.rows * c->st.ncols;
  s.cstride = 1;
  s.sstride = c->st.ncols;
}

int chain_sweep_launch(admm_ctx *c, bool dual);
int settle_swept_now(admm_ctx *c);
int flush_red(admm_ctx *c);
bool red_deferred(const admm_ctx *c);
inline int chain_terms(const admm_ctx *c) { return 2 * c->d.batch * (c->st.nrows + c->st.ncols); }  // one per half chain

// counts -> binary32 multipliers: out[i] = in[i] * (penalty / 2), exact
template <typename TC>
__global__ void widen_multipliers_kernel(float *__restrict__ out, const TC *__restrict__ in, long long n, float hpen) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (float)in[i] * hpen;
}
// int8 counts -> int16 counts
__global__ void widen_counts_kernel(int16_t *__restrict__ out, const int8_t *__restrict__ in, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = in[i];
}
// *flag = 1 when some |count| >= thr (16 counts per lane and trip; n is a multiple of 16 or the tail is walked bytewise)
__global__ __launch_bounds__(256) void scan_counts_kernel(const int8_t *__restrict__ in, long long n, int thr, int *__restrict__ flag) {
  int hit = 0;
  const long long n16 = n / 16;
  const int4 *in4 = reinterpret_cast<const int4 *>(in);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    const int4 v = in4[i];
    const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int bb = 0; bb < 4; bb++) {
        const int cnt = (int)(signed char)(w[e] >> (8 * bb));
        hit |= (cnt >= thr) | (-cnt >= thr);
      }
  }
  for (long long i = n16 * 16 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    hit |= (in[i] >= thr) | (-in[i] >= thr);
  if (__builtin_amdgcn_ballot_w64(hit != 0) && (threadIdx.x & 63) == 0) *flag = 1;
}

// Launches the widening of `n` counts at `in` (int8 or int16, as the context holds them) into binary32.
void launch_widen_f32(admm_ctx *c, float *out, const float *in, long long n) {
  const float hpen = c->d.penalty * 0.5f;
  if (n <= 0) return;
  if (c->lam8)
    hipLaunchKernelGGL(widen_multipliers_kernel<int8_t>, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, out, reinterpret_cast<const int8_t *>(in), n, hpen);
  else
    hipLaunchKernelGGL(widen_multipliers_kernel<int16_t>, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, out, reinterpret_cast<const int16_t *>(in), n, hpen);
}

// Second buffers for the multipliers of a context of several strips (one strip has its spare in lam_c).
int lam_spares(admm_ctx *c, long long nr, long long nc) {
  if (c->lamf_r) return ADMM_OK;
  int rc;
  if ((rc = dev_alloc(c, &c->lamf_r, (size_t)nr + 2 * c->lam_slack)) || (rc = dev_alloc(c, &c->lamf_c, (size_t)nc + 2 * c->lam_slack)))
    return rc;
  HIP_TRY(c, hipMemsetAsync(c->lamf_r, 0, sizeof(float) * ((size_t)nr + 2 * c->lam_slack), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->lamf_c, 0, sizeof(float) * ((size_t)nc + 2 * c->lam_slack), c->stream));
  c->lamf_r += c->lam_slack;
  c->lamf_c += c->lam_slack;
  return ADMM_OK;
}

// The multipliers as binary32, for readers that take binary32 only.  One strip:
// the one array, widened into the buffer the next sweep would write (free
// between two sweeps).  Several strips: both arrays, into two buffers of their
// own that are allocated when first needed.
int lam16_widen(admm_ctx *c, const float **fr, const float **fc) {
  const long long nr = (long long)c->d.batch * c->st.nrows * c->d.cols * c->d.k;
  const long long nc = (long long)c->d.batch * c->d.rows * c->st.ncols * c->d.k;
  float *dr = c->lam_c, *dc = nullptr;
  if (!c->one_strip()) {
    const int rc = lam_spares(c, nr, nc);
    if (rc) return rc;
    dr = c->lamf_r;
    dc = c->lamf_c;
  }
  launch_widen_f32(c, dr, c->lam_r, nr);
  if (dc) launch_widen_f32(c, dc, c->lam_c, nc);
  HIP_TRY(c, hipGetLastError());
  *fr = dr;
  *fc = dc ? dc : dr;
  return ADMM_OK;
}

// int8 counts -> int16 counts, for the rest of the run.
int lam8_to_16(admm_ctx *c) {
  const long long nr = (long long)c->d.batch * c->st.nrows * c->d.cols * c->d.k;
  const long long nc = (long long)c->d.batch * c->d.rows * c->st.ncols * c->d.k;
  if (c->one_strip()) {
    // into the buffer the next sweep would write, which then becomes the array
    if (nr > 0)
      hipLaunchKernelGGL(widen_counts_kernel, dim3(grid_for(nr, 256)), dim3(256), 0, c->stream,
                         reinterpret_cast<int16_t *>(c->lam_c), reinterpret_cast<const int8_t *>(c->lam_r), nr);
    std::swap(c->lam_r, c->lam_c);
  } else {
    const int rc = lam_spares(c, nr, nc);
    if (rc) return rc;
    if (nr > 0)
      hipLaunchKernelGGL(widen_counts_kernel, dim3(grid_for(nr, 256)), dim3(256), 0, c->stream,
                         reinterpret_cast<int16_t *>(c->lamf_r), reinterpret_cast<const int8_t *>(c->lam_r), nr);
    if (nc > 0)
      hipLaunchKernelGGL(widen_counts_kernel, dim3(grid_for(nc, 256)), dim3(256), 0, c->stream,
                         reinterpret_cast<int16_t *>(c->lamf_c), reinterpret_cast<const int8_t *>(c->lam_c), nc);
    // the old arrays become the spares
    std::swap(c->lam_r, c->lamf_r);
    std::swap(c->lam_c, c->lamf_c);
  }
  HIP_TRY(c, hipGetLastError());
  c->lam8 = false;
  for (auto &sc : c->scans) sc.pending = false;
  return ADMM_OK;
}

// Queues a scan of the int8 counts as they stand after `after` sweeps.
int lam8_queue_scan(admm_ctx *c, int after) {
  admm_ctx::CountScan &sc = c->scans[c->scan_next];
  const int slot = c->scan_next;
  c->scan_next = (c->scan_next + 1) % admm_ctx::kScans;
  if (sc.pending) {  // the oldest scan still unread: read it first (it completed long ago)
    HIP_TRY(c, hipEventSynchronize(sc.ev));
    if (c->h_scan_flag[slot]) return lam8_to_16(c);
    c->lam8_clean = std::max(c->lam8_clean, sc.at);
    sc.pending = false;
  }
  if (!sc.ev) HIP_TRY(c, hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming));
  const long long nr = (long long)c->d.batch * c->st.nrows * c->d.cols * c->d.k;
  const long long nc = (long long)c->d.batch * c->d.rows * c->st.ncols * c->d.k;
  HIP_TRY(c, hipMemsetAsync(c->d_scan_flag + slot, 0, sizeof(int), c->stream));
  if (nr > 0)
    hipLaunchKernelGGL(scan_counts_kernel, dim3(grid_for(nr / 16 + 1, 256)), dim3(256), 0, c->stream,
                       reinterpret_cast<const int8_t *>(c->lam_r), nr, c->lam8_thr, c->d_scan_flag + slot);
  if (!c->lam_shared() && nc > 0)
    hipLaunchKernelGGL(scan_counts_kernel, dim3(grid_for(nc / 16 + 1, 256)), dim3(256), 0, c->stream,
                       reinterpret_cast<const int8_t *>(c->lam_c), nc, c->lam8_thr, c->d_scan_flag + slot);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->h_scan_flag + slot, c->d_scan_flag + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(sc.ev, c->stream));
  sc.at = after;
  sc.pending = true;
  return ADMM_OK;
}

// Before sweep number `s` (0-based) is queued on int8 counts: make sure a clean scan covers it.
int lam8_cover(admm_ctx *c, int s) {
  const int reach = 126 - c->lam8_thr;  // sweeps a clean scan covers
  // results that have arrived by now, oldest first
  for (int pass = 0; pass < 2 && c->lam8; pass++) {
    const bool must = pass == 1;
    if (must && s - c->lam8_clean <= reach) break;
    for (int i = 0; i < admm_ctx::kScans && c->lam8; i++) {
      const int slot = (c->scan_next + i) % admm_ctx::kScans;
      admm_ctx::CountScan &sc = c->scans[slot];
      if (!sc.pending) continue;
      if (must) {
        if (s - c->lam8_clean <= reach) break;
        HIP_TRY(c, hipEventSynchronize(sc.ev));
      } else if (hipEventQuery(sc.ev) != hipSuccess) {
        break;  // the later ones have not completed either
      }
      if (c->h_scan_flag[slot]) return lam8_to_16(c);
      c->lam8_clean = std::max(c->lam8_clean, sc.at);
      sc.pending = false;
    }
  }
  if (c->lam8 && s - c->lam8_clean > reach) {
    // nothing pending reaches far enough: scan now and wait for it
    int rc = lam8_queue_scan(c, s);
    if (rc || !c->lam8) return rc;
    return lam8_cover(c, s);
  }
  return ADMM_OK;
}

// After a sweep that was queued without error: its outputs become the current ones.
void sweep_outputs_current(admm_ctx *c, const ChainArgs &a) {
  std::swap(c->d_chain_lin, c->d_chain_lin_alt);
  std::swap(c->d_chain_unary, c->d_chain_unary_alt);
  if (c->direct_labels) {
    c->xrow_r = a.seg[0].x;
    c->xcolT_c = a.seg[1].x;
  }
}

int chain_sweep(admm_ctx *c, bool dual) {
  if (!dual && c->lam8) {
    const int rc = lam8_cover(c, c->lam16_sweeps);
    if (rc) return rc;
  }
  if (!dual && c->lam16 && c->lam16_sweeps >= c->lam16_limit) {
    if (c->lam8) {
      const int rc8 = lam8_to_16(c);
      if (rc8) return rc8;
    }
    // a count could leave int16 from here on: binary32 for the rest of the run
    const float *fr = nullptr, *fc = nullptr;
    const int rc = lam16_widen(c, &fr, &fc);
    if (rc) return rc;
    if (c->one_strip()) {
      std::swap(c->lam_r, c->lam_c);
    } else {
      c->lam_r = c->lamf_r;
      c->lam_c = c->lamf_c;
      c->lamf_r = c->lamf_c = nullptr;  // now the arrays themselves; a later copy gets buffers of its own
    }
    c->lam16 = false;
  }
  const int rc = chain_sweep_launch(c, dual);
  // the row chains wrote the updated multipliers into the other buffer
  if (!rc && !dual && c->lam_shared()) std::swap(c->lam_r, c->lam_c);
  if (!rc && !dual && c->lam16) c->lam16_sweeps++;
  if (!rc && !dual && c->lam8 && c->lam16_sweeps % c->lam8_every == 0) return lam8_queue_scan(c, c->lam16_sweeps);
  return rc;
}

int chain_sweep_launch(admm_ctx *c, bool dual) {
  // The queued objective sums move into this launch's arguments; if the launch then fails
  // they go back on the queue, so that a later read still sums them.
  const std::vector<admm_ctx::RedJob> red_q_before = c->red_q;
  const int swept_before = c->swept_slot;
  bool moved_q = false;
  auto give_back = [&](int rc) {
    if (rc && moved_q) { c->red_q = red_q_before; c->swept_slot = swept_before; }
    return rc;
  };
  ChainArgs a{};
  a.nseg = 2;
  fill_seg_rows(c, a.seg[0], dual);
  fill_seg_cols(c, a.seg[1], dual);
  a.k = c->d.k;
  a.penalty = c->d.penalty;
  a.rho = c->d.rho;
  a.inv_k = c->inv_k;
  a.Wt = dual ? c->d_Wsplit : c->d_Wt;
  a.Wsplit = c->d_Wsplit;
  a.WtT = dual ? c->d_WsplitT : c->d_WtT;
  a.WsplitT = c->d_WsplitT;
  SweepShape sh = pick_shape(c->KP, (long long)a.seg[0].nchains + a.seg[1].nchains);
  if (!dual) {
    // this sweep's per-half-chain terms go where the last sweep's are not
    const long long off = 2LL * c->d.batch * c->st.nrows;
    a.seg[0].chain_lin = c->d_chain_lin_alt; a.seg[0].chain_unary = c->d_chain_unary_alt;
    a.seg[1].chain_lin = c->d_chain_lin_alt + off; a.seg[1].chain_unary = c->d_chain_unary_alt + off;
    if (c->direct_labels) {
      // the crossing chains' labels as of the last admm_update, read where those chains wrote them
      ChainSeg &r = a.seg[0], &cc = a.seg[1];
      r.x = c->xrow_r == c->xrow_pub ? c->xrow_buf[c->xrow_r == c->xrow_buf[0] ? 1 : 0] : c->xrow_r;
      cc.x = c->xcolT_c == c->xcolT_pub ? c->xcolT_buf[c->xcolT_c == c->xcolT_buf[0] ? 1 : 0] : c->xcolT_c;
      r.xo = c->xcolT_pub; r.xo_cstride = 1; r.xo_sstride = r.cpg;
      cc.xo = c->xrow_pub; cc.xo_cstride = 1; cc.xo_sstride = cc.cpg;
    }
    // queued sums over the last sweep's terms: one more workgroup of this launch
    if ((long long)a.seg[0].nchains + a.seg[1].nchains == 0) {
      const int rcq = flush_red(c);
      if (rcq) return rcq;
    }
    if (c->swept_slot && c->red_q.size() < 2 && red_deferred(c)) {
      c->red_q.push_back({c->d_obj_slots + 2 * (size_t)c->swept_slot, c->d_chain_unary, chain_terms(c)});
      c->swept_slot = 0;
    }
    // (unary terms of an earlier sweep that nobody has summed and that cannot ride along: now)
    if (c->swept_slot) {
      const int rcs = settle_swept_now(c);
      if (rcs) return rcs;
    }
    a.red_jobs = (int)c->red_q.size();
    for (int j = 0; j < a.red_jobs; j++) { a.red_dst[j] = c->red_q[j].dst; a.red_part[j] = c->red_q[j].part; a.red_n[j] = c->red_q[j].n; }
    c->red_q.clear();
    moved_q = true;
  }
  if (dual && c->lam16) {
    // the dual evaluation reads binary32 multipliers
    const float *fr = nullptr, *fc = nullptr;
    const int rc = lam16_widen(c, &fr, &fc);
    if (rc) return give_back(rc);
    a.seg[0].lam = fr;
    a.seg[1].lam = fc;
  }
  if (!dual && getenv("ADMM_SWEEP_STAMPS")) {
    // diagnostic: per-wavefront clock stamps (start, end of forward pass, end), summarised on stderr
    const size_t nw = 2 * ((size_t)a.seg[0].nchains + a.seg[1].nchains) + 16;
    unsigned long long *d_st = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_st, nw * 4 * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemset(d_st, 0, nw * 4 * sizeof(unsigned long long)));
    a.stamps = d_st;
    int rc = launch_sweep<false>(c, a, sh);
    if (rc) return give_back(rc);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> h(nw * 4);
    HIP_TRY(c, hipMemcpy(h.data(), d_st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_st);
    std::vector<double> fw, bt;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t w = 0; w < (size_t)a.wave_begin[a.nseg]; w++) {
      if (!h[3 * w + 2]) continue;
      fw.push_back((double)(h[3 * w + 1] - h[3 * w]));
      bt.push_back((double)(h[3 * w + 2] - h[3 * w + 1]));
      t0 = std::min(t0, h[3 * w]); t1 = std::max(t1, h[3 * w + 2]);
    }
    std::sort(fw.begin(), fw.end()); std::sort(bt.begin(), bt.end());
    if (!fw.empty())
      fprintf(stderr, "sweep stamps: %zu waves, forward median %.0f max %.0f, backtrack median %.0f max %.0f, span %.0f (clock ticks)\n",
              fw.size(), fw[fw.size() / 2], fw.back(), bt[bt.size() / 2], bt.back(), (double)(t1 - t0));
    // how long the wavefronts of each set of chains ran, in deciles (the counters of
    // different dies have different origins: only differences inside a wavefront mean anything)
    for (int si = 0; si < a.nseg; si++) {
      std::vector<double> du;
      for (int w = a.wave_begin[si]; w < a.wave_begin[si + 1]; w++)
        if (h[3 * (size_t)w + 2]) du.push_back((double)(h[3 * (size_t)w + 2] - h[3 * (size_t)w]));
      std::sort(du.begin(), du.end());
      if (du.empty()) continue;
      fprintf(stderr, "  set %d: %zu waves, duration deciles", si, du.size());
      for (int q = 0; q <= 10; q++) fprintf(stderr, " %.0f", du[std::min(du.size() - 1, du.size() * q / 10)]);
      fprintf(stderr, "\n");
    }
    // mean duration by die, by shader engine, by compute unit of a shader engine, by half (left / right)
    {
      const size_t W = (size_t)a.wave_begin[a.nseg];
      double sx[16] = {0}, ss[8] = {0}, sc[16] = {0}, sr[2] = {0};
      int nx[16] = {0}, ns[8] = {0}, nc[16] = {0}, nr[2] = {0};
      for (size_t w = 0; w < W; w++) {
        if (!h[3 * w + 2]) continue;
        const double du = (double)(h[3 * w + 2] - h[3 * w]);
        const unsigned long long id = h[3 * W + w];
        const int xcc = (int)(id >> 32) & 15, hw = (int)(id & 0xFFFFFFFFu);
        const int cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
        sx[xcc] += du; nx[xcc]++; ss[se] += du; ns[se]++; sc[cu] += du; nc[cu]++;
        sr[w & 1] += du; nr[w & 1]++;
      }
      fprintf(stderr, "  by die:");
      for (int i = 0; i < 16; i++) if (nx[i]) fprintf(stderr, " %d:%.0f(%d)", i, sx[i] / nx[i], nx[i]);
      fprintf(stderr, "\n  by shader engine:");
      for (int i = 0; i < 8; i++) if (ns[i]) fprintf(stderr, " %d:%.0f(%d)", i, ss[i] / ns[i], ns[i]);
      fprintf(stderr, "\n  by compute unit id:");
      for (int i = 0; i < 16; i++) if (nc[i]) fprintf(stderr, " %d:%.0f(%d)", i, sc[i] / nc[i], nc[i]);
      fprintf(stderr, "\n  left halves %.0f, right halves %.0f\n", nr[0] ? sr[0] / nr[0] : 0.0, nr[1] ? sr[1] / nr[1] : 0.0);
      // by position of the chains in their set (16 bins), with the mean start offset inside the die
      unsigned long long first_start[16];
      for (int i = 0; i < 16; i++) first_start[i] = ~0ull;
      for (size_t w = 0; w < W; w++)
        if (h[3 * w + 2]) { const int x = (int)(h[3 * W + w] >> 32) & 15; first_start[x] = std::min(first_start[x], h[3 * w]); }
      for (int si = 0; si < a.nseg; si++) {
        const int w0 = a.wave_begin[si], w1 = a.wave_begin[si + 1];
        fprintf(stderr, "  set %d by chain position, duration(start offset):", si);
        for (int bin = 0; bin < 16; bin++) {
          double sd = 0, so = 0; int n = 0;
          for (int w = w0 + (int)((long long)(w1 - w0) * bin / 16); w < w0 + (int)((long long)(w1 - w0) * (bin + 1) / 16); w++) {
            if (!h[3 * (size_t)w + 2]) continue;
            const int x = (int)(h[3 * W + w] >> 32) & 15;
            sd += (double)(h[3 * (size_t)w + 2] - h[3 * (size_t)w]); so += (double)(h[3 * (size_t)w] - first_start[x]); n++;
          }
          if (n) fprintf(stderr, " %.0f(%.0f)", sd / n, so / n);
        }
        fprintf(stderr, "\n");
      }
    }
    sweep_outputs_current(c, a);
    return ADMM_OK;
  }
  int rc;
  if (getenv("ADMM_SWEEP_SPLIT")) {  // diagnostic: one launch per set of chains
    ChainArgs b = a;
    b.nseg = 1;
    b.seg[0] = b.seg[1];
    b.red_jobs = 0;  // (the queued sums went with the first launch)
    ChainArgs a1 = a;
    a1.nseg = 1;
    rc = dual ? launch_sweep<true>(c, a1, sh) : launch_sweep<false>(c, a1, sh);
    if (!rc) rc = dual ? launch_sweep<true>(c, b, sh) : launch_sweep<false>(c, b, sh);
  } else {
    rc = dual ? launch_sweep<true>(c, a, sh) : launch_sweep<false>(c, a, sh);
  }
  if (!rc && !dual) sweep_outputs_current(c, a);
  return give_back(rc);
}

template <typename TB>
int chain_consensus(admm_ctx *c, float *m, const label_t *xa, const TB *xbT, int nr, int nc) {
  if (nr <= 0 || nc <= 0) return ADMM_OK;
  const int tr = (nr + 31) / 32, tc = (nc + 31) / 32;
  const long long blocks = (long long)c->d.batch * tr * tc;
  hipLaunchKernelGGL(chain_consensus_kernel<TB>, dim3((unsigned)blocks), dim3(256), 0, c->stream, m, xa,
                     xbT, nr, nc, c->d.k, c->inv_k, tc, tr * tc);
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

// Both transpositions of Update in one launch: out0 = in0 (nr0 x nc0 per
// grid) transposed, out1 = in1 (nr1 x nc1) transposed.
template <typename TI>
int transpose_labels2(admm_ctx *c, label_t *out0, const TI *in0, int nr0, int nc0, label_t *out1, const TI *in1, int nr1, int nc1) {
  TransposeArgs a{};
  int blocks = 0;
  auto add = [&](label_t *out, const TI *in, int nr, int nc) {
    if (nr <= 0 || nc <= 0) return;
    TransposeJob &j = a.job[a.njobs++];
    const int tr = (nr + 63) / 64, tc = (nc + 63) / 64;
    j.out = out; j.in = in; j.nr = nr; j.nc = nc; j.tiles_c = tc; j.tiles_per_grid = tr * tc;
    j.block_begin = blocks;
    blocks += c->d.batch * tr * tc;
  };
  add(out0, in0, nr0, nc0);
  add(out1, in1, nr1, nc1);
  if (blocks == 0) return ADMM_OK;
  hipLaunchKernelGGL(transpose_labels_kernel<TI>, dim3((unsigned)blocks), dim3(256), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

// ------------------------------------------------------------------ edge helpers

// EDGE partition on several ranks.  On entry h_edges / V / E describe the whole
// graph; on exit they describe this rank's slice (Grid.cpp:446-454) with its
// vertices renumbered in ascending global order (GetVerts, Grid.cpp:815-856),
// and the neighbour plan of InitNeighborsByEdge (Grid.cpp:198-331) is in place.
// Every rank holds the whole edge list, so the plan needs no communication.
void edge_partition(admm_ctx *c) {
  const int P = c->d.nranks, me = c->d.rank, E = c->Eglobal;
  auto slice = [&](int r, int *off, int *n) { admm_partition(E, P, r, off, n); };  // same split as Grid.cpp:446-454
  // sorted distinct vertices of a rank's slice with their local copy counts
  auto verts_of = [&](int r, std::vector<int> &vs, std::vector<int> &cn) {
    int off, n;
    slice(r, &off, &n);
    std::vector<int> v(c->h_edges.begin() + 2 * (size_t)off, c->h_edges.begin() + 2 * ((size_t)off + n));
    std::sort(v.begin(), v.end());
    vs.clear(); cn.clear();
    for (int x : v) {
      if (!vs.empty() && vs.back() == x) cn.back()++;
      else { vs.push_back(x); cn.push_back(1); }
    }
  };
  std::vector<int> mine, mycnt;
  verts_of(me, mine, mycnt);
  c->h_lverts = mine;
  c->h_cnt = mycnt;
  c->nb_rank.clear(); c->nb_off.assign(1, 0); c->h_send_idx.clear(); c->h_nb_cnt.clear();
  if (!mine.empty()) {
    for (int r = 0; r < P; r++) {
      if (r == me) continue;
      int off, n;
      slice(r, &off, &n);
      if (n == 0) continue;
      // a neighbour is a rank whose vertex range overlaps mine (Grid.cpp:240-251)
      int lo = c->h_edges[2 * (size_t)off], hi = lo;
      for (size_t i = 2 * (size_t)off; i < 2 * ((size_t)off + n); i++) { lo = std::min(lo, c->h_edges[i]); hi = std::max(hi, c->h_edges[i]); }
      if (lo > mine.back() || mine.front() > hi) continue;
      std::vector<int> theirs, theircnt;
      verts_of(r, theirs, theircnt);
      // shared vertices, ascending (Grid.cpp:264-279), and the copies the neighbour holds (Grid.cpp:288-319)
      size_t a = 0, b = 0;
      while (a < mine.size() && b < theirs.size()) {
        if (mine[a] == theirs[b]) {
          c->h_send_idx.push_back((int)a);
          c->h_nb_cnt.push_back(theircnt[b]);
          c->h_cnt[a] += theircnt[b];
          a++; b++;
        } else if (mine[a] < theirs[b]) a++; else b++;
      }
      c->nb_rank.push_back(r);
      c->nb_off.push_back((int)c->h_send_idx.size());
    }
  }
  // my edges with local vertex ids
  int off, n;
  slice(me, &off, &n);
  std::vector<int> local(std::max(1, 2 * n));
  for (int i = 0; i < 2 * n; i++)
    local[i] = (int)(std::lower_bound(mine.begin(), mine.end(), c->h_edges[2 * (size_t)off + i]) - mine.begin());
  c->h_edges.swap(local);
  c->e0 = off;
  c->E = n;
  c->V = (int)mine.size();
}

void build_incidence(admm_ctx *c) {
  // Copies of every vertex in edge-list order: GetVerts' count (Grid.cpp:815-856)
  // plus where the copies are, so the consensus can add them in that order.
  c->h_inc_off.assign(c->V + 1, 0);
  for (int i = 0; i < 2 * c->E; i++) c->h_inc_off[c->h_edges[i] + 1]++;
  for (int v = 0; v < c->V; v++) c->h_inc_off[v + 1] += c->h_inc_off[v];
  c->h_inc.assign(std::max(1, 2 * c->E), 0);
  std::vector<int> fillp(c->h_inc_off.begin(), c->h_inc_off.end() - 1);
  for (int i = 0; i < 2 * c->E; i++) c->h_inc[fillp[c->h_edges[i]]++] = i;
}

int edge_blocks(int E, int KP) {
  // up to 64 labels: 64 / KP edges per wavefront; beyond: one edge per wavefront (edge_solve_big_kernel)
  const int gpb = KP <= 64 ? (kEdgeBlock / 64) * (64 / KP) : edge_big_epb(KP);
  return (E + gpb - 1) / gpb;
}

template <int KP>
void launch_edge_t(const EdgeArgs &a, hipStream_t s) {
  hipLaunchKernelGGL((edge_solve_kernel<KP>), dim3(edge_blocks(a.E, KP)), dim3(kEdgeBlock), 0, s, a);
}

int edge_compute(admm_ctx *c) {
  if (c->E == 0) return ADMM_OK;
  EdgeArgs a{};
  a.E = c->E; a.V = c->V; a.k = c->d.k;
  a.penalty = c->d.penalty; a.penaltyPrime = c->d.penalty_prime; a.rho = c->d.rho;
  a.edgeList = c->d_edges;
  a.mu_node = c->d_mu_node; a.pot_node = c->d_pot_node; a.lam_node = c->d_lam_node;
  a.mu_edge = c->d_mu_edge; a.pot_edge = c->d_pot_edge; a.m_vert = c->d_m_vert;
  a.obj_part = c->d_edge_obj;
  switch (c->KP) {
    case 2: launch_edge_t<2>(a, c->stream); break;
    case 4: launch_edge_t<4>(a, c->stream); break;
    case 8: launch_edge_t<8>(a, c->stream); break;
    case 16: launch_edge_t<16>(a, c->stream); break;
    case 32: launch_edge_t<32>(a, c->stream); break;
    case 64: launch_edge_t<64>(a, c->stream); break;
    // more labels than lanes: the table of an edge stays in HBM
    case 128: hipLaunchKernelGGL((edge_solve_big_kernel<128>), dim3(edge_blocks(a.E, 128)), dim3(64 * edge_big_epb(128)), 0, c->stream, a); break;
    case 256: hipLaunchKernelGGL((edge_solve_big_kernel<256>), dim3(edge_blocks(a.E, 256)), dim3(64 * edge_big_epb(256)), 0, c->stream, a); break;
    case 512: hipLaunchKernelGGL((edge_solve_big_kernel<512>), dim3(edge_blocks(a.E, 512)), dim3(64 * edge_big_epb(512)), 0, c->stream, a); break;
    case 1024: hipLaunchKernelGGL((edge_solve_big_kernel<1024>), dim3(edge_blocks(a.E, 1024)), dim3(64 * edge_big_epb(1024)), 0, c->stream, a); break;
    default: return fail(c, ADMM_E_ARG, "edge solver supports k <= %d (k=%d)", kEdgeBigMaxK, c->d.k);
  }
  HIP_TRY(c, hipGetLastError());
  c->edge_obj_n = c->d_edge_obj ? edge_blocks(c->E, c->KP) : 0;
  c->launch_info = c->d_edge_obj ? "objective_summed_by_the_solve=1" : "objective_summed_by_the_solve=0";
  return ADMM_OK;
}

int check_ctx(admm_ctx *c) {
  if (!c) return ADMM_E_ARG;
  hipError_t e = hipSetDevice(c->d.device);
  if (e != hipSuccess) return fail(c, ADMM_E_HIP, "hipSetDevice(%d): %s", c->d.device, hipGetErrorString(e));
  return ADMM_OK;
}

}  // namespace

// ====================================================================== C ABI

static int flush_pending(admm_ctx *c);
static int settle_swept(admm_ctx *c);

extern "C" {

int admm_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

const char *admm_version(void) { return "admm_hip 0.1 (gfx950)"; }

int admm_edge_log_values(int device, const double *x, double *y, int64_t n) {
  if (!x || !y || n < 0) return ADMM_E_ARG;
  if (n == 0) return ADMM_OK;
  if (hipSetDevice(device) != hipSuccess) return ADMM_E_HIP;
  double *dx = nullptr, *dy = nullptr;
  int rc = ADMM_E_HIP;
  if (hipMalloc((void **)&dx, sizeof(double) * (size_t)n) == hipSuccess &&
      hipMalloc((void **)&dy, sizeof(double) * (size_t)n) == hipSuccess &&
      hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(edge_log_values_kernel, dim3(grid_for(n, kEdgeBlock)), dim3(kEdgeBlock), 0, 0, dy, dx, (long long)n);
    if (hipGetLastError() == hipSuccess && hipMemcpy(y, dy, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess)
      rc = ADMM_OK;
  }
  (void)hipFree(dx);
  (void)hipFree(dy);
  return rc;
}

int admm_edge_exp_values(int device, const float *x, double *own, double *lib, unsigned char *was_nan, int64_t n) {
  if (!x || !own || !lib || !was_nan || n < 0) return fail(nullptr, ADMM_E_ARG, "bad argument");
  if (n == 0) return ADMM_OK;
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, ADMM_E_HIP, "hipSetDevice(%d) failed", device);
  float *dx = nullptr;
  double *d1 = nullptr, *d2 = nullptr;
  unsigned char *dn = nullptr;
  int rc = fail(nullptr, ADMM_E_HIP, "admm_edge_exp_values: a HIP call failed");
  if (hipMalloc((void **)&dx, sizeof(float) * (size_t)n) == hipSuccess && hipMalloc((void **)&d1, sizeof(double) * (size_t)n) == hipSuccess &&
      hipMalloc((void **)&d2, sizeof(double) * (size_t)n) == hipSuccess && hipMalloc((void **)&dn, (size_t)n) == hipSuccess &&
      hipMemcpy(dx, x, sizeof(float) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(edge_exp_values_kernel, dim3(grid_for(n, kEdgeBlock)), dim3(kEdgeBlock), 0, 0, d1, d2, dn, dx, (long long)n);
    if (hipGetLastError() == hipSuccess && hipMemcpy(own, d1, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(lib, d2, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(was_nan, dn, (size_t)n, hipMemcpyDeviceToHost) == hipSuccess)
      rc = ADMM_OK;
  }
  (void)hipFree(dx); (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dn);
  return rc;
}

const char *admm_last_error(const admm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void admm_partition(int n, int nranks, int rank, int *start, int *count) {
  // Grid::PartitionByRow, Grid.cpp:377-390
  const int base = n / nranks, extra = n % nranks;
  int cnt, off;
  if (rank < extra) { cnt = base + 1; off = rank * cnt; }
  else { cnt = base; off = extra * (base + 1) + (rank - extra) * base; }
  if (start) *start = off;
  if (count) *count = cnt;
}

uint32_t admm_lcg_fill(uint32_t state, int64_t n, float *out) {
  for (int64_t i = 0; i < n; i++) {
    state = state * 1664525u + 1013904223u;
    out[i] = (float)(state >> 8) / 16777216.0f;
  }
  return state;
}

int admm_grid_edges(int rows, int cols, int32_t *el) {
  int e = 0;
  for (int r = 0; r < rows; r++) {
    for (int cidx = 0; cidx + 1 < cols; cidx++) { el[2 * e] = r * cols + cidx; el[2 * e + 1] = r * cols + cidx + 1; e++; }
    if (r == rows - 1) break;
    for (int cidx = 0; cidx < cols; cidx++) { el[2 * e] = r * cols + cidx; el[2 * e + 1] = (r + 1) * cols + cidx; e++; }
  }
  return e;
}

int admm_create(const admm_desc *desc, admm_ctx **out) {
  if (!desc || !out) return fail(nullptr, ADMM_E_ARG, "null argument");
  *out = nullptr;
  const admm_desc &d = *desc;
  if (d.k < 2) return fail(nullptr, ADMM_E_ARG, "k must be >= 2 (k=%d)", d.k);
  if (d.solver == ADMM_SOLVER_CHAIN_MINSUM && d.k > 64)
    return fail(nullptr, ADMM_E_ARG, "the chain solver holds a label vector across the lanes of one wavefront: k must be <= 64 (k=%d)", d.k);
  if (d.k > kEdgeBigMaxK) return fail(nullptr, ADMM_E_ARG, "k must be <= %d (k=%d)", kEdgeBigMaxK, d.k);
  if (d.batch < 1) return fail(nullptr, ADMM_E_ARG, "batch must be >= 1");
  if (d.nranks < 1 || d.rank < 0 || d.rank >= d.nranks) return fail(nullptr, ADMM_E_ARG, "bad rank %d of %d", d.rank, d.nranks);
  if (!(d.penalty > 0.f) || !(d.rho > 0.f)) return fail(nullptr, ADMM_E_ARG, "penalty and rho must be positive");
  if (d.solver != ADMM_SOLVER_EDGE_BETHE && d.solver != ADMM_SOLVER_CHAIN_MINSUM)
    return fail(nullptr, ADMM_E_ARG, "unknown solver %d", d.solver);
  const bool graph = d.edge_list != nullptr;
  if (!graph && (d.rows < 1 || d.cols < 1)) return fail(nullptr, ADMM_E_ARG, "rows and cols must be >= 1");
  if (!graph && (long long)d.rows * d.cols * d.batch > (1LL << 31) - 1)
    return fail(nullptr, ADMM_E_ARG, "more than 2^31-1 vertices");
  if (d.solver == ADMM_SOLVER_CHAIN_MINSUM) {
    if (graph) return fail(nullptr, ADMM_E_ARG, "the chain solver needs a grid, not an edge list");
    // the sweep reaches a half chain's operands through a buffer descriptor:
    // the bytes one wavefront spans (half a chain plus prefetch overrun, and
    // the distance between the chains that share a wavefront) must stay below 4 GiB
    const long long span_col = (((long long)d.rows + 1) / 2 + 130) * d.cols * d.k + 64LL * d.k;
    const long long span_row = (((long long)d.cols + 1) / 2 + 130) * d.k + 64LL * d.cols * d.k;
    if (span_col >= (1LL << 30) || span_row >= (1LL << 30))
      return fail(nullptr, ADMM_E_ARG, "grid too large for the sweep's 32-bit offsets: (rows / 2 + 130) * cols * k must stay below 2^30");
    // a batch: the chains that share a wavefront may belong to two consecutive
    // grids, so the lane offsets also span one whole grid
    const long long grid_el = (long long)d.rows * d.cols * d.k;
    if (d.batch > 1 && (span_col + grid_el >= (1LL << 30) || span_row + grid_el >= (1LL << 30)))
      return fail(nullptr, ADMM_E_ARG, "grids of a batch too large for the sweep's 32-bit offsets: rows * cols * k plus (rows / 2 + 130) * cols * k must stay below 2^30");
    if (d.nranks > 1 && d.batch != 1) return fail(nullptr, ADMM_E_ARG, "strips (nranks > 1) need batch == 1; shard a batch by grids instead");
    if (d.exchange != ADMM_EXCHANGE_LABELS && d.exchange != ADMM_EXCHANGE_DENSE)
      return fail(nullptr, ADMM_E_ARG, "unknown exchange mode %d", d.exchange);
  } else {
    if (d.batch != 1) return fail(nullptr, ADMM_E_ARG, "the edge solver takes batch == 1");
    if (!(d.penalty_prime > 0.f)) return fail(nullptr, ADMM_E_ARG, "penalty_prime must be positive");
    if (graph && (d.num_vertices < 1 || d.num_edges < 0)) return fail(nullptr, ADMM_E_ARG, "bad graph sizes");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, ADMM_E_HIP, "no HIP device available");
  if (d.device < 0 || d.device >= ndev) return fail(nullptr, ADMM_E_ARG, "device %d out of range (%d devices)", d.device, ndev);

  admm_ctx *c = new admm_ctx;
  c->d = d;
  c->d.edge_list = nullptr;
  c->KP = next_pow2(d.k);
  c->inv_k = (float)(1.0 / (double)(float)d.k);  // Tree.cpp:25
  int rc = ADMM_OK;
  auto bail = [&](int code) { g_create_error = c->err; admm_destroy(c); return code; };
#define HIP_TRY_BAIL(call)                                                       \
  do {                                                                           \
    hipError_t e__ = (call);                                                     \
    if (e__ != hipSuccess) {                                                     \
      c->err = std::string(#call " failed: ") + hipGetErrorString(e__);          \
      return bail(ADMM_E_HIP);                                                   \
    }                                                                            \
  } while (0)
  if (hipSetDevice(d.device) != hipSuccess) { c->err = "hipSetDevice failed"; return bail(ADMM_E_HIP); }
  (void)scratch_for(d.device, nullptr);  // (a failure only disables admm_download_labels_static until a later query succeeds)
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(ADMM_E_HIP); }

  if (d.solver == ADMM_SOLVER_EDGE_BETHE) {
    if (graph) {
      c->V = d.num_vertices; c->E = d.num_edges;
      c->h_edges.assign(d.edge_list, d.edge_list + 2 * (size_t)c->E);
      for (int v : c->h_edges) if (v < 0 || v >= c->V) { c->err = "edge list names a vertex outside [0, num_vertices)"; return bail(ADMM_E_ARG); }
    } else {
      c->V = d.rows * d.cols;
      c->E = d.rows * (d.cols - 1) + d.cols * (d.rows - 1);
      c->h_edges.resize(std::max(1, 2 * c->E));
      admm_grid_edges(d.rows, d.cols, c->h_edges.data());
    }
    c->Vglobal = c->V; c->Eglobal = c->E;
    if (d.nranks > 1) edge_partition(c);  // my slice of the edges, my vertices, my neighbours
    build_incidence(c);
    const size_t E = c->E, V = c->V, k = d.k, k2 = k * k;
    if ((rc = dev_alloc(c, &c->d_edges, 2 * E)) || (rc = dev_alloc(c, &c->d_inc_off, V + 1)) ||
        (rc = dev_alloc(c, &c->d_inc, 2 * E)) || (rc = dev_alloc(c, &c->d_vert_pot, V * k)) ||
        (rc = dev_alloc(c, &c->d_mu_node, E * 2 * k)) || (rc = dev_alloc(c, &c->d_pot_node, E * 2 * k)) ||
        (rc = dev_alloc(c, &c->d_lam_node, E * 2 * k)) || (rc = dev_alloc(c, &c->d_mu_edge, E * k2)) ||
        (rc = dev_alloc(c, &c->d_pot_edge, E * k2)) || (rc = dev_alloc(c, &c->d_m_vert, V * k)))
      return bail(rc);
    if (d.nranks > 1) {
      const size_t S = c->h_send_idx.size();
      // per vertex, its entries of the receive buffer in ascending (= neighbour) order
      std::vector<int> rcv_off(V + 1, 0), rcv_pos(std::max<size_t>(S, 1), 0);
      for (int v : c->h_send_idx) rcv_off[v + 1]++;
      for (size_t v = 0; v < V; v++) rcv_off[v + 1] += rcv_off[v];
      std::vector<int> fillp(rcv_off.begin(), rcv_off.end() - 1);
      for (size_t p = 0; p < S; p++) rcv_pos[fillp[c->h_send_idx[p]]++] = (int)p;
      if ((rc = dev_alloc(c, &c->d_cnt, V)) || (rc = dev_alloc(c, &c->d_send_idx, S)) ||
          (rc = dev_alloc(c, &c->d_rcv_off, V + 1)) || (rc = dev_alloc(c, &c->d_rcv_pos, S)) ||
          (rc = dev_alloc(c, &c->d_m_part, V * k)) || (rc = dev_alloc(c, &c->d_sendv, S * k)) ||
          (rc = dev_alloc(c, &c->d_recvv, S * k)))
        return bail(rc);
      hipError_t f1 = h2d_sync(c, c->d_cnt, c->h_cnt.data(), sizeof(int) * V);
      hipError_t f2 = S ? h2d_sync(c, c->d_send_idx, c->h_send_idx.data(), sizeof(int) * S) : hipSuccess;
      hipError_t f3 = h2d_sync(c, c->d_rcv_off, rcv_off.data(), sizeof(int) * (V + 1));
      hipError_t f4 = S ? h2d_sync(c, c->d_rcv_pos, rcv_pos.data(), sizeof(int) * S) : hipSuccess;
      if (f1 != hipSuccess || f2 != hipSuccess || f3 != hipSuccess || f4 != hipSuccess) { c->err = "copy of the exchange plan failed"; return bail(ADMM_E_HIP); }
    }
    c->n_partial = 1024;
    if ((rc = dev_alloc(c, &c->d_partial, (size_t)c->n_partial))) return bail(rc);
    // ADMM_EDGE_FUSED_OBJ=0 (test knob): the solve sums nothing and the objective is always the
    // separate pass over the marginals
    const char *fo = getenv("ADMM_EDGE_FUSED_OBJ");
    c->edge_obj_n = 0;
    if (!(fo && atoi(fo) == 0) &&
        (rc = dev_alloc(c, &c->d_edge_obj, (size_t)edge_blocks(c->E, c->KP))))
      return bail(rc);
    hipError_t e1 = h2d_sync(c, c->d_edges, c->h_edges.data(), sizeof(int) * 2 * E);
    hipError_t e2 = h2d_sync(c, c->d_inc_off, c->h_inc_off.data(), sizeof(int) * (V + 1));
    hipError_t e3 = h2d_sync(c, c->d_inc, c->h_inc.data(), sizeof(int) * 2 * E);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { c->err = "copy of the edge list failed"; return bail(ADMM_E_HIP); }
  } else {
    admm_partition(d.rows, d.nranks, d.rank, &c->st.row0, &c->st.nrows);
    admm_partition(d.cols, d.nranks, d.rank, &c->st.col0, &c->st.ncols);
    const size_t k = d.k, B = d.batch;
    const size_t nrow_nodes = B * (size_t)c->st.nrows * d.cols;
    const size_t ncol_nodes = B * (size_t)d.rows * c->st.ncols;
    // back-pointer scratch: two half chains per chain, each padded to whole 64-step chunks
    const size_t Tp_r = 2 * ((((size_t)d.cols + 1) / 2 + kChunk - 1) / kChunk * kChunk);
    const size_t Tp_c = 2 * ((((size_t)d.rows + 1) / 2 + kChunk - 1) / kChunk * kChunk);
    // The sweep prefetches operands up to one unrolled block (<= 64 steps)
    // past a half chain's end without predication, forwards for a left half
    // and backwards for a right half; slack on both sides of every operand
    // array keeps those reads inside the allocation.  A column chain's step is
    // one strip row.
    const size_t slack_r = 64 * (size_t)std::max(d.cols, 1) * k;
    const size_t slack_c = 64 * (size_t)std::max(d.nranks == 1 ? d.cols : c->st.ncols, 1) * k;
    const size_t slack = std::max(slack_r, slack_c);
    c->lam_slack = slack;
    if ((rc = dev_alloc(c, &c->th_r, nrow_nodes * k + 2 * slack)) || (rc = dev_alloc(c, &c->lam_r, nrow_nodes * k + 2 * slack)) ||
        (rc = dev_alloc(c, &c->lam_c, ncol_nodes * k + 2 * slack)) ||
        (rc = dev_alloc(c, &c->xrow_r, nrow_nodes)) || (rc = dev_alloc(c, &c->xcolT_c, ncol_nodes)) ||
        (rc = dev_alloc(c, &c->xo_r, nrow_nodes)) || (rc = dev_alloc(c, &c->xo_c, ncol_nodes)) ||
        // (+ 16 bytes per step: the Potts kernel's layout is per wavefront, and its last
        // wavefront may hold a single chain of 8 labels)
        (rc = dev_alloc(c, &c->bp_r, B * c->st.nrows * Tp_r * c->KP + 24 * Tp_r + 64)) ||
        (rc = dev_alloc(c, &c->bp_c, B * c->st.ncols * Tp_c * c->KP + 24 * Tp_c + 64)) ||
        (rc = dev_alloc(c, &c->d_W, k * k)) || (rc = dev_alloc(c, &c->d_Wsplit, k * k)) ||
        (rc = dev_alloc(c, &c->d_Wt, k * k)) || (rc = dev_alloc(c, &c->d_WsplitT, k * k)) ||
        (rc = dev_alloc(c, &c->d_WtT, k * k)) ||
        (rc = dev_alloc(c, &c->d_chain_min, B * ((size_t)c->st.nrows + c->st.ncols))) ||
        (rc = dev_alloc(c, &c->d_chain_lin, 2 * B * ((size_t)c->st.nrows + c->st.ncols))) ||
        (rc = dev_alloc(c, &c->d_chain_unary, 2 * B * ((size_t)c->st.nrows + c->st.ncols))) ||
        (rc = dev_alloc(c, &c->d_chain_lin_alt, 2 * B * ((size_t)c->st.nrows + c->st.ncols))) ||
        (rc = dev_alloc(c, &c->d_chain_unary_alt, 2 * B * ((size_t)c->st.nrows + c->st.ncols))))
      return bail(rc);
    // (second label buffers: one strip may run with direct labels, which admm_init decides)
    c->xrow_buf[0] = c->xrow_r;
    c->xcolT_buf[0] = c->xcolT_c;
    if (d.nranks == 1 && ((rc = dev_alloc(c, &c->xrow_buf[1], nrow_nodes)) || (rc = dev_alloc(c, &c->xcolT_buf[1], ncol_nodes))))
      return bail(rc);
    // [slack | data | slack]: the pointers the rest of the code sees are the data's
    HIP_TRY_BAIL(hipMemsetAsync(c->th_r, 0, sizeof(float) * (nrow_nodes * k + 2 * slack), c->stream));
    HIP_TRY_BAIL(hipMemsetAsync(c->lam_r, 0, sizeof(float) * (nrow_nodes * k + 2 * slack), c->stream));
    HIP_TRY_BAIL(hipMemsetAsync(c->lam_c, 0, sizeof(float) * (ncol_nodes * k + 2 * slack), c->stream));
    c->th_r += slack; c->lam_r += slack; c->lam_c += slack;
    if (d.nranks == 1) {
      c->th_c = c->th_r;
    } else {
      if ((rc = dev_alloc(c, &c->th_c, ncol_nodes * k + 2 * slack)) ||
          (rc = dev_alloc(c, &c->xrow_c, ncol_nodes)) || (rc = dev_alloc(c, &c->xcolT_r, nrow_nodes)) ||
          (rc = dev_alloc(c, &c->xo_r_alt, nrow_nodes)) || (rc = dev_alloc(c, &c->xo_c_alt, ncol_nodes)) ||
          (rc = dev_alloc(c, &c->sendbuf, nrow_nodes + ncol_nodes)))
        return bail(rc);
      if (d.exchange == ADMM_EXCHANGE_DENSE &&
          ((rc = dev_alloc(c, &c->sendbuf_mu, (nrow_nodes + ncol_nodes) * k)) ||
           (rc = dev_alloc(c, &c->recvbuf_mu, (nrow_nodes + ncol_nodes) * k))))
        return bail(rc);
      HIP_TRY_BAIL(hipMemsetAsync(c->th_c, 0, sizeof(float) * (ncol_nodes * k + 2 * slack), c->stream));
      c->th_c += slack;
    }
    c->n_partial = 2048;
    if ((rc = dev_alloc(c, &c->d_partial, (size_t)c->n_partial * 2))) return bail(rc);
    c->n_obj_slots = 4096;
    if ((rc = dev_alloc(c, &c->d_obj_slots, (size_t)c->n_obj_slots * 2))) return bail(rc);
  }
  *out = c;
  return ADMM_OK;
}

int admm_destroy(admm_ctx *c) {
  if (!c) return ADMM_OK;
  (void)hipSetDevice(c->d.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
  if (c->ev_pulled) (void)hipEventDestroy(c->ev_pulled);
  for (auto &pr : c->prof_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto &sc : c->scans) if (sc.ev) (void)hipEventDestroy(sc.ev);
  if (c->h_scan_flag) (void)hipHostFree(c->h_scan_flag);
  for (void *p : c->allocs) (void)hipFree(p);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ADMM_OK;
}

int admm_edge_plan(const int32_t *edge_list, int num_edges, int nranks, int rank, int32_t *info,
                   int32_t *verts, int32_t *counts, int32_t *neigh_ranks, int32_t *neigh_off,
                   int32_t *shared_local, int32_t *shared_counts) {
  if (!info || num_edges < 0 || (!edge_list && num_edges > 0) || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(nullptr, ADMM_E_ARG, "bad argument");
  admm_ctx c;  // host fields only: no device is touched
  c.d.nranks = nranks;
  c.d.rank = rank;
  c.Eglobal = c.E = num_edges;
  c.h_edges.assign(edge_list, edge_list + 2 * (size_t)num_edges);
  edge_partition(&c);
  info[0] = c.e0; info[1] = c.E; info[2] = c.V; info[3] = (int)c.nb_rank.size(); info[4] = (int)c.h_send_idx.size();
  for (int v = 0; v < c.V; v++) {
    if (verts) verts[v] = c.h_lverts[v];
    if (counts) counts[v] = c.h_cnt[v];
  }
  for (size_t q = 0; q < c.nb_rank.size(); q++) if (neigh_ranks) neigh_ranks[q] = c.nb_rank[q];
  for (size_t q = 0; q < c.nb_off.size(); q++) if (neigh_off) neigh_off[q] = c.nb_off[q];
  for (size_t p = 0; p < c.h_send_idx.size(); p++) {
    if (shared_local) shared_local[p] = c.h_send_idx[p];
    if (shared_counts) shared_counts[p] = c.h_nb_cnt[p];
  }
  return ADMM_OK;
}

int admm_edge_local_info(const admm_ctx *c, int *edge0, int *nedges, int *nverts) {
  if (!c || is_chain(c)) return ADMM_E_ARG;
  if (edge0) *edge0 = c->e0;
  if (nedges) *nedges = c->E;
  if (nverts) *nverts = c->V;
  return ADMM_OK;
}

int admm_edge_local_vertices(const admm_ctx *c, int32_t *ids) {
  if (!c || is_chain(c) || !ids) return ADMM_E_ARG;
  for (int v = 0; v < c->V; v++) ids[v] = c->d.nranks > 1 ? c->h_lverts[v] : v;
  return ADMM_OK;
}

int admm_strip_info(const admm_ctx *c, int *row0, int *nrows, int *col0, int *ncols) {
  if (!c) return ADMM_E_ARG;
  if (row0) *row0 = c->st.row0;
  if (nrows) *nrows = c->st.nrows;
  if (col0) *col0 = c->st.col0;
  if (ncols) *ncols = c->st.ncols;
  return ADMM_OK;
}

int admm_upload_unaries(admm_ctx *c, const float *unary) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!unary) return fail(c, ADMM_E_ARG, "null unary array");
  if ((rc = flush_pending(c))) return rc;  // a queued objective still needs the old unaries
  const size_t k = c->d.k;
  if (!is_chain(c)) {
    if (c->d.nranks > 1) {
      // this rank's vertices, in its own numbering; kept on the host for admm_init
      c->h_unary_local.resize((size_t)c->V * k);
      for (int v = 0; v < c->V; v++)
        memcpy(c->h_unary_local.data() + (size_t)v * k, unary + (size_t)c->h_lverts[v] * k, sizeof(float) * k);
      unary = c->h_unary_local.data();
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_vert_pot, unary, sizeof(float) * (size_t)c->V * k, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    const admm_desc &d = c->d;
    const size_t Vg = (size_t)d.rows * d.cols;
    for (int g = 0; g < d.batch; g++) {
      // whole rows of the strip are contiguous in the caller's array
      const float *src = unary + (g * Vg + (size_t)c->st.row0 * d.cols) * k;
      float *dst = c->th_r + (size_t)g * c->st.nrows * d.cols * k;
      HIP_TRY(c, hipMemcpyAsync(dst, src, sizeof(float) * (size_t)c->st.nrows * d.cols * k, hipMemcpyHostToDevice, c->stream));
    }
    if (d.nranks > 1) {
      // column strip: a pitched copy, rows x (ncols*k floats) out of rows x (cols*k)
      HIP_TRY(c, hipMemcpy2DAsync(c->th_c, sizeof(float) * c->st.ncols * k,
                                  unary + (size_t)c->st.col0 * k, sizeof(float) * d.cols * k,
                                  sizeof(float) * c->st.ncols * k, d.rows, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->seen_r.assign((size_t)c->st.nrows, 1);
    c->seen_c.assign((size_t)d.rows, 1);
  }
  c->have_unary = true;
  c->theta_scaled = false;
  c->inited = false;
  return ADMM_OK;
}

// A share of the unaries of the chain solver.  side 0: whole grid rows of the
// row strip, [nrows][cols][k]; side 1: the column strip's part of grid rows,
// [nrows][strip cols][k].
static int upload_share(admm_ctx *c, int side, int row0, int nrows, const float *data) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!is_chain(c)) return fail(c, ADMM_E_STATE, "shares of the unaries are a chain-solver call");
  if (c->d.batch != 1) return fail(c, ADMM_E_ARG, "shares of the unaries need batch == 1");
  if (!data && nrows > 0) return fail(c, ADMM_E_ARG, "null unary array");
  const admm_desc &d = c->d;
  const int lo = side == 0 ? c->st.row0 : 0, hi = side == 0 ? c->st.row0 + c->st.nrows : d.rows;
  if (nrows < 0 || row0 < lo || (long long)row0 + nrows > hi)
    return fail(c, ADMM_E_ARG, "rows [%d, %d) lie outside this strip's rows [%d, %d)", row0, row0 + nrows, lo, hi);
  if ((rc = flush_pending(c))) return rc;  // a queued objective still needs the old unaries
  if (c->theta_scaled || c->inited || c->seen_r.size() != (size_t)c->st.nrows || c->seen_c.size() != (size_t)d.rows) {
    // first share after a solve (or ever): everything has to arrive again
    c->seen_r.assign((size_t)c->st.nrows, 0);
    c->seen_c.assign((size_t)d.rows, 0);
    c->theta_scaled = false;
    c->inited = false;
    c->have_unary = false;
  }
  const size_t k = d.k;
  if (side == 0) {
    const size_t w = (size_t)d.cols * k;
    if (nrows > 0) HIP_TRY(c, h2d_sync(c, c->th_r + (size_t)(row0 - c->st.row0) * w, data, sizeof(float) * w * nrows));
    for (int r = 0; r < nrows; r++) c->seen_r[(size_t)(row0 - c->st.row0 + r)] = 1;
    // one strip: the column chains read the same array
    if (c->one_strip()) for (int r = 0; r < nrows; r++) c->seen_c[(size_t)(row0 + r)] = 1;
  } else {
    if (!c->one_strip()) {
      const size_t w = (size_t)c->st.ncols * k;
      if (nrows > 0 && w > 0) HIP_TRY(c, h2d_sync(c, c->th_c + (size_t)row0 * w, data, sizeof(float) * w * nrows));
    }
    // (one strip: th_c is th_r and the rows call fills it; nothing to copy)
    if (!c->one_strip()) for (int r = 0; r < nrows; r++) c->seen_c[(size_t)(row0 + r)] = 1;
  }
  bool all = true;
  for (uint8_t v : c->seen_r) all = all && v;
  for (uint8_t v : c->seen_c) all = all && v;
  c->have_unary = all;
  return ADMM_OK;
}

int admm_upload_unaries_rows(admm_ctx *c, int row0, int nrows, const float *rows) { return upload_share(c, 0, row0, nrows, rows); }
int admm_upload_unaries_cols(admm_ctx *c, int row0, int nrows, const float *cols) { return upload_share(c, 1, row0, nrows, cols); }

int admm_upload_pairwise(admm_ctx *c, const float *table, int per_edge) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!table) return fail(c, ADMM_E_ARG, "null pairwise table");
  const size_t k2 = (size_t)c->d.k * c->d.k;
  if (is_chain(c)) {
    if (per_edge) return fail(c, ADMM_E_ARG, "the chain solver takes one shared k x k table (per_edge = 0)");
    HIP_TRY(c, h2d_sync(c, c->d_W, table, sizeof(float) * k2));
  } else {
    const size_t n = per_edge ? k2 * (size_t)c->E : k2;
    if (!c->d_edge_pot || c->pair_per_edge != (per_edge ? 1 : 0)) {
      if ((rc = dev_alloc(c, &c->d_edge_pot, n))) return rc;
    }
    c->pair_per_edge = per_edge ? 1 : 0;
    // a per-edge table covers the whole graph: take this rank's edges
    HIP_TRY(c, h2d_sync(c, c->d_edge_pot, table + (per_edge ? k2 * (size_t)c->e0 : 0), sizeof(float) * n));
  }
  c->have_pair = true;
  c->inited = false;
  return ADMM_OK;
}

// In-place element-wise set-up kernels of the chain solver.
namespace {
__global__ void chain_init_theta_kernel(float *__restrict__ th, long long n, float rho) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    // Grid.cpp:1478-1487 with two copies, then Grid.cpp:953-956
    const float t = th[i];
    float s = 0.0f;
    s += t;
    s += t;
    s = s / 2;
    th[i] = s / (rho * 2);
  }
}
__global__ void chain_init_w_kernel(float *Wsplit, float *Wt, float *WsplitT, float *WtT, const float *W, int k, float rho) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k * k) {
    const float ws = W[i] / rho;  // Grid.cpp:975-976
    Wsplit[i] = ws;
    Wt[i] = rho * ws;  // Tree.cpp:99
    const int t = (i % k) * k + i / k;
    WsplitT[t] = ws;
    WtT[t] = rho * ws;
  }
}
}  // namespace

int admm_init(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->have_unary || !c->have_pair) return fail(c, ADMM_E_STATE, "upload unaries (every row of this strip's shares) and the pairwise table before admm_init");
  if ((rc = flush_pending(c))) return rc;  // a queued objective still needs the old labels
  const admm_desc &d = c->d;
  hipStream_t s = c->stream;
  if (!is_chain(c)) {
    const long long E = c->E, V = c->V, k = d.k, k2 = k * k;
    const float inv_k2 = (float)(1.0 / (double)(float)(d.k * d.k));
    c->edge_obj_n = 0;  // the last solve's objective terms belong to the state this call replaces
    if (E > 0) {
      hipLaunchKernelGGL(edge_init_node_kernel, dim3(grid_for(E * 2 * k, 256)), dim3(256), 0, s, c->d_pot_node,
                         c->d_mu_node, c->d_lam_node, c->d_vert_pot, c->d_edges, c->d_inc_off, (int)E, (int)k, d.rho, c->inv_k);
      if (d.nranks > 1) {
        // Grid.cpp:1478-1487 across ranks: my copies of theta_v added one by
        // one, then every neighbour's sum (its copies added one by one) in
        // ascending rank order, divided by the global count; then InitTree
        // (Grid.cpp:952-956).  All operands are known here, so the exchange of
        // Grid.cpp:1486 is evaluated locally.  This overwrites what the
        // kernel above wrote into pot_node with single-rank counts.
        std::vector<float> avg((size_t)V * k);
        std::vector<int> own(c->V, 0);
        for (long long i = 0; i < 2 * E; i++) own[c->h_edges[i]]++;
        for (long long v = 0; v < V; v++)
          for (long long l = 0; l < k; l++) {
            const float th = c->h_unary_local[v * k + l];
            float sum = 0.0f;
            for (int q = 0; q < own[v]; q++) sum += th;
            avg[v * k + l] = sum;
          }
        for (size_t p = 0; p < c->h_send_idx.size(); p++) {  // neighbour-major = ascending rank
          const long long v = c->h_send_idx[p];
          for (long long l = 0; l < k; l++) {
            const float th = c->h_unary_local[v * k + l];
            float part = 0.0f;
            for (int q = 0; q < c->h_nb_cnt[p]; q++) part += th;
            avg[v * k + l] += part;
          }
        }
        std::vector<float> pot((size_t)E * 2 * k);
        for (long long i = 0; i < 2 * E; i++) {
          const long long v = c->h_edges[i];
          for (long long l = 0; l < k; l++) {
            float a = avg[v * k + l];
            a /= c->h_cnt[v];
            pot[i * k + l] = a / (d.rho * c->h_cnt[v]);
          }
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_pot_node, pot.data(), sizeof(float) * pot.size(), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipStreamSynchronize(s));
      }
      hipLaunchKernelGGL(edge_init_edge_kernel, dim3(grid_for(E * k2, 256)), dim3(256), 0, s, c->d_pot_edge,
                         c->d_mu_edge, c->d_edge_pot, c->pair_per_edge, E * k2, (int)k2, d.rho, inv_k2);
    }
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(V * k, 256)), dim3(256), 0, s, c->d_m_vert, V * k, c->inv_k);
    HIP_TRY(c, hipGetLastError());
  } else {
    if (c->inited) return fail(c, ADMM_E_STATE, "admm_init was already called; upload the unaries again to restart");
    const long long k = d.k;
    const long long nrow = (long long)d.batch * c->st.nrows * d.cols, ncol = (long long)d.batch * d.rows * c->st.ncols;
    // the shares of the unaries are worked out in place, once per upload: a
    // restart with a new pairwise table alone finds them in place already
    const bool scale = !c->theta_scaled;
    if (scale) hipLaunchKernelGGL(chain_init_theta_kernel, dim3(grid_for(nrow * k, 256)), dim3(256), 0, s, c->th_r, nrow * k, d.rho);
    HIP_TRY(c, hipMemsetAsync(c->lam_r, 0, sizeof(float) * nrow * k, s));
    HIP_TRY(c, hipMemsetAsync(c->lam_c, 0, sizeof(float) * ncol * k, s));
    // -1 in every label, whatever its width
    c->xrow_r = c->xrow_pub = c->xrow_buf[0];
    c->xcolT_c = c->xcolT_pub = c->xcolT_buf[0];
    c->xo_stale = false;
    c->pulled = false;
    c->red_q.clear();
    if (d.nranks == 1) {
      if (nrow > 0) HIP_TRY(c, hipMemsetAsync(c->xrow_buf[1], 0xFF, sizeof(label_t) * nrow, s));
      if (ncol > 0) HIP_TRY(c, hipMemsetAsync(c->xcolT_buf[1], 0xFF, sizeof(label_t) * ncol, s));
    }
    if (nrow > 0) {
      HIP_TRY(c, hipMemsetAsync(c->xrow_r, 0xFF, sizeof(label_t) * nrow, s));
      HIP_TRY(c, hipMemsetAsync(c->xo_r, 0xFF, sizeof(label_t) * nrow, s));
    }
    if (ncol > 0) {
      HIP_TRY(c, hipMemsetAsync(c->xcolT_c, 0xFF, sizeof(label_t) * ncol, s));
      HIP_TRY(c, hipMemsetAsync(c->xo_c, 0xFF, sizeof(label_t) * ncol, s));
    }
    if (d.nranks > 1) {
      if (scale) hipLaunchKernelGGL(chain_init_theta_kernel, dim3(grid_for(ncol * k, 256)), dim3(256), 0, s, c->th_c, ncol * k, d.rho);
      hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(nrow, 256)), dim3(256), 0, s, c->xcolT_r, nrow, -1);
      hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(ncol, 256)), dim3(256), 0, s, c->xrow_c, ncol, -1);
    }
    hipLaunchKernelGGL(chain_init_w_kernel, dim3((d.k * d.k + 255) / 256), dim3(256), 0, s, c->d_Wsplit, c->d_Wt, c->d_WsplitT, c->d_WtT, c->d_W, d.k, d.rho);
    HIP_TRY(c, hipGetLastError());
    // Potts form of the table the sweep adds (checked on the device's own values)
    std::vector<float> wt((size_t)d.k * d.k);
    HIP_TRY(c, hipMemcpyAsync(wt.data(), c->d_Wt, sizeof(float) * wt.size(), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    bool potts = d.k >= 2 && wt[1] >= 0.0f;
    for (int a = 0; a < d.k && potts; a++)
      for (int b = 0; b < d.k && potts; b++) potts = (a == b) ? wt[(size_t)a * d.k + b] == 0.0f : wt[(size_t)a * d.k + b] == wt[1];
    c->potts = potts;
    {
      // W[a][b] = g[|a - b|], g constant beyond distance 3 and nowhere above that constant
      const int k = d.k;
      bool tr = true;
      for (int a = 0; a < k && tr; a++)
        for (int b = 0; b < k && tr; b++) tr = wt[(size_t)a * k + b] == wt[(size_t)std::abs(a - b)];
      const float wfar = k > 4 ? wt[4] : INFINITY;
      for (int dd = 5; dd < k && tr; dd++) tr = wt[(size_t)dd] == wfar;
      for (int dd = 0; dd < std::min(k, 4) && tr; dd++) tr = wt[(size_t)dd] <= wfar;
      // Four chains share a wavefront in that kernel: it takes about half the instructions
      // per chain and step of the general one, which spreads a chain over a whole wavefront,
      // but on a quarter of the wavefronts (twice that many up to 1024, see twowave_chosen).
      // Small grids leave most SIMDs empty either way and the general kernel's shorter step
      // wins (256 x 256: 0.030 against 0.026 ms per sweep; 384 x 384: 0.039 against 0.042;
      // 512 x 512: 0.049 against 0.053; 1024 x 1024: 0.104 against 0.158).
      // ADMM_SWEEP_TRUNC=0/1 (tuning and test knob) overrides the choice.
      const long long tw = 2 * (((long long)d.batch * c->st.nrows + 3) / 4) + 2 * (((long long)d.batch * c->st.ncols + 3) / 4);
      const char *et = getenv("ADMM_SWEEP_TRUNC");
      c->trunc3 = tr && c->KP == 16 && (et ? atoi(et) != 0 : tw >= 384);
    }
    // int16 multiplier counts (the LAM16 kernels): the Potts kernel or the one for truncated
    // tables (the general kernel reads binary32 only), penalty / 2 a power of two well inside
    // the binary32 range.
    // ADMM_LAM16=0 keeps binary32; ADMM_LAM16_LIMIT=n switches to it after n sweeps (test knobs).
    {
      const float hpen = d.penalty * 0.5f;
      int ex = 0;
      const bool pow2 = hpen > 0.0f && std::frexp(hpen, &ex) == 0.5f && ex > -100 && ex < 100;
      const char *e16 = getenv("ADMM_LAM16"), *elim = getenv("ADMM_LAM16_LIMIT");
      // (the general kernel: with one lane per label and 16 or 32 label lanes, where the bytes count -
      // 2048 x 2048 with a dense table, k = 16: 0.339 -> 0.297 ms per iteration, k = 32: 0.841 -> 0.830;
      // 64 labels are bound by the vector ALU and lose 1 % to the conversions; ADMM_LAM16_GENERAL=0
      // keeps binary32)
      bool general_counts = false;
      {
        const SweepShape shg = pick_shape(c->KP, (long long)d.batch * c->st.nrows + (long long)d.batch * c->st.ncols);
        const char *eg = getenv("ADMM_LAM16_GENERAL");
        general_counts = shg.NS == 1 && (c->KP == 16 || c->KP == 32) && !(eg && atoi(eg) == 0);
      }
      c->lam16 = ((potts && c->KP >= 8 && potts_enabled()) || c->trunc3 || general_counts) && pow2 && !(e16 && atoi(e16) == 0);
      c->lam16_sweeps = 0;
      c->lam16_limit = elim ? std::max(0, std::min(32000, atoi(elim))) : 32000;
      // int8 counts: ADMM_LAM8=0 never; ADMM_LAM8_THR / ADMM_LAM8_EVERY move the scan's threshold and
      // spacing (test knobs: a threshold of 1 or 2 makes a short run go through the change to int16)
      const char *e8 = getenv("ADMM_LAM8"), *ethr = getenv("ADMM_LAM8_THR"), *eev = getenv("ADMM_LAM8_EVERY");
      c->lam8 = c->lam16 && !(e8 && atoi(e8) == 0);
      c->lam8_thr = ethr ? std::max(1, std::min(100, atoi(ethr))) : 64;
      c->lam8_every = eev ? std::max(1, std::min(126 - c->lam8_thr, atoi(eev))) : 16;
      c->lam8_clean = 0;
      for (auto &sc : c->scans) sc.pending = false;
      if (c->lam8 && !c->d_scan_flag) {
        int rc8;
        if ((rc8 = dev_alloc(c, &c->d_scan_flag, (size_t)admm_ctx::kScans))) return rc8;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_scan_flag, sizeof(int) * admm_ctx::kScans, hipHostMallocDefault));
      }
    }
  }
  if (is_chain(c)) c->direct_labels = direct_labels_chosen(c);
  HIP_TRY(c, hipStreamSynchronize(s));
  c->inited = true;
  if (is_chain(c)) c->theta_scaled = true;
  c->lin_valid = false;
  return ADMM_OK;
}

int admm_compute(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "admm_compute before admm_init");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = c->prof && (c->prof_seen++ % c->prof_every) == 0;
  if (timed) {
    if (c->prof_used == c->prof_events.size()) {
      hipEvent_t a, b;
      HIP_TRY(c, hipEventCreate(&a));
      HIP_TRY(c, hipEventCreate(&b));
      c->prof_events.push_back({a, b});
    }
    e0 = c->prof_events[c->prof_used].first;
    e1 = c->prof_events[c->prof_used].second;
    c->prof_used++;
    HIP_TRY(c, hipEventRecord(e0, c->stream));
  }
  // (per-chain unary terms of an earlier sweep that nobody has summed yet are summed by a
  // spare workgroup of this sweep's launch, or by a launch of their own just before it)
  rc = is_chain(c) ? chain_sweep(c, false) : edge_compute(c);
  if (rc) return rc;
  if (is_chain(c) && c->pending_slot) {
    // this sweep summed, per chain, the unaries of the labels it started from:
    // the missing term of the objective queued for the previous solve.  The
    // sum over chains rides along with the next queued objective's reduction.
    c->swept_slot = c->pending_slot;
    c->pending_slot = 0;
  }
  c->lin_valid = true;
  if (timed) HIP_TRY(c, hipEventRecord(e1, c->stream));
  return ADMM_OK;
}

namespace {

extern "C++" const ExchangePlan &plan_of(admm_ctx *c) {  // (a reference to a C++ type: not for the C linkage of the block around it)
  if (c->plan_ready) return c->plan;
  const admm_desc &d = c->d;
  const int G = d.nranks;
  ExchangePlan &p = c->plan;
  p.off_a.assign(G + 1, 0); p.off_b.assign(G + 1, 0);
  p.r0.resize(G); p.nr.resize(G); p.c0.resize(G); p.nc.resize(G);
  for (int h = 0; h < G; h++) {
    admm_partition(d.rows, G, h, &p.r0[h], &p.nr[h]);
    admm_partition(d.cols, G, h, &p.c0[h], &p.nc[h]);
  }
  for (int h = 0; h < G; h++) {
    p.off_a[h + 1] = p.off_a[h] + (long long)c->st.nrows * p.nc[h];
    p.off_b[h + 1] = p.off_b[h] + (long long)c->st.ncols * p.nr[h];
  }
  c->plan_ready = true;
  return p;
}

// Gathers the strided label blocks into sendbuf (queued on the context's stream).
// own_direct: this rank's own blocks go straight into its receive arrays (label
// mode only; the caller then leaves itself out of the exchange).
int pack_labels(admm_ctx *c, const ExchangePlan &p, bool own_direct = false) {
  const admm_desc &d = c->d;
  const int nr = c->st.nrows, nc = c->st.ncols;
  int *send_a = c->sendbuf, *send_b = c->sendbuf + p.off_a[d.nranks];
  const long long total = (long long)nr * d.cols + (long long)nc * d.rows;
  if (total > 0)
    hipLaunchKernelGGL(pack_strip_labels_kernel, dim3(grid_for(total, 256)), dim3(256), 0, c->stream, send_a, send_b,
                       c->xrow_r, c->xcolT_c, nr, nc, d.rows, d.cols, d.nranks, d.rank,
                       own_direct ? c->xrow_c + (size_t)p.r0[d.rank] * nc : nullptr,
                       own_direct ? c->xcolT_r + (size_t)p.c0[d.rank] * nr : nullptr);
  if (d.exchange == ADMM_EXCHANGE_DENSE) {
    const long long n = p.off_a[d.nranks] + p.off_b[d.nranks];
    if (n > 0)
      hipLaunchKernelGGL(labels_to_marginals_kernel, dim3(grid_for(n * d.k, 256)), dim3(256), 0, c->stream,
                         c->sendbuf_mu, c->sendbuf, n, d.k, c->inv_k);
  }
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

// What goes over the wire, in 4-byte elements: labels, or their marginals
// (k floats per label) in the dense mode.  Receive side: region a = the peers'
// row-chain results on my columns ([rows][nc]), region b = their column-chain
// results on my rows ([cols][nr]); in the dense mode both are staged in
// recvbuf_mu and turned back into labels by unpack_received().
struct Wire {
  int m;               // elements per label
  const int32_t *send; // packed blocks
  int32_t *recv_a, *recv_b;
};

Wire wire_of(admm_ctx *c) {
  const admm_desc &d = c->d;
  Wire w;
  if (d.exchange == ADMM_EXCHANGE_DENSE) {
    w.m = d.k;
    w.send = reinterpret_cast<const int32_t *>(c->sendbuf_mu);
    w.recv_a = reinterpret_cast<int32_t *>(c->recvbuf_mu);
    w.recv_b = reinterpret_cast<int32_t *>(c->recvbuf_mu + (size_t)d.rows * c->st.ncols * d.k);
  } else {
    w.m = 1;
    w.send = c->sendbuf;
    w.recv_a = c->xrow_c;
    w.recv_b = c->xcolT_r;
  }
  return w;
}

int unpack_received(admm_ctx *c) {
  const admm_desc &d = c->d;
  if (d.exchange != ADMM_EXCHANGE_DENSE) return ADMM_OK;
  // Before the first solve every copy is uniform and every label is -1 on
  // every strip already.
  if (!c->lin_valid) return ADMM_OK;
  const long long na = (long long)d.rows * c->st.ncols, nb = (long long)d.cols * c->st.nrows;
  if (na > 0)
    hipLaunchKernelGGL(marginals_to_labels_kernel, dim3(grid_for(na, 256)), dim3(256), 0, c->stream, c->xrow_c,
                       c->recvbuf_mu, na, d.k);
  if (nb > 0)
    hipLaunchKernelGGL(marginals_to_labels_kernel, dim3(grid_for(nb, 256)), dim3(256), 0, c->stream, c->xcolT_r,
                       c->recvbuf_mu + (size_t)na * d.k, nb, d.k);
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

}  // namespace

// Edge solver on several ranks: this rank's partial consensus sums, and the
// ones it shares packed neighbour by neighbour (Grid.cpp:1645-1651).
static int edge_pack_partials(admm_ctx *c) {
  const long long n = (long long)c->V * c->d.k;
  hipLaunchKernelGGL(edge_partial_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d_m_part, c->d_mu_node,
                     c->d_inc_off, c->d_inc, c->V, c->d.k);
  const int S = (int)c->h_send_idx.size();
  if (S > 0)
    hipLaunchKernelGGL(edge_pack_kernel, dim3(grid_for((long long)S * c->d.k, 256)), dim3(256), 0, c->stream, c->d_sendv,
                       c->d_m_part, c->d_send_idx, S, c->d.k);
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

int admm_communicate(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "admm_communicate before admm_init");
  const admm_desc &d = c->d;
  if (d.nranks == 1) return ADMM_OK;  // no neighbours: Grid.cpp:1643 loops zero times
  if (!c->comm) return fail(c, ADMM_E_COMM, "admm_comm_init must be called before admm_communicate when nranks > 1");
  if (!is_chain(c)) {
    // CommunicateByEdge (Grid.cpp:1639-1663): one message to and one from every neighbour
    if ((rc = edge_pack_partials(c))) return rc;
    NcclGroup grp(c);
    for (size_t q = 0; q < c->nb_rank.size(); q++) {
      const size_t off = (size_t)c->nb_off[q] * d.k, n = (size_t)(c->nb_off[q + 1] - c->nb_off[q]) * d.k;
      if (n == 0) continue;
      grp.add(ncclSend(c->d_sendv + off, n, ncclFloat, c->nb_rank[q], c->comm, c->stream), "ncclSend");
      grp.add(ncclRecv(c->d_recvv + off, n, ncclFloat, c->nb_rank[q], c->comm, c->stream), "ncclRecv");
    }
    return grp.end();
  }
  hipStream_t s = c->stream;
  const int G = d.nranks;
  const int nr = c->st.nrows, nc = c->st.ncols;
  const ExchangePlan &p = plan_of(c);
  const bool own_direct = d.exchange == ADMM_EXCHANGE_LABELS;
  if ((rc = pack_labels(c, p, own_direct))) return rc;
  const Wire w = wire_of(c);
  const size_t m = (size_t)w.m;
  const int32_t *send_a = w.send, *send_b = w.send + p.off_a[G] * m;
  // Both kinds of block arrive contiguous on the other side: region a is
  // [rows][nc] (a block = whole rows of it) and region b is [cols][nr].
  // Elements travel as 4-byte words whatever they hold.
  // my own blocks stay on the device, no send to self: the pack kernel has
  // already put them in place, or (dense mode) a plain copy does
  if (!own_direct) {
    const int h = d.rank;
    if ((long long)nr * p.nc[h] > 0)
      HIP_TRY(c, hipMemcpyAsync(w.recv_a + (size_t)p.r0[h] * nc * m, send_a + p.off_a[h] * m, (size_t)nr * p.nc[h] * m * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if ((long long)nc * p.nr[h] > 0)
      HIP_TRY(c, hipMemcpyAsync(w.recv_b + (size_t)p.c0[h] * nr * m, send_b + p.off_b[h] * m, (size_t)nc * p.nr[h] * m * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  }
  {
    NcclGroup grp(c);
    for (int h = 0; h < G; h++) {
      if (h == d.rank) continue;
      if ((long long)nr * p.nc[h] > 0) grp.add(ncclSend(send_a + p.off_a[h] * m, (size_t)nr * p.nc[h] * m, ncclInt32, h, c->comm, s), "ncclSend");
      if ((long long)p.nr[h] * nc > 0) grp.add(ncclRecv(w.recv_a + (size_t)p.r0[h] * nc * m, (size_t)p.nr[h] * nc * m, ncclInt32, h, c->comm, s), "ncclRecv");
      if ((long long)nc * p.nr[h] > 0) grp.add(ncclSend(send_b + p.off_b[h] * m, (size_t)nc * p.nr[h] * m, ncclInt32, h, c->comm, s), "ncclSend");
      if ((long long)p.nc[h] * nr > 0) grp.add(ncclRecv(w.recv_b + (size_t)p.c0[h] * nr * m, (size_t)p.nc[h] * nr * m, ncclInt32, h, c->comm, s), "ncclRecv");
    }
    if ((rc = grp.end())) return rc;
  }
  return unpack_received(c);
}

int admm_exchange_counts(admm_ctx *c, int64_t *send_counts, int64_t *recv_counts) {
  if (!c) return ADMM_E_ARG;
  if (!is_chain(c)) {
    // edge solver: k floats per shared vertex to and from every neighbouring rank
    for (int h = 0; h < c->d.nranks; h++) {
      if (send_counts) send_counts[h] = 0;
      if (recv_counts) recv_counts[h] = 0;
    }
    for (size_t q = 0; q < c->nb_rank.size(); q++) {
      const int64_t n = (int64_t)(c->nb_off[q + 1] - c->nb_off[q]) * c->d.k;
      if (send_counts) send_counts[c->nb_rank[q]] = n;
      if (recv_counts) recv_counts[c->nb_rank[q]] = n;
    }
    return ADMM_OK;
  }
  const ExchangePlan &p = plan_of(c);
  for (int h = 0; h < c->d.nranks; h++) {
    const int64_t m = c->d.exchange == ADMM_EXCHANGE_DENSE ? c->d.k : 1;
    if (send_counts) send_counts[h] = m * ((int64_t)c->st.nrows * p.nc[h] + (int64_t)c->st.ncols * p.nr[h]);
    if (recv_counts) recv_counts[h] = m * ((int64_t)p.nr[h] * c->st.ncols + (int64_t)p.nc[h] * c->st.nrows);
  }
  return ADMM_OK;
}

int admm_exchange_pack_host(admm_ctx *c, int32_t *host_send) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "exchange before admm_init");
  if (!host_send) return fail(c, ADMM_E_ARG, "null buffer");
  if (c->d.nranks == 1) return ADMM_OK;
  if (!is_chain(c)) {
    // the neighbours are in ascending rank order, so the packed buffer is peer-major as it stands
    if ((rc = edge_pack_partials(c))) return rc;
    const size_t n = c->h_send_idx.size() * (size_t)c->d.k;
    if (n) HIP_TRY(c, hipMemcpyAsync(host_send, c->d_sendv, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ADMM_OK;
  }
  const ExchangePlan &p = plan_of(c);
  if ((rc = pack_labels(c, p))) return rc;
  const int G = c->d.nranks;
  const Wire w = wire_of(c);
  const size_t m = (size_t)w.m;
  const int32_t *send_a = w.send, *send_b = w.send + p.off_a[G] * m;
  int32_t *dst = host_send;
  for (int h = 0; h < G; h++) {  // peer-major: [a_h | b_h]
    const size_t na = (size_t)c->st.nrows * p.nc[h] * m, nb = (size_t)c->st.ncols * p.nr[h] * m;
    if (na) HIP_TRY(c, hipMemcpyAsync(dst, send_a + p.off_a[h] * m, na * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    dst += na;
    if (nb) HIP_TRY(c, hipMemcpyAsync(dst, send_b + p.off_b[h] * m, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    dst += nb;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADMM_OK;
}

int admm_exchange_unpack_host(admm_ctx *c, const int32_t *host_recv) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "exchange before admm_init");
  if (!host_recv) return fail(c, ADMM_E_ARG, "null buffer");
  if (c->d.nranks == 1) return ADMM_OK;
  if (!is_chain(c)) {
    const size_t n = c->h_send_idx.size() * (size_t)c->d.k;
    if (n) HIP_TRY(c, hipMemcpyAsync(c->d_recvv, host_recv, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ADMM_OK;
  }
  const ExchangePlan &p = plan_of(c);
  const int nr = c->st.nrows, nc = c->st.ncols;
  const int32_t *src = host_recv;
  const Wire w = wire_of(c);
  const size_t m = (size_t)w.m;
  for (int h = 0; h < c->d.nranks; h++) {  // from peer h: its rows on my columns, then its columns on my rows
    const size_t na = (size_t)p.nr[h] * nc * m, nb = (size_t)p.nc[h] * nr * m;
    if (na) HIP_TRY(c, hipMemcpyAsync(w.recv_a + (size_t)p.r0[h] * nc * m, src, na * sizeof(int), hipMemcpyHostToDevice, c->stream));
    src += na;
    if (nb) HIP_TRY(c, hipMemcpyAsync(w.recv_b + (size_t)p.c0[h] * nr * m, src, nb * sizeof(int), hipMemcpyHostToDevice, c->stream));
    src += nb;
  }
  if ((rc = unpack_received(c))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's buffer may be reused at once
  return ADMM_OK;
}

static bool group_pull_probe(admm_ctx **ctxs, int n);
// Can every strip of the group read the others' label arrays from its own kernels?  Label
// exchange only; all strips on one device, or on devices that grant each other peer access
// (asked for here, once per pair).  ADMM_GROUP_PULL=0 (tuning and test knob) keeps the
// pack / copy / transpose form.
static bool group_pull_ok(admm_ctx **ctxs, int n) {
  if (n > kPullMaxStrips || !is_chain(ctxs[0]) || ctxs[0]->d.exchange != ADMM_EXCHANGE_LABELS) return false;
  if (const char *e = getenv("ADMM_GROUP_PULL")) if (atoi(e) == 0) return false;
  // (the answer for this group is worked out once: it is a property of the strips' devices)
  admm_ctx *first = ctxs[0];
  if (first->pull_group_n == n && first->pull_group_last == ctxs[n - 1]) return first->pull_group_ok;
  const bool ok = group_pull_probe(ctxs, n);
  first->pull_group_n = n;
  first->pull_group_last = ctxs[n - 1];
  first->pull_group_ok = ok;
  return ok;
}
static bool group_pull_probe(admm_ctx **ctxs, int n) {
  for (int g = 0; g < n; g++)
    for (int h = 0; h < n; h++) {
      const int dg = ctxs[g]->d.device, dh = ctxs[h]->d.device;
      if (dg == dh) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, dg, dh) != hipSuccess || !can) return false;
      if (hipSetDevice(dg) != hipSuccess) return false;
      const hipError_t e = hipDeviceEnablePeerAccess(dh, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
      if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
    }
  return true;
}

int admm_communicate_group(admm_ctx **ctxs, int n) {
  if (!ctxs || n < 1) return ADMM_E_ARG;
  for (int g = 0; g < n; g++) {
    admm_ctx *c = ctxs[g];
    if (!c) return ADMM_E_ARG;
    if (c->d.solver != ctxs[0]->d.solver || c->d.nranks != n || c->d.rank != g || c->d.rows != ctxs[0]->d.rows ||
        c->d.cols != ctxs[0]->d.cols || c->d.k != ctxs[0]->d.k || c->d.exchange != ctxs[0]->d.exchange ||
        c->Eglobal != ctxs[0]->Eglobal || c->Vglobal != ctxs[0]->Vglobal)
      return fail(c, ADMM_E_ARG, "admm_communicate_group needs the n strips of one problem, in rank order");
    if (!c->inited) return fail(c, ADMM_E_STATE, "admm_communicate_group before admm_init");
  }
  if (n == 1) return ADMM_OK;
  if (!is_chain(ctxs[0])) {
    // edge solver: every rank packs, then pulls each neighbour's block for it
    int rc;
    for (int g = 0; g < n; g++) {
      admm_ctx *c = ctxs[g];
      if ((rc = check_ctx(c)) || (rc = edge_pack_partials(c))) return rc;
      if (!c->ev_packed) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
      if (!c->ev_pulled) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pulled, hipEventDisableTiming));
      HIP_TRY(c, hipEventRecord(c->ev_packed, c->stream));
    }
    for (int g = 0; g < n; g++) {
      admm_ctx *r = ctxs[g];
      if ((rc = check_ctx(r))) return rc;
      for (size_t q = 0; q < r->nb_rank.size(); q++) {
        admm_ctx *sd = ctxs[r->nb_rank[q]];
        // my block in the neighbour's send buffer: where it lists me
        size_t qq = 0;
        while (qq < sd->nb_rank.size() && sd->nb_rank[qq] != g) qq++;
        const size_t cnt = (size_t)(r->nb_off[q + 1] - r->nb_off[q]);
        if (qq == sd->nb_rank.size() || (size_t)(sd->nb_off[qq + 1] - sd->nb_off[qq]) != cnt)
          return fail(r, ADMM_E_STATE, "the exchange plans of ranks %d and %d disagree", g, r->nb_rank[q]);
        HIP_TRY(r, hipStreamWaitEvent(r->stream, sd->ev_packed, 0));
        if (cnt)
          HIP_TRY(r, hipMemcpyPeerAsync(r->d_recvv + (size_t)r->nb_off[q] * r->d.k, r->d.device,
                                        sd->d_sendv + (size_t)sd->nb_off[qq] * r->d.k, sd->d.device,
                                        cnt * r->d.k * sizeof(float), r->stream));
      }
      HIP_TRY(r, hipEventRecord(r->ev_pulled, r->stream));
    }
    for (int g = 0; g < n; g++) {
      admm_ctx *c = ctxs[g];
      if ((rc = check_ctx(c))) return rc;
      for (int h : c->nb_rank) HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[h]->ev_pulled, 0));
    }
    return ADMM_OK;
  }
  int rc;
  if (group_pull_ok(ctxs, n)) {
    // Every strip pulls its labels from the other strips' arrays and lays them out along its
    // chains in the same launch: no pack, no copies, and nothing left for admm_update.
    for (int g = 0; g < n; g++) {
      admm_ctx *c = ctxs[g];
      if ((rc = check_ctx(c))) return rc;
      if (!c->ev_packed) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
      if (!c->ev_pulled) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pulled, hipEventDisableTiming));
      HIP_TRY(c, hipEventRecord(c->ev_packed, c->stream));  // my chains' labels are final
    }
    for (int h = 0; h < n; h++) {
      admm_ctx *r = ctxs[h];
      if ((rc = check_ctx(r))) return rc;
      for (int g = 0; g < n; g++)
        if (g != h) HIP_TRY(r, hipStreamWaitEvent(r->stream, ctxs[g]->ev_packed, 0));
      PullArgs a{};
      for (int g = 0; g < n; g++) { a.src_row[g] = ctxs[g]->xrow_r; a.src_colT[g] = ctxs[g]->xcolT_c; }
      a.G = n; a.rows = r->d.rows; a.cols = r->d.cols;
      a.r0 = r->st.row0; a.nr = r->st.nrows; a.c0 = r->st.col0; a.nc = r->st.ncols;
      a.recv_a = r->xrow_c; a.recv_b = r->xcolT_r; a.xo_c = r->xo_c_alt; a.xo_r = r->xo_r_alt;
      a.tiles_a_minor = (a.nc + 63) / 64; a.tiles_b_minor = (a.nr + 63) / 64;
      a.tiles_a = a.nc > 0 ? ((a.rows + 63) / 64) * a.tiles_a_minor : 0;
      const int tiles_b = a.nr > 0 ? ((a.cols + 63) / 64) * a.tiles_b_minor : 0;
      if (a.tiles_a + tiles_b > 0) {
        hipLaunchKernelGGL(pull_labels_kernel, dim3((unsigned)(a.tiles_a + tiles_b)), dim3(256), 0, r->stream, a);
        HIP_TRY(r, hipGetLastError());
      }
      r->pulled = true;
      HIP_TRY(r, hipEventRecord(r->ev_pulled, r->stream));
    }
    // a strip's next sweep overwrites its labels: not before everybody has pulled them
    for (int g = 0; g < n; g++) {
      admm_ctx *c = ctxs[g];
      if ((rc = check_ctx(c))) return rc;
      for (int h = 0; h < n; h++)
        if (h != g) HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[h]->ev_pulled, 0));
    }
    return ADMM_OK;
  }
  std::vector<const ExchangePlan *> plans;
  // 1. every strip packs on its own stream and marks the point
  for (int g = 0; g < n; g++) {
    admm_ctx *c = ctxs[g];
    if ((rc = check_ctx(c))) return rc;
    plans.push_back(&plan_of(c));
    // label mode: a strip's own blocks go straight into its receive arrays (as in admm_communicate)
    if ((rc = pack_labels(c, *plans[g], c->d.exchange == ADMM_EXCHANGE_LABELS))) return rc;
    if (!c->ev_packed) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    if (!c->ev_pulled) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pulled, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_packed, c->stream));
  }
  // 2. every strip pulls its blocks on its own stream
  for (int h = 0; h < n; h++) {
    admm_ctx *r = ctxs[h];
    if ((rc = check_ctx(r))) return rc;
    const int nr = r->st.nrows, nc = r->st.ncols;
    for (int g = 0; g < n; g++) {
      admm_ctx *sd = ctxs[g];
      const ExchangePlan &p = *plans[g];
      if (g == h && r->d.exchange == ADMM_EXCHANGE_LABELS) continue;  // already in place
      HIP_TRY(r, hipStreamWaitEvent(r->stream, sd->ev_packed, 0));
      const Wire ws = wire_of(sd), wr = wire_of(r);
      const size_t m = (size_t)ws.m;
      const int32_t *send_a = ws.send, *send_b = ws.send + p.off_a[n] * m;
      const size_t na = (size_t)sd->st.nrows * nc * m, nb = (size_t)sd->st.ncols * nr * m;
      if (na) HIP_TRY(r, hipMemcpyPeerAsync(wr.recv_a + (size_t)p.r0[g] * nc * m, r->d.device, send_a + p.off_a[h] * m, sd->d.device, na * sizeof(int), r->stream));
      if (nb) HIP_TRY(r, hipMemcpyPeerAsync(wr.recv_b + (size_t)p.c0[g] * nr * m, r->d.device, send_b + p.off_b[h] * m, sd->d.device, nb * sizeof(int), r->stream));
    }
    HIP_TRY(r, hipEventRecord(r->ev_pulled, r->stream));
    if ((rc = unpack_received(r))) return rc;
  }
  // 3. a strip's send buffer is free again once everybody has pulled from it
  for (int g = 0; g < n; g++) {
    admm_ctx *c = ctxs[g];
    if ((rc = check_ctx(c))) return rc;
    for (int h = 0; h < n; h++)
      if (h != g) HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[h]->ev_pulled, 0));
  }
  return ADMM_OK;
}

int admm_update(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "admm_update before admm_init");
  const admm_desc &d = c->d;
  if (!is_chain(c)) {
    const long long n = (long long)c->V * d.k;
    if (d.nranks > 1)  // UpdateMuArrAfterComm on the sums Communicate left (Grid.cpp:163-181)
      hipLaunchKernelGGL(edge_finalize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d_m_vert,
                         c->d_m_part, c->d_recvv, c->d_rcv_off, c->d_rcv_pos, c->d_cnt, c->V, d.k);
    else
      hipLaunchKernelGGL(edge_consensus_kernel, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d_m_vert,
                         c->d_mu_node, c->d_inc_off, c->d_inc, c->V, d.k);
    HIP_TRY(c, hipGetLastError());
    return ADMM_OK;
  }
  // The consensus of a node is a function of the two labels it received
  // (Grid.cpp:163-181 with one-hot marginals); the sweep evaluates it from
  // them, so Update only lays each set of labels out along the crossing chains.
  // [cols][nrows] -> [nrows][cols] and [rows][ncols] -> [ncols][rows]
  if (c->direct_labels) {
    // the newest labels become the ones the crossing chains read; nothing to launch
    c->xrow_pub = c->xrow_r;
    c->xcolT_pub = c->xcolT_c;
    c->xo_stale = true;
    return ADMM_OK;
  }
  if (c->one_strip())
    return transpose_labels2<label_t>(c, c->xo_r, c->xcolT_c, d.cols, c->st.nrows, c->xo_c, c->xrow_r, d.rows, c->st.ncols);
  if (c->pulled) {  // admm_communicate_group laid them out already, in the spare arrays
    c->pulled = false;
    std::swap(c->xo_r, c->xo_r_alt);
    std::swap(c->xo_c, c->xo_c_alt);
    return ADMM_OK;
  }
  return transpose_labels2<int>(c, c->xo_r, c->xcolT_r, d.cols, c->st.nrows, c->xo_c, c->xrow_c, d.rows, c->st.ncols);
}

// Direct labels: the transposed copies (the published labels laid out along the crossing
// chains), for the kernels that read them; made when first asked for after an admm_update.
static int ensure_xo(admm_ctx *c) {
  if (!c->direct_labels || !c->xo_stale) return ADMM_OK;
  c->xo_stale = false;
  const admm_desc &d = c->d;
  return transpose_labels2<label_t>(c, c->xo_r, c->xcolT_pub, d.cols, c->st.nrows, c->xo_c, c->xrow_pub, d.rows, c->st.ncols);
}

namespace {
// Sums per-row partial pairs into dst[0] (+=primal) and dst[1] (+=linear).
__global__ __launch_bounds__(256) void objective_reduce_kernel(double *__restrict__ dst,
                                                               const double *__restrict__ part,
                                                               int n, int accumulate) {
  __shared__ double s_red[2][4];
  double p = 0.0, l = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { p += part[2 * i]; l += part[2 * i + 1]; }
  p = wave_sum(p);
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = p; s_red[1][threadIdx.x >> 6] = l; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ps = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const double ls = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    dst[0] = (accumulate ? dst[0] : 0.0) + ps;
    dst[1] = (accumulate ? dst[1] : 0.0) + ls;
  }
}
}  // namespace

namespace {
// dst[0] = sum of part[i * stride], i < n (+ the old dst[0] when accumulate).
__global__ __launch_bounds__(256) void linear_reduce_kernel(double *__restrict__ dst,
                                                            const double *__restrict__ part, int n,
                                                            int stride, int accumulate) {
  __shared__ double s_red[4];
  double l = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) l += part[(size_t)i * stride];
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] = (accumulate ? dst[0] : 0.0) + s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
}  // namespace

namespace {
// Up to two sums in one launch, one workgroup each: dst[j][0] = sum of part[j][0..n[j]).
struct Reduce2Args {
  double *dst[2];
  const double *part[2];
  int n[2];
};
__global__ __launch_bounds__(1024) void linear_reduce2_kernel(const Reduce2Args a) {
  __shared__ double s_red[16];
  const int j = blockIdx.x;
  const double *__restrict__ part = a.part[j];
  const int n = a.n[j];
  double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n; i += 4 * 1024) {
    l0 += part[i]; l1 += part[i + 1024]; l2 += part[i + 2048]; l3 += part[i + 3072];
  }
  for (; i < n; i += 1024) l0 += part[i];
  double l = wave_sum((l0 + l1) + (l2 + l3));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; w++) t += s_red[w];
    a.dst[j][0] = t;
  }
}
// Many terms (a batch of grids has one per half chain of every grid): kStage1
// workgroups per sum leave partial sums of contiguous slices in `mid`, and
// linear_reduce2_kernel then sums those.  One workgroup alone would walk
// megabytes at the latency of a single wavefront's loads.
constexpr int kStage1 = 256;
__global__ __launch_bounds__(256) void linear_stage1_kernel(const Reduce2Args a, double *__restrict__ mid) {
  __shared__ double s_red[4];
  const int j = blockIdx.y, b = blockIdx.x;
  const double *__restrict__ part = a.part[j];
  const int n = a.n[j];
  const int per = (n + kStage1 - 1) / kStage1;
  const int lo = b * per, hi = min(n, lo + per);
  double l = 0.0;
  for (int i = lo + (int)threadIdx.x; i < hi; i += 256) l += part[i];
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) mid[j * kStage1 + b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Up to two sums, queued on the context's stream.
int launch_reduce2(admm_ctx *c, Reduce2Args a, int jobs) {
  if (a.n[0] > 64 * 1024) {
    static_assert(2 * kStage1 <= 2048 * 2, "the partial-sum buffer holds both jobs");
    hipLaunchKernelGGL(linear_stage1_kernel, dim3(kStage1, jobs), dim3(256), 0, c->stream, a, c->d_partial);
    for (int j = 0; j < jobs; j++) { a.part[j] = c->d_partial + j * kStage1; a.n[j] = kStage1; }
  }
  hipLaunchKernelGGL(linear_reduce2_kernel, dim3(jobs), dim3(1024), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return ADMM_OK;
}

// Sums of objective terms may wait for the next sweep's launch when one workgroup can do them
// in the time a sweep takes anyway (a batch of many grids has megabytes of terms: two stages of
// their own).  ADMM_DEFER_SUMS=0 (test knob): every sum is launched where it is queued.
bool red_deferred(const admm_ctx *c) {
  const char *e = getenv("ADMM_DEFER_SUMS");
  return !(e && atoi(e) == 0) && chain_terms(c) <= 64 * 1024;
}

// Launches the queued sums now.
int flush_red(admm_ctx *c) {
  if (c->red_q.empty()) return ADMM_OK;
  Reduce2Args a{};
  const int jobs = (int)c->red_q.size();
  for (int j = 0; j < jobs; j++) { a.dst[j] = c->red_q[j].dst; a.part[j] = c->red_q[j].part; a.n[j] = c->red_q[j].n; }
  c->red_q.clear();
  return launch_reduce2(c, a, jobs);
}
}  // namespace

// Queues the chain objective kernels; dst (device, 2 doubles) receives the
// primal and the linear sums of this strip, not yet scaled by rho.
static int chain_obj_enqueue(admm_ctx *c, double *dst) {
  hipStream_t s = c->stream;
  int pass = ensure_xo(c);
  if (pass) return pass;
  for (int seg = 0; seg < 2; seg++) {
    ChainSeg sg{};
    if (seg == 0) fill_seg_rows(c, sg, false); else fill_seg_cols(c, sg, false);
    const long long total = (long long)sg.nchains * sg.T;
    if (total <= 0) continue;
    const int blocks = (int)std::min<long long>(c->n_partial, (total + 255) / 256);
    hipLaunchKernelGGL(chain_objective_kernel, dim3(blocks), dim3(256), 0, s, c->d_partial, sg.theta, c->d_Wsplit,
                       sg.x, sg.xo, sg.nchains, sg.T, sg.cpg, sg.gstride, sg.cstride, sg.sstride, c->d.k,
                       seg == 0 ? 1 : 0);
    hipLaunchKernelGGL(objective_reduce_kernel, dim3(1), dim3(256), 0, s, dst, c->d_partial, blocks, pass);
    HIP_TRY(c, hipGetLastError());
    pass++;
  }
  if (pass == 0) HIP_TRY(c, hipMemsetAsync(dst, 0, 2 * sizeof(double), s));
  return ADMM_OK;
}

static int chain_obj_partials(admm_ctx *c, double *primal, double *linear) {
  int rc = chain_obj_enqueue(c, c->d_obj_slots);
  if (rc) return rc;
  double h[2];
  HIP_TRY(c, hipMemcpyAsync(h, c->d_obj_slots, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (primal) *primal = h[0] * c->d.rho;
  if (linear) *linear = h[1] * c->d.rho;
  return ADMM_OK;
}

// A queued or returned linear objective is slot[0] + slot[1]: the unary and
// the pairwise energy of every chain's own labels (Grid.cpp:1665-1680).

// slot[1] = pairwise term; the last sweep left it per half chain in
// d_chain_lin.  The same launch sums the unary terms that sweep left in
// d_chain_unary into the slot they belong to (swept_slot), if any.
// defer: the sums may wait for the next sweep's launch (a read of the slot launches them).
static int chain_pairwise_enqueue(admm_ctx *c, double *slot, bool defer = false) {
  int rc = flush_red(c);  // sums queued for an earlier solve whose sweep never came
  if (rc) return rc;
  Reduce2Args a{};
  const int n = chain_terms(c);
  a.dst[0] = slot + 1; a.part[0] = c->d_chain_lin; a.n[0] = n;
  int jobs = 1;
  if (c->swept_slot) {
    a.dst[1] = c->d_obj_slots + 2 * (size_t)c->swept_slot; a.part[1] = c->d_chain_unary; a.n[1] = n;
    c->swept_slot = 0;
    jobs = 2;
  }
  if (defer && red_deferred(c)) {
    for (int j = 0; j < jobs; j++) c->red_q.push_back({a.dst[j], a.part[j], a.n[j]});
    return ADMM_OK;
  }
  return launch_reduce2(c, a, jobs);
}

// Sums the unary terms the last sweep left in d_chain_unary into their slot now.
static int settle_swept(admm_ctx *c) {
  if (!c->swept_slot) return ADMM_OK;
  Reduce2Args a{};
  a.dst[0] = c->d_obj_slots + 2 * (size_t)c->swept_slot; a.part[0] = c->d_chain_unary;
  a.n[0] = chain_terms(c);
  c->swept_slot = 0;
  return launch_reduce2(c, a, 1);
}
namespace {
int settle_swept_now(admm_ctx *c) { return settle_swept(c); }
}  // namespace

// slot[0] = unary term of the current labels, gathered now.
static int chain_unary_gather(admm_ctx *c, double *slot) {
  hipStream_t s = c->stream;
  int pass = 0;
  for (int seg = 0; seg < 2; seg++) {
    ChainSeg sg{};
    if (seg == 0) fill_seg_rows(c, sg, false); else fill_seg_cols(c, sg, false);
    const long long total = (long long)sg.nchains * sg.T;
    if (total <= 0) continue;
    const int blocks = (int)std::min<long long>(c->n_partial, (total + 255) / 256);
    hipLaunchKernelGGL(chain_objective_kernel, dim3(blocks), dim3(256), 0, s, c->d_partial, sg.theta, c->d_Wsplit,
                       sg.x, sg.xo, sg.nchains, sg.T, sg.cpg, sg.gstride, sg.cstride, sg.sstride, c->d.k, 2);
    hipLaunchKernelGGL(linear_reduce_kernel, dim3(1), dim3(256), 0, s, slot, c->d_partial + 1, blocks, 2, pass);
    HIP_TRY(c, hipGetLastError());
    pass++;
  }
  if (pass == 0) HIP_TRY(c, hipMemsetAsync(slot, 0, sizeof(double), s));
  return ADMM_OK;
}

// Completes the queued objective that still waits for its unary term.  Must
// run before anything changes the labels or the unaries.
static int flush_gather(admm_ctx *c);
static int flush_pending(admm_ctx *c) {
  int rc = flush_red(c);
  if (rc) return rc;
  if ((rc = settle_swept(c))) return rc;
  return flush_gather(c);
}
static int flush_gather(admm_ctx *c) {
  if (!c->pending_slot) return ADMM_OK;
  const int slot = c->pending_slot;
  c->pending_slot = 0;
  return chain_unary_gather(c, c->d_obj_slots + 2 * (size_t)slot);
}

int admm_linear_objective(admm_ctx *c, double *obj) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "objective before admm_init");
  if (!obj) return fail(c, ADMM_E_ARG, "null output");
  if (is_chain(c)) {
    if (!c->lin_valid) return chain_obj_partials(c, nullptr, obj);  // before the first Compute
    if ((rc = chain_pairwise_enqueue(c, c->d_obj_slots)) || (rc = chain_unary_gather(c, c->d_obj_slots))) return rc;
    double h[2];
    HIP_TRY(c, hipMemcpyAsync(h, c->d_obj_slots, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *obj = (h[0] + h[1]) * c->d.rho;
    return ADMM_OK;
  }
  if (c->edge_obj_n > 0) {  // the last solve summed its own marginals, workgroup by workgroup
    hipLaunchKernelGGL(edge_objective_sum_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_partial, c->d_edge_obj,
                       c->edge_obj_n);
    HIP_TRY(c, hipGetLastError());
    double h = 0.0;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_partial, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *obj = h * c->d.rho;
    return ADMM_OK;
  }
  const long long nn = (long long)c->E * 2 * c->d.k, ne = (long long)c->E * c->d.k * c->d.k;
  const int blocks = std::min(c->n_partial, grid_for(std::max(nn, ne), 256));
  hipLaunchKernelGGL(edge_objective_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_partial, c->d_pot_node,
                     c->d_mu_node, nn, c->d_pot_edge, c->d_mu_edge, ne);
  HIP_TRY(c, hipGetLastError());
  std::vector<double> h(blocks);
  HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_partial, sizeof(double) * blocks, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double s = 0.0;
  for (double v : h) s += v;
  *obj = s * c->d.rho;
  return ADMM_OK;
}

int admm_objective(admm_ctx *c, double *primal, double *dual) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "objective before admm_init");
  if (!is_chain(c)) {
    if (dual) *dual = std::nan("");
    return primal ? admm_linear_objective(c, primal) : ADMM_OK;
  }
  if (primal && (rc = chain_obj_partials(c, primal, nullptr))) return rc;
  if (dual) {
    if ((rc = chain_sweep(c, true))) return rc;
    const size_t n = (size_t)c->d.batch * ((size_t)c->st.nrows + c->st.ncols);
    std::vector<float> h(n);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_chain_min, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (float v : h) s += (double)v;
    *dual = s * c->d.rho;
  }
  return ADMM_OK;
}

int admm_objective_enqueue(admm_ctx *c, int slot) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!c->inited) return fail(c, ADMM_E_STATE, "objective before admm_init");
  if (!is_chain(c)) return fail(c, ADMM_E_STATE, "queued objectives are a chain-solver feature");
  if (slot < 1 || slot >= c->n_obj_slots) return fail(c, ADMM_E_ARG, "slot must be in [1, %d)", c->n_obj_slots);
  if (!c->lin_valid) return fail(c, ADMM_E_STATE, "admm_objective_enqueue needs a preceding admm_compute");
  if ((rc = flush_gather(c))) return rc;  // two requests for the same solve
  if ((rc = chain_pairwise_enqueue(c, c->d_obj_slots + 2 * (size_t)slot, true))) return rc;
  c->pending_slot = slot;  // the next sweep (or a read before it) supplies the unary term
  return ADMM_OK;
}

int admm_objective_read(admm_ctx *c, int slot, int n, double *linear) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!is_chain(c)) return fail(c, ADMM_E_STATE, "queued objectives are a chain-solver feature");
  if (slot < 1 || n < 0 || slot + n > c->n_obj_slots) return fail(c, ADMM_E_ARG, "slot range out of bounds");
  if ((rc = flush_red(c))) return rc;
  if (c->swept_slot >= slot && c->swept_slot < slot + n && (rc = settle_swept(c))) return rc;
  if (c->pending_slot >= slot && c->pending_slot < slot + n && (rc = flush_pending(c))) return rc;
  std::vector<double> h(2 * (size_t)n);
  HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_obj_slots + 2 * (size_t)slot, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++)
    if (linear) linear[i] = (h[2 * i] + h[2 * i + 1]) * c->d.rho;
  return ADMM_OK;
}

int admm_synchronize(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADMM_OK;
}

int admm_query(admm_ctx *c, int *busy) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!busy) return fail(c, ADMM_E_ARG, "null output");
  const hipError_t e = hipStreamQuery(c->stream);
  if (e != hipSuccess && e != hipErrorNotReady) return fail(c, ADMM_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
  *busy = e == hipErrorNotReady ? 1 : 0;
  return ADMM_OK;
}

int admm_finalize(admm_ctx *c) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  return ADMM_OK;
}

extern "C++" {
namespace {
template <typename TB>
__global__ void chain_decode_kernel(int *__restrict__ out, const label_t *__restrict__ xa,
                                    const TB *__restrict__ xbT, int nr, int nc, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long g = i / ((long long)nr * nc), rem = i % ((long long)nr * nc);
    const long long r = rem / nc, cc = rem % nc;
    out[i] = chain_decode(xa[i], xbT[g * nr * nc + cc * nr + r]);
  }
}
__global__ void transpose_int_kernel(int *__restrict__ out, const label_t *__restrict__ in, int nr, int nc,
                                     long long total) {
  // in: [g][nc][nr] -> out: [g][nr][nc]
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long g = i / ((long long)nr * nc), rem = i % ((long long)nr * nc);
    const long long r = rem / nc, cc = rem % nc;
    out[i] = in[g * nr * nc + cc * nr + r];
  }
}
__global__ void widen_labels_kernel(int *__restrict__ out, const label_t *__restrict__ in, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = in[i];
}
}  // namespace
}  // extern "C++"

// Decoded labelling of this strip into `stage` (device, `total` ints), then to the host.
static int decode_labels_into(admm_ctx *c, int *stage, int32_t *labels, long long total) {
  if (c->one_strip())
    hipLaunchKernelGGL(chain_decode_kernel<label_t>, dim3(grid_for(total, 256)), dim3(256), 0, c->stream, stage, c->xrow_r,
                       c->xcolT_c, c->st.nrows, c->d.cols, total);
  else
    hipLaunchKernelGGL(chain_decode_kernel<int>, dim3(grid_for(total, 256)), dim3(256), 0, c->stream, stage, c->xrow_r,
                       c->xcolT_r, c->st.nrows, c->d.cols, total);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(labels, stage, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADMM_OK;
}

int admm_download_labels(admm_ctx *c, int32_t *labels) {
  int rc = check_ctx(c);
  if (rc) return rc;
  StageRelease release(c);
  if (!is_chain(c)) return fail(c, ADMM_E_STATE, "the edge solver decodes no labels");
  if (!c->inited) return fail(c, ADMM_E_STATE, "download before admm_init");
  if (!labels) return fail(c, ADMM_E_ARG, "null output");
  const long long total = (long long)c->d.batch * c->st.nrows * c->d.cols;
  if (total == 0) return ADMM_OK;
  void *stage = nullptr;
  if ((rc = stage_buf(c, &stage, sizeof(int) * (size_t)total))) return rc;
  return decode_labels_into(c, (int *)stage, labels, total);
}

int admm_download_labels_static(admm_ctx *c, int32_t *labels) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!is_chain(c)) return fail(c, ADMM_E_STATE, "the edge solver decodes no labels");
  if (!c->inited) return fail(c, ADMM_E_STATE, "download before admm_init");
  if (!labels) return fail(c, ADMM_E_ARG, "null output");
  const long long total = (long long)c->d.batch * c->st.nrows * c->d.cols;
  if (total > (long long)ADMM_STATIC_LABELS_MAX)
    return fail(c, ADMM_E_ARG, "admm_download_labels_static: this strip has %lld labels, the static staging holds %lld; use admm_download_labels",
                total, (long long)ADMM_STATIC_LABELS_MAX);
  const int dev = c->d.device;
  std::string why;
  int32_t *stage = scratch_for(dev, &why);  // (check_ctx made the context's device current)
  if (!stage)
    return fail(c, ADMM_E_HIP, "admm_download_labels_static: no static staging on device %d (%s)", dev, why.c_str());
  if (total == 0) return ADMM_OK;
  // the decode kernel writes every element it later copies out: nothing of an
  // earlier call's contents is read
  return decode_labels_into(c, stage, labels, total);
}

int admm_download_consensus(admm_ctx *c, float *out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  StageRelease release(c);
  if (!c->inited) return fail(c, ADMM_E_STATE, "download before admm_init");
  if (!out) return fail(c, ADMM_E_ARG, "null output");
  if (!is_chain(c)) {
    HIP_TRY(c, hipMemcpyAsync(out, c->d_m_vert, sizeof(float) * (size_t)c->V * c->d.k, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ADMM_OK;
  }
  // the k-vectors exist only here: built from the labels on request
  const size_t n = (size_t)c->d.batch * c->st.nrows * c->d.cols * c->d.k;
  if (n == 0) return ADMM_OK;
  void *stage = nullptr;
  if ((rc = stage_buf(c, &stage, sizeof(float) * n))) return rc;
  float *tmp = (float *)stage;
  rc = c->one_strip() ? chain_consensus<label_t>(c, tmp, c->xrow_r, c->xcolT_c, c->st.nrows, c->d.cols)
                      : chain_consensus<int>(c, tmp, c->xrow_r, c->xcolT_r, c->st.nrows, c->d.cols);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(out, tmp, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ADMM_OK;
}

size_t admm_state_size(const admm_ctx *c, int which) {
  if (!c) return 0;
  const size_t k = c->d.k;
  if (!is_chain(c)) {
    if (which >= 0 && which <= 3) return (size_t)c->E * 2 * k;
    if (which == 4 || which == 5) return (size_t)c->E * k * k;
    return 0;
  }
  const size_t nrow = (size_t)c->d.batch * c->st.nrows * c->d.cols, ncol = (size_t)c->d.batch * c->d.rows * c->st.ncols;
  switch (which) {
    case 10: case 12: return nrow * k;
    case 11: return ncol * k;
    case 13: return nrow;
    case 14: return ncol;
  }
  return 0;
}

int admm_download_state(admm_ctx *c, int which, float *out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  StageRelease release(c);
  if (!c->inited) return fail(c, ADMM_E_STATE, "download before admm_init");
  if (!out) return fail(c, ADMM_E_ARG, "null output");
  const size_t n = admm_state_size(c, which);
  if (n == 0) return fail(c, ADMM_E_ARG, "unknown state id %d for this solver", which);
  const void *src = nullptr;
  void *tmp = nullptr;
  if (!is_chain(c)) {
    switch (which) {
      case 0: src = c->d_mu_node; break;
      case 1: src = c->d_pot_node; break;
      case 3: src = c->d_lam_node; break;
      case 4: src = c->d_mu_edge; break;
      case 5: src = c->d_pot_edge; break;
      case 2:
        if ((rc = stage_buf(c, &tmp, sizeof(float) * n))) return rc;
        hipLaunchKernelGGL(edge_gather_m_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, c->stream,
                           (float *)tmp, c->d_m_vert, c->d_edges, c->E, c->d.k);
        src = tmp;
        break;
    }
  } else {
    switch (which) {
      case 10: case 11:
        if (c->lam16) {
          if ((rc = stage_buf(c, &tmp, sizeof(float) * n))) return rc;
          launch_widen_f32(c, (float *)tmp, (which == 10 || c->lam_shared()) ? c->lam_r : c->lam_c, (long long)n);
          src = tmp;
        } else {
          src = (which == 10 || c->lam_shared()) ? c->lam_r : c->lam_c;
        }
        break;
      case 12: src = c->th_r; break;
      case 13:
        if ((rc = stage_buf(c, &tmp, sizeof(int) * n))) return rc;
        hipLaunchKernelGGL(widen_labels_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, c->stream,
                           (int *)tmp, c->xrow_r, (long long)n);
        src = tmp;
        break;
      case 14:
        // stored chain-major ([col][row]); hand it out as [row][col]
        if ((rc = stage_buf(c, &tmp, sizeof(int) * n))) return rc;
        hipLaunchKernelGGL(transpose_int_kernel, dim3(grid_for((long long)n, 256)), dim3(256), 0, c->stream,
                           (int *)tmp, c->xcolT_c, c->d.rows, c->st.ncols, (long long)n);
        src = tmp;
        break;
    }
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, src, 4 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // the column multipliers are kept as the row ones negated (never -0: a sum
  // that starts from +0 cannot round to it, so 0 - x restores the exact bits)
  if (is_chain(c) && which == 11 && c->lam_shared())
    for (size_t i = 0; i < n; i++) out[i] = 0.0f - out[i];
  return ADMM_OK;
}

int admm_comm_get_unique_id(void *id) {
  if (!id) return ADMM_E_ARG;
  static_assert(sizeof(ncclUniqueId) <= ADMM_COMM_ID_BYTES, "id buffer too small");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return fail(nullptr, ADMM_E_COMM, "ncclGetUniqueId: %s", ncclGetErrorString(r));
  memset(id, 0, ADMM_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return ADMM_OK;
}

int admm_comm_init(admm_ctx *c, const void *id) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!id) return fail(c, ADMM_E_ARG, "null id");
  if (c->comm) return fail(c, ADMM_E_STATE, "communicator already initialised");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = ncclCommInitRank(&comm, c->d.nranks, u, c->d.rank);
  if (r != ncclSuccess) {
    // Whatever a failed bootstrap left in `comm` is not a communicator: it is
    // dropped here, never handed to ncclCommDestroy (which frees it a second
    // time and aborts the process at teardown).
    c->comm = nullptr;
    return fail(c, ADMM_E_COMM, "ncclCommInitRank(nranks=%d, rank=%d) failed: %s", c->d.nranks, c->d.rank, ncclGetErrorString(r));
  }
  c->comm = comm;
  return ADMM_OK;
}

int admm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, ADMM_E_ARG, "device %d out of range (%d devices)", device, ndev);
  size_t f = 0, t = 0;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMemGetInfo(&f, &t);
  if (e != hipSuccess) return fail(nullptr, ADMM_E_HIP, "hipMemGetInfo failed: %s", hipGetErrorString(e));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return ADMM_OK;
}

int admm_profile_reset(admm_ctx *c, int enable) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->prof = enable > 0;
  c->prof_every = std::max(1, enable);
  c->prof_seen = 0;
  c->prof_used = 0;
  return ADMM_OK;
}

int admm_sweep_kernel_info(admm_ctx *c, int *label_lanes, int *lanes_per_label, int *potts_out) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!is_chain(c)) return fail(c, ADMM_E_ARG, "the edge solver has no sweep kernel");
  if (!c->inited) return fail(c, ADMM_E_STATE, "admm_sweep_kernel_info before admm_init");
  SweepShape sh = pick_shape(c->KP, (long long)c->d.batch * ((long long)c->st.nrows + c->st.ncols));
  const bool potts = potts_chosen(c, sh);
  if (potts || trunc_chosen(c, sh)) sh.NS = 1;
  if (label_lanes) *label_lanes = sh.KP;
  if (lanes_per_label) *lanes_per_label = sh.NS;
  if (potts_out) *potts_out = potts ? 1 : 0;
  return ADMM_OK;
}

int admm_sweep_kernel_name(admm_ctx *c, char *buf, int len) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!is_chain(c)) return fail(c, ADMM_E_ARG, "the edge solver has no sweep kernel");
  if (!c->inited) return fail(c, ADMM_E_STATE, "admm_sweep_kernel_name before admm_init");
  if (!buf || len < 1) return fail(c, ADMM_E_ARG, "null buffer");
  SweepShape sh = pick_shape(c->KP, (long long)c->d.batch * ((long long)c->st.nrows + c->st.ncols));
  const bool potts = potts_chosen(c, sh);
  if (potts) {
    // the same count of wavefronts launch_sweep works out
    const int CPW = 64 / sh.KP;
    const int waves = 2 * ((c->d.batch * c->st.nrows + CPW - 1) / CPW) + 2 * ((c->d.batch * c->st.ncols + CPW - 1) / CPW);
    const bool two = twowave_chosen(sh.KP, waves);
    snprintf(buf, (size_t)len, "chain_sweep_kernel<%d, 1, %d, %d, false, true, %s, %s, %s, %s, 0>", sh.KP, two ? kSplitGpb : potts_gpb(sh.KP),
             two ? (split_lone(waves) ? 2 : 3) : 4, c->lam16 ? "true" : "false",
             (!two && pcomp_chosen(sh.KP, waves)) ? "true" : "false", (c->lam16 && c->lam8) ? "true" : "false", two ? "true" : "false");
  } else if (trunc_chosen(c, sh))
  {
    const int waves = 2 * ((c->d.batch * c->st.nrows + 3) / 4) + 2 * ((c->d.batch * c->st.ncols + 3) / 4);
    snprintf(buf, (size_t)len, "chain_sweep_kernel<16, 1, 8, 4, false, false, %s, false, %s, %s, 3>", c->lam16 ? "true" : "false",
             (c->lam16 && c->lam8) ? "true" : "false", twowave_chosen(16, waves, true) ? "true" : "false");
  }
  else
    snprintf(buf, (size_t)len, "chain_sweep_kernel<%d, %d, %d, %d, false, false, %s, false, %s, false, 0>", sh.KP, sh.NS, sweep_gpb(sh.KP, sh.NS), sweep_minw(sh.KP, sh.NS),
             c->lam16 ? "true" : "false", (c->lam16 && c->lam8) ? "true" : "false");
  return ADMM_OK;
}

int admm_launch_info(admm_ctx *c, char *buf, int len) {
  int rc = check_ctx(c);
  if (rc) return rc;
  if (!buf || len < 1) return fail(c, ADMM_E_ARG, "null buffer");
  snprintf(buf, (size_t)len, "%s", c->launch_info.c_str());
  return ADMM_OK;
}

int admm_profile_read(admm_ctx *c, double *ms_avg, int *launches) {
  int rc = check_ctx(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double tot = 0.0;
  for (size_t i = 0; i < c->prof_used; i++) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
    tot += ms;
  }
  if (ms_avg) *ms_avg = c->prof_used ? tot / (double)c->prof_used : 0.0;
  if (launches) *launches = (int)c->prof_used;
  return ADMM_OK;
}

}  // extern "C"
