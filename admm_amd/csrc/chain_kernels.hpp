// Chain (min-sum / Viterbi) solver kernels for gfx950.
//
// Trees are the row chains and the column chains of the grid.  A wavefront
// sweeps half a chain with the label dimension across lanes (several
// short-label half chains share a wavefront when k is small), and the two
// wavefronts of a chain meet in its middle.  The ADMM shell around the sweep
// keeps the reference's formulas and operation order:
//   dual update      lambda += penalty * (mu - m)                  Tree.cpp:70
//   potential adjust t = rho*(theta + lambda) - penalty*(m - mu)   Tree.cpp:85-86
//   consensus        m = (mu_row + mu_col) / 2                     Grid.cpp:176-180
// mu is the one-hot vector of a chain's labels, so the consensus of a node is
// a function of the two labels it received; the sweep rebuilds it from them
// instead of reading a k-vector per node (the k-vector is only written out
// when a caller asks for it).  Only mu - m enters the formulas, and before
// the first solve mu = m = 1/k (Tree.cpp:24-33), i.e. mu - m = 0: the label
// -1 stands for that state and matches no lane.
// The sweep itself has no counterpart in the reference, whose only subproblem
// solver is the per-edge one in Tree.cpp.  A chain of T nodes is swept from
// both ends towards the middle by two wavefronts of one workgroup: with
// L = (T + 1) / 2,
//   left   A_0 = t_0,          A_i[b] = t_i[b] + min_a (A_{i-1}[a] + W[a][b]),  i < L
//   right  B_{T-1} = t_{T-1},  B_j[a] = t_j[a] + min_b (B_{j+1}[b] + W[a][b]),  j >= L
//   join   J[a] = A_{L-1}[a] + min_b (B_L[b] + W[a][b])
// every minimum remembering the lowest index that attains it.  x_{L-1} is the
// lowest argmin of J and x_L its remembered b; the left half is traced back
// from x_{L-1}, the right half forward from x_L.  The right half is the same
// recursion as the left one on the reversed chain with the transposed table,
// so one code path serves both; the halves meet through LDS and two workgroup
// barriers.  Halving the serial length of a chain is what keeps the machine
// busy when chains are scarce (small grids, one strip of many).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace admm {

// A chain's label for a node: 0 .. k-1 with k <= 64, or -1 before the first solve.
// One byte, so that the label arrays cost the sweep a quarter of what int32 would.
typedef int8_t label_t;

// One set of chains that share a stride pattern (all row chains, or all
// column chains).  Node t of chain c is
//   (c / cpg) * gstride + (c % cpg) * cstride + t * sstride,
// the k floats of a node are contiguous, and the labels of a chain are
// contiguous in x (chain-major).
struct ChainSeg {
  const float *theta;  // per-copy share of the unaries
  const float *lam;    // Lagrange multipliers of these chains, as read ...
  float *lam_out;      // ... and where the updated ones go (may be lam itself); null: not stored
                       // (kernels built with LAM16 read and write int16 counts through both, see there)
  unsigned lam_sign;   // 0, or 0x80000000: lam holds the multipliers negated.  The two copies of a node
                       // get opposite increments (mu_row - m = -(mu_col - m)), so starting from zero the
                       // column multipliers are the row ones negated, bit for bit; on one device the
                       // column chains read the row chains' array and store nothing, and the row chains
                       // write into a second buffer that becomes current after the sweep.
  label_t *x;          // [nchains][T] labels of these chains as the sweep writes them ...
  const label_t *xin;  // ... and the ones it starts from (-1 before the first solve); may be x itself
  const label_t *xo;   // labels the crossing chains gave the same nodes: node p of chain c at
                       //   (c / cpg) * xo_gstride + (c % cpg) * xo_cstride + p * xo_sstride.
                       // Either laid out along these chains ([nchains][T]: xo_cstride = T, xo_sstride = 1,
                       // what Update's transposition leaves), or the crossing chains' own array as they
                       // wrote it ([T][cpg] per grid: xo_cstride = 1, xo_sstride = cpg), read with a stride.
  long long xo_gstride, xo_cstride, xo_sstride;
  uint8_t *bp;         // [nchains][2][half length padded to 64][KP] back-pointer scratch
  float *chain_min;    // [nchains] (dual evaluation only)
  double *chain_lin;   // [nchains][2] pairwise energy of the new labels per half chain: sum of (W / rho)[x_t][x_t+1]
                       //              (the edge between the halves counts with the right half)
  double *chain_unary; // [nchains][2] unary energy of the labels the sweep started from: sum of theta_share[x_t]
  int nchains;
  int T;
  int cpg;
  long long gstride, cstride, sstride;
};

struct ChainArgs {
  ChainSeg seg[2];
  int nseg;
  int wave_begin[3];  // waves [wave_begin[i], wave_begin[i+1]) work on seg[i]; even, since
                      // waves 2p and 2p+1 are the left and the right half of the same chains
  int k;
  float penalty, rho, inv_k;
  const float *Wt;     // k*k, rho * (W / rho)
  const float *Wsplit; // k*k, W / rho
  const float *WtT, *WsplitT;  // their transposes, for the right halves
  unsigned long long *stamps;  // diagnostics: 3 clock reads per wavefront, or null
  // Two sets of chains: the first mix_blocks workgroups hold one pair of
  // wavefronts of seg[0] and one of seg[1] each, so that every compute unit runs
  // the same blend of the two sets whatever order the workgroups are handed
  // out in; the workgroups after them hold two pairs of seg[mix_big], the set
  // that has pairs left over.  0: workgroups are filled set after set.
  int mix_blocks, mix_big;
  // Sums of per-half-chain objective terms an earlier sweep left (up to two arrays of
  // binary64 terms, each into one binary64 word), done by workgroup red_block of this
  // launch, which then takes no part in the sweep.  red_jobs = 0: none.
  double *red_dst[2];
  const double *red_part[2];
  int red_n[2];
  int red_jobs, red_block;  // red_block is also the number of workgroups that sweep
  // Workgroups are handed to the eight dies in turn.  xcd_remap: workgroup b works on the
  // chains of workgroup (b % 8) * (n / 8) + b / 8 instead, so that every die holds a contiguous
  // range of each set's chains (chains that lie side by side read the same cache lines of the
  // crossing chains' labels, and every die has a cache of its own).
  int xcd_remap;
  // Mixed workgroups only.  The memory system moves whole 128-byte lines; where the chains of
  // one wavefront of seg[1] use less than a line of an operand per step (two adjacent column
  // chains of 32 int8 counts: 64 bytes), the other part of the line belongs to the next
  // wavefront pair of the set.  pair_xcd: the pairs 2j and 2j + 1 of seg[1] go to workgroups
  // eight apart, i.e. to the same die a moment apart, so that the second one finds the line
  // in that die's cache instead of fetching it from memory again.
  int pair_xcd;
};

constexpr int kChainBlock = 256;  // two pairs of wavefronts per workgroup (a pair = the two halves of the same chains)
constexpr int kChunk = 64;        // steps whose back-pointers are staged in LDS

__device__ __forceinline__ float chain_mu(int x, int l, float inv_k) {
  return x < 0 ? inv_k : (x == l ? 1.0f : 0.0f);
}

// Compiler-level ordering between LDS accesses of different lanes of one
// wavefront.  The hardware executes a wavefront's LDS instructions in issue
// order; this only stops the compiler from moving them across the point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Compile-time loop: f(integral_constant<int, I>) for I in [0, N).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Minimum of c[0..N) with three-input minima.
template <int N>
__device__ __forceinline__ float min_of(const float *c) {
  float r = c[0];
  if constexpr (N % 2 == 0) {
    r = __builtin_fminf(r, c[1]);
#pragma unroll
    for (int q = 2; q + 1 < N; q += 2) r = __builtin_fminf(__builtin_fminf(r, c[q]), c[q + 1]);
  } else {
#pragma unroll
    for (int q = 1; q + 1 < N; q += 2) r = __builtin_fminf(__builtin_fminf(r, c[q]), c[q + 1]);
  }
  return r;
}

// V0 * [c0 == best] else V1 * [c1 == best] else V2 * [c2 == best] else V3: the
// first of four candidates that equals the minimum, as a scaled index.  Three
// compares into three masks, then three selects: written out because a select
// may not read a mask in the two issue slots after the compare that writes it,
// and the compiler pairs every compare with its select and pads with s_nop.
template <int V0, int V1, int V2, int V3>
__device__ __forceinline__ int first_of_four(const float c0, const float c1, const float c2, const float best) {
  int idx;
  unsigned long long m1, m2;
  asm("v_cmp_eq_f32_e64 %[m2], %[c2], %[best]\n\t"
      "v_cmp_eq_f32_e64 %[m1], %[c1], %[best]\n\t"
      "v_cmp_eq_f32_e32 vcc, %[c0], %[best]\n\t"
      "v_cndmask_b32_e64 %[idx], %[v3], %[v2], %[m2]\n\t"
      "v_cndmask_b32_e64 %[idx], %[idx], %[v1], %[m1]\n\t"
      "v_cndmask_b32_e64 %[idx], %[idx], %[v0], vcc"
      : [idx] "=&v"(idx), [m1] "=&s"(m1), [m2] "=&s"(m2)
      : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [best] "v"(best), [v0] "n"(V0), [v1] "n"(V1), [v2] "n"(V2), [v3] "n"(V3)
      : "vcc");
  return idx;
}

// Lowest index i in [0, N) with c[i] == best, given that one exists.  A flat
// scan costs a compare and a select per candidate; halving the candidate set
// on "is the minimum in the lower half" costs one compare per level and a
// select per surviving candidate, about two thirds of that.
// HAND: the last four go through first_of_four (pays with one wavefront per
// SIMD, costs with two: the group scheme's vector loads get split).
template <int N, bool HAND = false>
__device__ __forceinline__ int first_equal(const float *c, const float best) {
  if constexpr (N == 4 && HAND) {
    return first_of_four<0, 1, 2, 3>(c[0], c[1], c[2], best);
  } else if constexpr (N <= 4) {
    int idx = N - 1;
#pragma unroll
    for (int q = N - 2; q >= 0; q--) idx = (c[q] == best) ? q : idx;
    return idx;
  } else {
    constexpr int H = N / 2;
    const bool in_lo = min_of<H>(c) == best;
    float sel[H];
#pragma unroll
    for (int q = 0; q < H; q++) sel[q] = in_lo ? c[q] : c[H + q];
    const int r = first_equal<H, HAND>(sel, best);
    return r | (in_lo ? 0 : H);
  }
}

// The same with the result scaled: SC * (lowest index i with c[i] == best).
template <int N, int SC>
__device__ __forceinline__ int first_equal_sc(const float *c, const float best) {
  if constexpr (N <= 4) {
    int idx = SC * (N - 1);
#pragma unroll
    for (int q = N - 2; q >= 0; q--) idx = (c[q] == best) ? SC * q : idx;
    return idx;
  } else {
    constexpr int H = N / 2;
    const bool in_lo = min_of<H>(c) == best;
    float sel[H];
#pragma unroll
    for (int q = 0; q < H; q++) sel[q] = in_lo ? c[q] : c[H + q];
    return first_equal_sc<H, SC>(sel, best) | (in_lo ? 0 : SC * H);
  }
}
// ... and with an offset folded into the top-level select: lo + SC * index,
// given hi = lo + SC * N / 2 and lo a multiple of SC * N.
template <int N, int SC>
__device__ __forceinline__ int first_equal_sc_base(const float *c, const float best, const int lo, const int hi) {
  static_assert(N >= 8, "the offset rides on the first halving");
  constexpr int H = N / 2;
  const bool in_lo = min_of<H>(c) == best;
  float sel[H];
#pragma unroll
  for (int q = 0; q < H; q++) sel[q] = in_lo ? c[q] : c[H + q];
  return first_equal_sc<H, SC>(sel, best) | (in_lo ? lo : hi);
}

// First of four: v[i] for the lowest i with c[i] == best (the values are
// registers, so an offset costs nothing).
__device__ __forceinline__ int first_equal4_vals(const float *c, const float best, const int *v) {
  int idx = v[3];
#pragma unroll
  for (int q = 2; q >= 0; q--) idx = (c[q] == best) ? v[q] : idx;
  return idx;
}

// Minimum of x over each aligned group of W lanes, in every lane of the group.
// Butterfly on the cross-lane data path: quad swaps, the two row mirrors, then
// the row and half swaps.  (Mirrors instead of xor 4 and xor 8: once the groups
// of 4, of 8, hold one value any partner from the other group will do.)
// Written out because the compiler cannot see that the bits a cross-lane move
// returns are ordinary floats and quiets them before every minimum; the s_nop
// are the two wait states between a vector write and a cross-lane read of the
// same register, which nothing inserts inside an asm statement.
template <int W>
__device__ __forceinline__ float chain_lanes_min(float x) {
  static_assert(W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group of lanes");
  float r;
  asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  if constexpr (W >= 4) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(r));
  if constexpr (W >= 8) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(r));
  if constexpr (W >= 16) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r));
  if constexpr (W >= 32) {
    float t;
    asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_min_f32_e32 %0, %0, %1" : "+v"(r), "=&v"(t));
  }
  if constexpr (W >= 64) {
    float t;
    asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_min_f32_e32 %0, %0, %1" : "+v"(r), "=&v"(t));
  }
  return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// dst[0] = sum of part[0 .. n), by one workgroup of 256 threads, in the association of a
// workgroup of 1024 threads (thread t: four running sums over part[t + 1024 j], j = 0 .. 3
// modulo 4, their pairwise sum, a butterfly over the 64 lanes of its wavefront, then the 16
// wavefronts in order): wavefront r stands in for the wavefronts r, r + 4, r + 8, r + 12 of
// that workgroup, so the result has the same bits whichever launch computes it.
__device__ __forceinline__ void sum_terms_1024(double *__restrict__ dst, const double *__restrict__ part, const int n, double *s_red) {
  const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
#pragma unroll 1
  for (int v = 0; v < 4; v++) {
    const int w = r + 4 * v;  // the wavefront stood in for
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
    int i = w * 64 + lane;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
      l0 += part[i]; l1 += part[i + 1024]; l2 += part[i + 2048]; l3 += part[i + 3072];
    }
    for (; i < n; i += 1024) l0 += part[i];
    const double l = wave_sum((l0 + l1) + (l2 + l3));
    if (lane == 0) s_red[w] = l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; w++) t += s_red[w];
    dst[0] = t;
  }
  __syncthreads();
}

// Lowest bit of every W-bit field of a 64-bit lane mask.
template <int W>
constexpr unsigned long long kFieldLsb = W >= 64 ? 1ull : (~0ull / ((1ull << (W & 63)) - 1));

// KP   : labels padded to a power of two (lanes b >= k idle with M = +inf)
// NS   : lanes that share one target label; lane s of them scans the source
//        labels [s*KP/NS, (s+1)*KP/NS) and holds the node operands of the
//        steps t with t % NS == s, so one 64-lane load covers NS steps
// GPB  : load groups (of NS steps) per unrolled block = prefetch distance
// MINW : wavefronts per SIMD the register allocation must leave room for
// DUAL : evaluate the chain minimum under theta + lambda only (no dual
//        update, no back-pointers, no labels)
// TRUNC: n > 0: the step specialised for tables W[a][b] = g(|a - b|) with g constant from
//        distance n + 1 on and nowhere above that constant (truncated linear, truncated
//        quadratic, Potts ...): the 2n + 1 nearest sources through DPP row shifts of the
//        register-resident messages plus one chain-wide minimum, see there
// SPLIT: two wavefronts per half chain.  One (the shell) loads the operands, does the dual
//        update and the potential adjust, stores the multipliers and sums the unary term of
//        the objective; the other (the recursion) takes the adjusted unaries from a ring in
//        LDS and does the min-plus step, the back-pointers and the trace-back.  A wavefront
//        that has its SIMD to itself is bound by the instructions it issues, so halving a
//        step's instructions per wavefront is what speeds a launch with few chains up.
// PCOMP: Potts step with path compression for the trace-back (see there)
// LAM8 : with LAM16, the counts are int8: the host runs these kernels while it knows
//        every count to be far from +-127 (it scans the arrays every few sweeps)
// LAM16: the multiplier arrays hold int16 counts n with lambda = n * penalty / 2.
//        Every update adds -penalty/2, 0 or +penalty/2, so when penalty is a power
//        of two and the counts stay below 2^15 the binary32 sums are exact multiples
//        and the count is the same number in half of the bytes.  The
//        host picks these kernels only under those conditions and turns the counts
//        into binary32 before they could overflow.
//
// Back-pointers: byte (t, b) = lowest source label attaining the minimum for
// target b at step t.  They are packed four steps to a dword and written
// straight from registers, dword (t/4)*KP + b of the half chain's scratch;
// the backtrack streams them back one 64-step chunk at a time (t counts the
// steps of the half chain in sweep order).
template <int KP, int NS, int GPB, int MINW, bool DUAL, bool POTTS = false, bool LAM16 = false, bool PCOMP = false, bool LAM8 = false, bool SPLIT = false, int TRUNC = 0>
__global__ __launch_bounds__(kChainBlock, MINW) void chain_sweep_kernel(const ChainArgs args) {
  static_assert(!PCOMP || POTTS, "path compression belongs to the Potts step");
  static_assert(!LAM8 || LAM16, "LAM8: the counts of LAM16, one byte each");
  static_assert(!SPLIT || !DUAL, "the dual evaluation is never split over two wavefronts");
  static_assert(TRUNC == 0 || (!POTTS && !DUAL && NS == 1 && KP <= 16 && TRUNC <= 7),
                "the truncated step shifts messages inside a row of sixteen lanes");
  typedef typename std::conditional<LAM8, int8_t, int16_t>::type count_t;
  constexpr unsigned CB = sizeof(count_t);  // bytes per count
  static_assert(!POTTS || (NS == 1 && !DUAL), "the Potts step keeps a label's message in its one lane");
  static_assert(!LAM16 || (NS == 1 && !DUAL), "int16 multipliers: one step per load group, never the dual evaluation");
  constexpr int LPC = KP * NS;   // lanes per chain
  constexpr int CPW = 64 / LPC;  // chains per wavefront
  // trace-back on the scalar unit with the back-pointers in registers (see there; with four
  // chains the walk through LDS, one lane per chain, is the faster one: 29 k against 37 k clock
  // ticks for 256 steps)
  constexpr bool SWALK = CPW <= 2;
  constexpr int A = KP / NS;     // source labels scanned per lane
  constexpr int BLK = GPB * NS;  // steps per unrolled block
  constexpr int WIN = 4 * NS;    // steps per back-pointer store (one dword per lane)
  constexpr int DW = 16 / NS;    // back-pointer dwords per lane per chunk
  static_assert(LPC <= 64 && A >= 1 && NS <= 8, "bad lane layout");
  static_assert(BLK % WIN == 0 && kChunk % BLK == 0, "blocks must tile windows and chunks");
  // Work out a step's arg-min one step later (see below)?  It needs a second
  // set of candidate registers, which 64 candidates per lane do not leave.
  // Many candidates per lane: keep only the minima of groups of four and, once
  // the group that holds the first minimum is known, form its four candidates
  // again from LDS (messages of the step, pairwise column of the lane).  That
  // replaces a select per candidate per halving level with two LDS reads.  The
  // messages of a step must then outlive the step: two alternating rows.
  // (tried for the split k = 32 kernel too, 16 candidates per lane: slower, its
  // single wavefront per SIMD feels the extra LDS traffic)
  constexpr bool RECOMP = !POTTS && !DUAL && NS == 1 && A >= 32 && TRUNC == 0;
  constexpr int NG = A / 4;      // candidate groups per lane
  constexpr int NB = RECOMP ? 2 : 1;
  constexpr int WLD = KP + 4;    // padded row of the LDS pairwise table
  constexpr bool PIPE = !POTTS && TRUNC == 0 && (RECOMP || A <= 32);
  constexpr int WPB = kChainBlock / 64;
  constexpr int MROW = CPW * KP + 64;

  __shared__ __attribute__((aligned(16))) float s_M[WPB][NB * MROW];  // messages + a dump row
  __shared__ __attribute__((aligned(16))) float s_W[2][RECOMP ? KP * WLD : 4];  // per direction: [target][source]
  __shared__ float s_join[WPB / 2][CPW * KP];  // last message of the left halves
  __shared__ int s_joinx[WPB / 2][CPW];        // label of the left halves' last node, from the join
  __shared__ __attribute__((aligned(16))) int s_x[WPB][CPW][kChunk];
  __shared__ __attribute__((aligned(16))) int s_xo[WPB][CPW][kChunk];
  __shared__ __attribute__((aligned(16))) uint32_t s_bp[WPB][CPW][PCOMP ? 1 : 16 * KP];
  // Trace-back of the Potts step (see there): a round's step records, transposed so that one
  // lane per block of eight steps reads its block's, then the round's labels
  constexpr int TB_RSTEPS = KP == 64 ? 256 : 512 / CPW;     // steps of every chain that one round covers
  constexpr int TB_NBR = TB_RSTEPS / 8;                      // blocks of eight steps per chain and round
  constexpr int TB_NW = CPW * TB_NBR;                        // walking lanes: 32 (64 labels) or 64
  constexpr int TB_SW = TB_NW == 64 ? 72 : 40;               // row pitch that keeps both the transposed writes and the reads free of bank conflicts
  constexpr int TB_NA = KP == 64 ? 3 : (KP == 32 ? 2 : 1);   // dwords per record: keep field (one or two), i1 (packed into the field's dword up to 16 labels)
  constexpr int TB_W = TB_NA * 8 * TB_SW;
  static_assert(!PCOMP || TB_W >= CPW * TB_RSTEPS, "the round's labels reuse the record staging");
  __shared__ __attribute__((aligned(16))) uint32_t s_tb[WPB][PCOMP ? TB_W : 4];

  // SPLIT: adjusted unaries on their way from the shell to the recursion wavefront, three
  // blocks of GPB load groups per half chain (written, waiting, being read)
  __shared__ float s_ring[SPLIT ? 2 : 1][SPLIT ? 3 * GPB : 1][64];

  if (args.red_jobs > 0 && (int)blockIdx.x == args.red_block) {
    // the spare workgroup: sums an earlier sweep's objective terms (see ChainArgs)
    __shared__ double s_red[16];
    for (int j = 0; j < args.red_jobs; j++) sum_terms_1024(args.red_dst[j], args.red_part[j], args.red_n[j], s_red);
    return;
  }
  const int lane = threadIdx.x & 63;
  // the wavefront index is the same in all 64 lanes: keep it (and what
  // derives from it) in scalar registers
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int blk_id = blockIdx.x;
  if (args.xcd_remap) {
    const int per = args.red_block >> 3;
    if (blk_id < (per << 3)) blk_id = (blk_id & 7) * per + (blk_id >> 3);
  }
  int wave = blk_id * WPB + wib;
  // SPLIT: a workgroup holds the two halves of one group of chains, two wavefronts each
  // (role 0: recursion, role 1: shell); the host fills workgroups set after set
  const int role = SPLIT ? (wib & 1) : 0;
  if constexpr (SPLIT) wave = blk_id * 2 + (wib >> 1);
  int si;
  if (!SPLIT && args.mix_blocks > 0) {
    static_assert(WPB == 4, "a workgroup is two pairs of wavefronts");
    const int blk = blk_id;
    int wl;
    if (blk < args.mix_blocks) {
      si = wib >> 1;
      int p = blk;
      if (si == 1 && args.pair_xcd && blk < (args.mix_blocks & ~15)) {
        // workgroup 8q + x (die x): pair 16 (q / 2) + 2x + q % 2, a bijection of every 16 workgroups
        const int q = blk >> 3, x = blk & 7;
        p = ((q >> 1) << 4) + 2 * x + (q & 1);
      }
      wl = 2 * p + (wib & 1);
    } else {
      si = args.mix_big;
      wl = 2 * args.mix_blocks + (blk - args.mix_blocks) * WPB + wib;
    }
    if (wl >= args.wave_begin[si + 1] - args.wave_begin[si]) return;
    wave = args.wave_begin[si] + wl;
  } else {
    if (wave >= args.wave_begin[args.nseg]) return;
    si = (args.nseg > 1 && wave >= args.wave_begin[1]) ? 1 : 0;
  }
  const ChainSeg &sg = args.seg[si];

  unsigned long long st0 = 0;
  if (args.stamps) st0 = __builtin_readcyclecounter();
  const int cs = lane / LPC;
  const int within = lane % LPC;
  const int s = within / KP;
  const int b = within % KP;
  const int k = args.k;
  // waves 2p, 2p+1 of a segment: left and right half of the chains of group p
  const int wlocal = wave - args.wave_begin[si];
  const int rev = wlocal & 1;   // 1: the right half, swept from the chain's last node backwards
  const int pair = SPLIT ? 0 : wib >> 1;  // wave_begin is even, so the two halves sit in one workgroup
  const int chain_raw = (wlocal >> 1) * CPW + cs;
  const bool cvalid = chain_raw < sg.nchains;
  const int chain = cvalid ? chain_raw : sg.nchains - 1;
  const int Tfull = sg.T;
  const int Lhalf = (Tfull + 1) >> 1;
  const int T = rev ? Tfull - Lhalf : Lhalf;  // steps of this half (the right half of a 1-node chain has none)
  const int xdir = rev ? -1 : 1;
  const bool bvalid = b < k;
  const bool lvalid = cvalid && bvalid;
  // first node of this half's sweep: the chain's node 0, or its node T-1
  const long long half_first = rev ? (long long)(Tfull - 1) * sg.sstride : 0;
  const long long first = (long long)(chain / sg.cpg) * sg.gstride + (long long)(chain % sg.cpg) * sg.cstride + half_first;
  // the same for the wavefront's first chain: wave-uniform, no later chain lies below it
  const int chain0 = (wlocal >> 1) * CPW;
  const long long first0 = (long long)(chain0 / sg.cpg) * sg.gstride + (long long)(chain0 % sg.cpg) * sg.cstride + half_first;
  const float INF = __builtin_inff();
  const float pen = args.penalty, rho = args.rho;
  // mu - m is (mu_own - mu_other) / 2, one of -1/2, 0, +1/2, so penalty * (mu - m)
  // is +-hpen or 0 with hpen = penalty / 2 (exact scaling by a power of two)
  const float hpen = pen * 0.5f;

  // pairwise column of this lane's target label, kept in registers; the
  // right half runs on the transposed table (source = label of the later node)
  const float *__restrict__ Wuse = rev ? args.WtT : args.Wt;
  float Wr[(POTTS || TRUNC > 0) ? 1 : A];
  if constexpr (!POTTS && TRUNC == 0) {
#pragma unroll
    for (int j = 0; j < A; j++) {
      const int a = s * A + j;
      Wr[j] = (a < k && bvalid) ? Wuse[a * k + b] : INF;
    }
  }
  // Potts table (zero diagonal, one non-negative value w everywhere else; the
  // host checks that before it picks this kernel).  Rounding is monotone, so
  //   min_a (M[a] + W[a][b]) = min(M[b], C),  C = min_a (M[a] + w):
  // the candidate a = b that C also covers is M[b] + w >= M[b] and cannot win.
  // With i1 the lowest a whose M[a] + w equals C, the lowest source attaining
  // the minimum is b when (M[b], b) < (C, i1) lexicographically, else i1 (b
  // itself is never a lower index with candidate C than i1).  One add, one
  // minimum over the chain's lanes and three compares replace the k-candidate
  // scan, and the messages never leave their registers.
  const float wpot = POTTS ? Wuse[1] : 0.0f;
  // lanes past k: their message stays +inf (median with +inf is the maximum)
  const float potts_pad = bvalid ? -INF : INF;
  // Truncated tables (TRUNC = n): W[a][b] = g[|a - b|] with g[d] = g[n + 1] for every d > n
  // and g[d] <= g[n + 1] below (the host checks that before it picks this kernel).  Rounding
  // is monotone, so with C = min_a fl(M[a] + g[n + 1]) over ALL sources
  //   min_a fl(M[a] + W[a][b]) = min( min_{|a - b| <= n} fl(M[a] + g[|a - b|]), C ):
  // a source further away than n contributes exactly its term of C, and a near source's term
  // of C is not below its true candidate, so it cannot win wrongly.  The lowest source
  // attaining the minimum is the lowest near source whose true candidate attains it or, when
  // the minimum equals C, the lower of that and i_far, the lowest source whose term attains
  // C (if i_far lies near b its true candidate attains the minimum as well).  The messages
  // stay in registers: the near sources arrive through DPP row shifts (a chain is at most a
  // row of sixteen lanes), with +inf as the weight where the shift would reach outside the
  // chain's labels.
  constexpr int NWT = TRUNC > 0 ? TRUNC : 1;
  float tg0 = 0.0f, tfar = INF, tgm[NWT], tgp[NWT];
  int trunc_arg = 0;
  if constexpr (TRUNC > 0) {
    tg0 = bvalid ? Wuse[0] : INF;
#pragma unroll
    for (int d = 1; d <= TRUNC; d++) {
      const float g = d < k ? Wuse[d] : INF;  // W[0][d]
      tgm[d - 1] = (bvalid && b - d >= 0) ? g : INF;
      tgp[d - 1] = (bvalid && b + d < k) ? g : INF;
    }
    tfar = (bvalid && TRUNC + 1 < k) ? Wuse[TRUNC + 1] : INF;
  }
  // Back-pointers of the Potts step: two lane masks per step, keep (the
  // minimum for target b is attained by b itself first) and eq (lanes whose
  // M + w equals the chain's C; the lowest one of a chain's field is i1).  Step
  // tt of a 64-step chunk goes to lane tt of four registers, which are stored
  // once per chunk: 16 bytes per step and wavefront instead of one per label.
  int vk0 = 0, vk1 = 0, ve0 = 0, ve1 = 0;
  // PIPEB: the masks of a step are worked out one step late, threaded through the next
  // step's chain of dependent instructions.  A wavefront that has its SIMD to itself issues
  // one instruction of any kind every 4.8 clock ticks and waits 8.5 for a result it needs
  // at once (16 behind a DPP move): the step's own chain - add, four-stage lane minimum, row
  // swap, median, add - leaves two slots behind every link, and the compares, the scalar mask
  // arithmetic and the lane write of the step before fit into exactly those.  The registers
  // of the masks are pairs here (one 64-bit move per mask).
  constexpr bool PIPEB = POTTS && KP >= 8;
  unsigned long long vk64 = 0, ve64 = 0;
  float pend_c = 0.0f, pend_C1 = 0.0f, pend_M = 0.0f;  // step whose masks are still to be written:
  int pend_ls = 0;                                      // c, C1, the messages before it, its lane
  bool pend = false;  // (nothing pending: the next step is lane 0 of a chunk, whose own masks overwrite what this writes)
  auto mask_write = [&](const int ls, const unsigned long long keepm, const unsigned long long eq) {
    asm("s_bfm_b64 exec, 1, %4\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3\n\ts_mov_b64 exec, -1" : "+v"(vk64), "+v"(ve64) : "s"(keepm), "s"(eq), "s"(ls));
  };
  float Mreg = bvalid ? 0.0f : INF;  // A_{-1}: min(0, 0 + w) = 0 makes the first step M_0 = t_0
  // Path compression for the trace-back.  The back-pointer of a Potts step is "keep the
  // label, or go to i1", so the composition of the back-pointers of a block of eight steps,
  //   F(b) = label at the node before the block when its last node has label b,
  // follows the recursion  F <- keep[b] ? F[b] : F[i1]  step by step: one cross-lane fetch
  // of lane i1's value and one select, off the recursion's own dependency chain (the fetch
  // is consumed one step later).  F is kept as the LANE that holds the label (so that the
  // trace-back can use it as a lane select), restarted as the identity every eight steps
  // and recorded, one byte per block, four blocks to a dword.  The trace-back then hops
  // over the records (an eighth of the steps, two scalar instructions each) and fills in the
  // labels inside all blocks at once, one lane per block.
  int org = lane;
  int org_pend = lane;
  unsigned long long keep_pend = ~0ull;
  uint32_t rec = 0;

  if constexpr (RECOMP) {
    // every wavefront of the block writes the whole table (same values), so
    // no block-wide barrier is needed before its own reads
    for (int i = lane; i < KP * KP; i += 64) {
      const int bb = i / KP, a = i % KP;
      s_W[rev][bb * WLD + a] = (a < k && bb < k) ? Wuse[a * k + bb] : INF;
    }
    wave_lds_fence();
  }

  // Addresses: the operand arrays are read and written through buffer
  // descriptors whose base is the lowest address this half chain touches; a
  // step's position is a scalar byte offset and the lane's share a 32-bit
  // offset that never changes, so the sweep spends no vector instruction on
  // addressing.  Operand loads are never predicated: steps past the half
  // chain's end and labels past k read the other half or the slack around every
  // operand array, and their values are never used.
  // A group of NS steps starting at step tg lies at the uniform signed offset
  // tg * sdir + adj from the half's first node, and inside it lane s holds
  // step tg + s at the non-negative offset lane_off (towards lower addresses
  // when the half runs backwards, hence the mirrored s).
  constexpr int kRsrcWord3 = 0x00020000;  // raw buffer of dwords
  const unsigned nstride = (unsigned)(sg.sstride * (long long)k);
  // byte offset of the group that starts at step tg: soff0 + tg * sstep, modulo 2^32
  // (the host checks that a half chain's span, slack included, stays below 4 GiB)
  const unsigned reach = (unsigned)T + 96u;  // steps a backward half may touch, prefetch overrun included
  const unsigned sstep = rev ? 0u - 4u * nstride : 4u * nstride;
  const unsigned soff0 = rev ? 4u * nstride * (reach - (unsigned)(NS - 1)) : 0u;
  const long long base_el = first0 * k - (rev ? (long long)reach * nstride : 0);
  const __amdgpu_buffer_rsrc_t th_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(sg.theta + base_el), 0, 0xFFFFFFFF, kRsrcWord3);
  // (int16 counts: the same element offsets, two bytes each)
  float *const lam_out_or_in = sg.lam_out ? sg.lam_out : const_cast<float *>(sg.lam);
  const __amdgpu_buffer_rsrc_t lam_rs = __builtin_amdgcn_make_buffer_rsrc(
      LAM16 ? (void *)(reinterpret_cast<count_t *>(const_cast<float *>(sg.lam)) + base_el) : (void *)const_cast<float *>(sg.lam + base_el),
      0, 0xFFFFFFFF, kRsrcWord3);
  const __amdgpu_buffer_rsrc_t lamo_rs = __builtin_amdgcn_make_buffer_rsrc(
      LAM16 ? (void *)(reinterpret_cast<count_t *>(lam_out_or_in) + base_el) : (void *)(lam_out_or_in + base_el),
      0, 0xFFFFFFFF, kRsrcWord3);
  const unsigned lam_sign = sg.lam_sign;
  // int16 counts: count * (+-penalty / 2) is the multiplier with the set's sign on it
  const float shpen = lam_sign ? -hpen : hpen;
  const unsigned sstep_l = rev ? 0u - CB * nstride : CB * nstride;
  const unsigned soff0_l = rev ? CB * nstride * (reach - (unsigned)(NS - 1)) : 0u;
  // the lane's own step of group 0, in bytes
  const unsigned lane_off = 4u * ((unsigned)((first - first0) * k) + (unsigned)(rev ? NS - 1 - s : s) * nstride + (unsigned)b);
  // Stores of lanes that own no label (padding past k, the duplicate of a
  // clamped chain) get an offset beyond the descriptor's range, which the
  // hardware drops: no branch around the store.
  const unsigned lane_off_c = LAM8 ? lane_off >> 2 : lane_off >> 1;  // the lane's offset in an array of counts
  const unsigned lane_off_st = (lvalid && sg.lam_out) ? (LAM16 ? lane_off_c : lane_off) : 0xFFFFFFFFu;
  // The updated multipliers are read again by the next launch at the earliest.  The Potts kernels
  // of 32 and 64 labels store them with the non-temporal hint, so that they do not push lines out
  // of the cache that this launch still reads (2048 x 2048, k = 32: 0.345 -> 0.342 ms per sweep;
  // the general kernels: no change).  Not with fewer labels: a step's counts are then a small
  // part of a line, and pieces written around the cache reach memory one by one (256 grids of
  // 256 x 256, k = 8: 0.379 -> 0.548 ms).  (The same hint on the loads of the unaries costs 3 %.)
  constexpr int kStoreNT = (POTTS && KP >= 32) ? 2 : 0;
  auto count_store = [&](const int n, const unsigned step) {
    if constexpr (LAM8) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)n, lamo_rs, lane_off_st, soff0_l + step * sstep_l, kStoreNT);
    else __builtin_amdgcn_raw_buffer_store_b16((short)n, lamo_rs, lane_off_st, soff0_l + step * sstep_l, kStoreNT);
  };
  auto buf_load = [&](const __amdgpu_buffer_rsrc_t &r, unsigned soff) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, soff, 0));
  };
  // a multiplier: binary32, or (LAM16) the int16 count, sign-extended, as the bits of the float.
  // (SPLIT: the count stays in the width it was loaded in until it is used, count_of().
  // Widened here, the compiler moves that instruction to where the two forms of the block
  // meet again, the end of the round, with a wait for every load of the round in front of
  // it: the memory latency of the prefetch, once per round.)
  typedef typename std::conditional<SPLIT && LAM8, unsigned char, typename std::conditional<SPLIT && LAM16, unsigned short, float>::type>::type lamreg_t;
  auto lam_load = [&](unsigned step) -> lamreg_t {
    if constexpr (LAM8) {
      const unsigned char u = __builtin_amdgcn_raw_buffer_load_b8(lam_rs, lane_off_c, soff0_l + step * sstep_l, 0);
      if constexpr (SPLIT) return u;
      else return __int_as_float((int)__builtin_bit_cast(signed char, u));
    } else if constexpr (LAM16) {
      const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(lam_rs, lane_off_c, soff0_l + step * sstep_l, 0);
      if constexpr (SPLIT) return u;
      else return __int_as_float((int)__builtin_bit_cast(short, u));
    } else {
      return buf_load(lam_rs, soff0 + step * sstep);
    }
  };
  auto count_of = [&](const lamreg_t raw) -> int {
    if constexpr (SPLIT && LAM8) return (int)__builtin_bit_cast(signed char, raw);
    else if constexpr (SPLIT && LAM16) return (int)__builtin_bit_cast(short, raw);
    else return __float_as_int(raw);
  };
  // labels of the half in sweep order: xchain[xdir * step]
  label_t *xchain = sg.x + (long long)chain * Tfull + (rev ? Tfull - 1 : 0);
  const label_t *xinchain = DUAL ? nullptr : sg.xin + (long long)chain * Tfull + (rev ? Tfull - 1 : 0);
  // ... and the crossing chains' labels of the same nodes: xochain[xostep * step]
  const long long xostep = (long long)xdir * sg.xo_sstride;
  const label_t *xochain = DUAL ? nullptr
                                : sg.xo + (long long)(chain / sg.cpg) * sg.xo_gstride + (long long)(chain % sg.cpg) * sg.xo_cstride +
                                      (rev ? (long long)(Tfull - 1) * sg.xo_sstride : 0);
  const int Tpad = (T + kChunk - 1) / kChunk * kChunk;
  const int TpadH = (Lhalf + kChunk - 1) / kChunk * kChunk;  // scratch per half chain, in steps
  uint32_t *bpchain = DUAL ? nullptr : reinterpret_cast<uint32_t *>(sg.bp) + ((long long)chain * 2 + rev) * (TpadH / 4) * KP;
  // Potts step: the wavefront's share of the scratch, [half][chunk][4][64] dwords
  uint32_t *bpwave = DUAL ? nullptr : reinterpret_cast<uint32_t *>(sg.bp) + (long long)chain0 * 2 * (TpadH / 4) * KP + (long long)rev * 4 * TpadH;
  // ... and of the block records behind the masks of both halves: [32 steps][64] dwords
  uint32_t *bprec = DUAL ? nullptr : reinterpret_cast<uint32_t *>(sg.bp) + (long long)chain0 * 2 * (TpadH / 4) * KP + 8LL * TpadH + (long long)rev * 2 * TpadH;
  // the same scratch for the stores of the forward pass: descriptor at the
  // wavefront's first chain, the lane's chain and label as a fixed offset
  const __amdgpu_buffer_rsrc_t bp_rs = __builtin_amdgcn_make_buffer_rsrc(
      DUAL ? nullptr : reinterpret_cast<uint32_t *>(sg.bp) + ((long long)chain0 * 2 + rev) * (TpadH / 4) * KP, 0, 0xFFFFFFFF, kRsrcWord3);
  const unsigned bp_lane = lvalid ? 4u * ((unsigned)(chain - chain0) * 2u * (unsigned)(TpadH / 4) * KP + (unsigned)s * KP + (unsigned)b) : 0xFFFFFFFFu;

  float *Mrow = &s_M[wib][cs * KP];
  // where this lane's M_new goes at sub-step u: the real slot when the lane
  // holds that step's operands, a private dump slot otherwise
  int m_addr[NS];
#pragma unroll
  for (int u = 0; u < NS; u++) m_addr[u] = (s == u) ? (cs * KP + b) : (CPW * KP + lane);

  float pth[GPB];
  lamreg_t plam[GPB];
#pragma unroll
  for (int j = 0; j < GPB; j++) {
    const unsigned soff = soff0 + (unsigned)(j * NS) * sstep;
    pth[j] = (!SPLIT || role == 1) ? buf_load(th_rs, soff) : 0.0f;
    plam[j] = (!SPLIT || role == 1) ? lam_load((unsigned)(j * NS)) : lamreg_t(0);
  }
  // labels of the next 64-step chunk, in flight (lane i of a chain: steps i, i + LPC, ...)
  constexpr int NXL = LPC >= kChunk ? 1 : kChunk / LPC;
  constexpr bool XNEXT = NXL <= 4;  // narrower chains fill LDS at the chunk's end, with a wait for the loads
  int xnext[NXL], xonext[NXL];
#pragma unroll
  for (int q = 0; q < NXL; q++) xnext[q] = xonext[q] = -1;
  auto labels_fetch = [&](const int at) {  // at: first step of the chunk to fetch
#pragma unroll
    for (int q = 0; q < NXL; q++) {
      const int i = within + q * LPC;
      const bool ok = i < kChunk && at + i < T;
      xnext[q] = ok ? xinchain[xdir * (at + i)] : -1;
      xonext[q] = ok ? xochain[xostep * (at + i)] : -1;
    }
  };
  auto labels_to_lds = [&]() {
#pragma unroll
    for (int q = 0; q < NXL; q++) {
      const int i = within + q * LPC;
      if (i < kChunk) {
        s_x[wib][cs][i] = xnext[q];
        s_xo[wib][cs][i] = xonext[q];
      }
    }
  };
  if (!DUAL && (!SPLIT || role == 1)) {
    for (int i = within; i < kChunk; i += LPC) {
      s_x[wib][cs][i] = (i < T) ? xinchain[xdir * i] : -1;
      s_xo[wib][cs][i] = (i < T) ? xochain[xostep * i] : -1;
    }
  }

  uint32_t acc = 0, keep = 0;
  double unary_acc = 0.0;

  // Software pipeline: the back-pointer of a step (which source attained the
  // minimum) is not needed to go on, only the minimum is.  So a step's
  // candidates are kept, and their arg-min is worked out one step later, in
  // the shadow of that step's LDS round trip.
  constexpr int NPREV = RECOMP ? NG : A;
  float c_prev[NPREV];  // candidates of the step, or the minima of their groups
  float best_prev = 0.0f, blo_prev = 0.0f, bhi_prev = 0.0f;
  float qlo_prev = 0.0f, qhi_prev = 0.0f;  // four lanes per label: the first (row) level of the combine
#pragma unroll
  for (int q = 0; q < NPREV; q++) c_prev[q] = 0.0f;

  // Arg-min and back-pointer bookkeeping of the step at block position SBP of
  // the block that starts at step tbp.  live = false: the step lies past the
  // chain's end and only the packing goes on, so a partly filled window of
  // four steps is still stored.
  // With the group scheme the arg-min has two stages, one step apart, so that
  // the LDS reads of the first are never waited for: locate() finds the group
  // and issues the reads of its four candidates, finish() compares them.
  typedef float f4_t __attribute__((ext_vector_type(4)));
  f4_t m4_pend = {0.f, 0.f, 0.f, 0.f}, w4_pend = {0.f, 0.f, 0.f, 0.f};
  // 4 * group + the lane's first source + the chain's first lane: the message
  // index of the group's first candidate, and the lane that holds that source
  int grp_pend = 0;
  float best_pend = 0.0f, blo_pend = 0.0f, bhi_pend = 0.0f;
  const int g_lo = cs * LPC + s * A, g_hi = g_lo + 4 * (NG / 2);
  const int g_val[4] = {g_lo, g_lo + 4, g_lo + 8, g_lo + 12};  // four groups per lane: picked directly
  const int w_base = b * WLD - cs * LPC;  // the lane's pairwise row, less what grp_pend carries
  auto locate = [&](auto sbp_c, const bool live) {
    constexpr int SBP = decltype(sbp_c)::value;
    if constexpr (RECOMP) {
      if (live) {
        if constexpr (NG >= 8) grp_pend = first_equal_sc_base<NG >= 8 ? NG : 8, 4>(c_prev, best_prev, g_lo, g_hi);
        else grp_pend = first_equal4_vals(c_prev, best_prev, g_val);
        best_pend = best_prev;
        blo_pend = blo_prev;
        bhi_pend = bhi_prev;
        // the step's messages are still in the row it read them from (one
        // chain per wavefront, or one lane per label: the chain's first lane
        // is also its first message)
        m4_pend = *reinterpret_cast<const f4_t *>(&s_M[wib][(SBP & 1) * MROW + grp_pend]);
        w4_pend = *reinterpret_cast<const f4_t *>(&s_W[rev][w_base + grp_pend]);
      }
    }
  };
  auto finish = [&](auto sbp_c, const int tbp, const bool live) {
    constexpr int SBP = decltype(sbp_c)::value;
    if (DUAL) return;
    int arg = 0;
    if (live) {
      if constexpr (TRUNC > 0) {
        arg = trunc_arg;
      } else if constexpr (RECOMP) {
        const f4_t c4 = m4_pend + w4_pend;
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
        arg = grp_pend | first_equal<4>(cc, best_pend);  // the lane that holds the source label
      } else {
        arg = s * A + first_equal<A, (NS > 1)>(c_prev, best_prev);
      }
      if constexpr (NS == 2 && KP == 32) {
        const auto ra = __builtin_amdgcn_permlane32_swap((unsigned)arg, (unsigned)arg, false, false);
        // ties keep the lower half's (lower) label
        if constexpr (RECOMP) arg = (bhi_pend < blo_pend) ? (int)ra[1] : (int)ra[0];
        else arg = (bhi_prev < blo_prev) ? (int)ra[1] : (int)ra[0];
      } else if constexpr (NS == 4 && KP == 16) {
        // rows s = 0..3 of sixteen lanes hold the four slices of a label's
        // sources: first rows 0|1 and 2|3 (one row swap), then the halves
        const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)arg, (unsigned)arg, false, false);
        arg = (qhi_prev < qlo_prev) ? (int)r1[1] : (int)r1[0];
        const auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)arg, (unsigned)arg, false, false);
        arg = (bhi_prev < blo_prev) ? (int)r2[1] : (int)r2[0];
      } else if constexpr (NS > 1) {
        // partner values were exchanged when the minimum was combined; redo
        // the same comparisons on the (minimum, label) pairs
        float bb = best_prev;
#pragma unroll
        for (int d = KP; d < LPC; d <<= 1) {
          const float ob = __shfl_xor(bb, d);
          const int oa = __shfl_xor(arg, d);
          const bool take = (ob < bb) || (ob == bb && oa < arg);
          arg = take ? oa : arg;
          bb = __builtin_fminf(bb, ob);
        }
      }
    }
    // With one or two chains per wavefront the byte stored is the LANE that
    // holds the source label (label + chain slot * lanes per chain): the
    // backtrack then uses it as a v_readlane selector as it stands.
    if constexpr (SWALK && !RECOMP) arg += cs * LPC;
    acc = __builtin_amdgcn_alignbit((unsigned)arg, acc, 8);
    if ((SBP & 3) == 3) {
      if (s == ((SBP >> 2) % NS)) keep = acc;
      if (((SBP >> 2) % NS) == NS - 1) {
        const int w0 = (tbp + SBP - (WIN - 1)) >> 2;  // first dword of the window
        __builtin_amdgcn_raw_buffer_store_b32(keep, bp_rs, bp_lane, (unsigned)w0 * (4u * KP), 0);
      }
    }
  };

  // Potts step, path compression: the select that finishes the previous step's update
  // (`after`: a value the select must not be scheduled ahead of -- left to itself the
  // compiler hoists it to right behind the fetch and then waits for the fetch there)
  auto origin_settle = [&](const unsigned long long after) {
    asm("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(org) : "v"(org_pend), "s"(keep_pend), "s"(after));
  };
  // ... one step: eq's lowest lane of this lane's chain is i1; fetch that lane's value now,
  // select it (where the label is not kept) when the next step comes by
  const int cbase = cs * LPC, cbase4 = cbase << 2;
  auto origin_step = [&](const unsigned long long eq, const unsigned long long keepm) {
    origin_settle(keepm);
    if constexpr (KP == 64) {
      org_pend = __builtin_amdgcn_readlane(org, __builtin_ctzll(eq));
    } else {
      const unsigned field = (unsigned)(eq >> cbase);  // the chain's field has a bit set
      int src;  // byte address of the lane: (i1 << 2) | (first lane of the chain << 2)
      asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(src) : "v"(__builtin_ctz(field)), "v"(cbase4));
      org_pend = __builtin_amdgcn_ds_bpermute(src, org);
    }
    keep_pend = keepm;
  };
  // ... a block of eight steps starts at step tb: record the block before it, restart
  auto origin_block = [&](const int tb) {
    origin_settle(keep_pend);
    keep_pend = ~0ull;
    if (tb > 0) {
      rec = __builtin_amdgcn_alignbit((unsigned)org, rec, 8);
      const int r = (tb >> 3) - 1;
      if ((r & 3) == 3) bprec[(r >> 2) * 64 + lane] = rec;
    }
    org = lane;
  };

  // the masks of the step still pending (PIPEB), outside the next step's instructions
  auto flush_pending = [&]() {
    if (!pend) return;
    const unsigned long long eq = __builtin_amdgcn_ballot_w64(pend_c == pend_C1);
    const unsigned long long lt = __builtin_amdgcn_ballot_w64(pend_M < pend_C1);
    const unsigned long long le = __builtin_amdgcn_ballot_w64(pend_M <= pend_C1);
    const unsigned long long below = (eq - kFieldLsb<KP>) & ~eq;
    mask_write(pend_ls, lt | (le & below), eq);
    if constexpr (PCOMP) origin_step(eq, lt | (le & below));
    pend = false;
    pend_ls = 0;
  };

  // One unrolled block of BLK steps starting at step tb (tt0 = tb - chunk start).
  // The order inside a step is pinned with scheduling barriers: LDS reads
  // first, then the previous step's arg-min (it covers the LDS latency), then
  // the min-plus of this step.
  // role_tag: 0 the whole step; SPLIT: 1 the shell's part (slot: the ring's block it fills),
  // 2 the recursion's part (thr: the block's adjusted unaries, fetched from the ring a block ago)
  float thr[SPLIT ? GPB : 1];
  constexpr bool XPRE = TRUNC > 0 && NS == 1 && GPB % 4 == 0 && !DUAL && !SPLIT;
  int xq_own[XPRE ? GPB : 1], xq_oth[XPRE ? GPB : 1], xn_own[XPRE ? GPB : 1], xn_oth[XPRE ? GPB : 1];
  auto load_block_labels = [&](int *own, int *oth, const int at) {
#pragma unroll
    for (int q = 0; q < GPB / 4; q++) {
      const int4 a = *reinterpret_cast<const int4 *>(&s_x[wib][cs][at + 4 * q]);
      const int4 o = *reinterpret_cast<const int4 *>(&s_xo[wib][cs][at + 4 * q]);
      own[4 * q + 0] = a.x; own[4 * q + 1] = a.y; own[4 * q + 2] = a.z; own[4 * q + 3] = a.w;
      oth[4 * q + 0] = o.x; oth[4 * q + 1] = o.y; oth[4 * q + 2] = o.z; oth[4 * q + 3] = o.w;
    }
  };
  auto block = [&](const int tb, const int tt0, auto full_tag, auto role_tag, const int slot) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr bool SHELL = ROLE != 2, REC = ROLE != 1;
    if constexpr (PCOMP && REC) {
      static_assert(!PCOMP || BLK == 8, "the trace-back's records are one per unrolled block of eight steps");
      // (PIPEB, whole block: the last step of the block before is still pending; the block's
      // first step settles it and starts the new record then)
      if constexpr (PIPEB && !FULL) flush_pending();
      if constexpr (!(PIPEB && FULL)) origin_block(tb);
    }
    int xown = -1, xoth = -1;  // labels of the current group (own step of this lane)
    if constexpr (XPRE) {
      // the block's labels are in registers a block ahead of their use: a lone wavefront
      // has nothing else to put in front of an LDS read's latency
      if (tt0 == 0) {
        load_block_labels(xq_own, xq_oth, 0);
      } else {
#pragma unroll
        for (int q = 0; q < GPB; q++) { xq_own[q] = xn_own[q]; xq_oth[q] = xn_oth[q]; }
      }
    } else if (!DUAL && SHELL) {
      xown = s_x[wib][cs][tt0 + s];
      xoth = s_xo[wib][cs][tt0 + s];
    }
    // integer counts: the shell of two steps at a time (PAIRED), whatever the step
    constexpr bool PAIRED = LAM16 && GPB % 2 == 0;
    int xown1 = -1, xoth1 = -1;  // labels of the step after the current one
    float th_pair = INF;         // adjusted unary of the odd step, worked out with the even one
    if constexpr (PAIRED && SHELL && !XPRE) {
      xown1 = s_x[wib][cs][tt0 + 1];
      xoth1 = s_xo[wib][cs][tt0 + 1];
    }
    static_for<0, GPB>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int tg = tb + j * NS;
      const bool glive = FULL || tg < T;
      const float th_in = pth[SHELL ? j : 0];
      const int n_in = LAM16 ? count_of(plam[SHELL ? j : 0]) : 0;  // LAM16: the count
      float lam_in;
      if constexpr (LAM16) lam_in = (float)n_in * shpen;
      else lam_in = __uint_as_float(__float_as_uint(plam[SHELL ? j : 0]) ^ lam_sign);
      const unsigned goff = soff0 + (unsigned)tg * sstep;  // uniform
      if constexpr (XPRE) {
        xown = xq_own[j];
        xoth = xq_oth[j];
        if constexpr (PAIRED && j % 2 == 0) {
          xown1 = xq_own[j + 1];
          xoth1 = xq_oth[j + 1];
        }
        if (j == GPB / 2 && tt0 + BLK < kChunk) load_block_labels(xn_own, xn_oth, tt0 + BLK);
      }
      float th_own = INF;
      if constexpr (!SHELL) {
        th_own = thr[SPLIT ? j : 0];
      } else if constexpr (PAIRED) {
        // The shell is elementwise, so steps j and j + 1 go through it together on
        // packed binary32 instructions.  With d = [x_own = b] - [x_other = b] and
        // m = +-n + d (the set's sign on the count):
        //   lambda + penalty (mu - m) = m * penalty/2,  penalty (mu - m) = d * penalty/2,
        // both exact, hence the same bits as the sums of Tree.cpp:70 and :85-86.
        // A step past the chain's end has labels -1 (d = 0, nothing added to the
        // unary sum), is never stored and its value never used.
        if constexpr (j % 2 == 0) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          const int sgn = lam_sign ? -1 : 1;
          const int d0 = (int)(xown == b) - (int)(xoth == b), d1 = (int)(xown1 == b) - (int)(xoth1 == b);
          // m = n * sgn + d in one instruction (the counts are 16-bit; left to itself the
          // compiler takes a 64-bit multiply-add for it)
          int m0, m1;
          asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(m0) : "v"(n_in), "s"(sgn), "v"(d0));
          asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(m1) : "v"(count_of(plam[j + 1])), "s"(sgn), "v"(d1));
          const f2 hp2 = {hpen, hpen}, rho2 = {rho, rho};
          const f2 la2 = f2{(float)m0, (float)m1} * hp2;   // Tree.cpp:70
          const f2 p2 = f2{(float)d0, (float)d1} * hp2;
          const f2 th2 = {th_in, pth[j + 1]};
          const f2 own2 = rho2 * (th2 + la2) + p2;          // Tree.cpp:85-86
          th_own = own2.x;
          th_pair = own2.y;
          float sel0 = (xown == b) ? th_in : 0.0f, sel1 = (xown1 == b) ? pth[j + 1] : 0.0f;
          asm("" : "+v"(sel0), "+v"(sel1));
          unary_acc += (double)sel0;
          unary_acc += (double)sel1;
          // (stored by the sets that keep their own array, where sgn = 1 and m is the new count)
          if (FULL || tg < T) count_store(m0, (unsigned)tg);
          if (FULL || tg + 1 < T) count_store(m1, (unsigned)(tg + 1));
        } else {
          th_own = th_pair;
        }
      } else if (DUAL) {
        th_own = th_in + lam_in;
      } else if (glive) {
        // p = penalty * (mu - m) = -(penalty * (m - mu)), exactly
        const float p = ((xown == b) ? hpen : 0.0f) - ((xoth == b) ? hpen : 0.0f);
        // Unary energy of the labels this sweep starts from (one of the two
        // terms of Grid.cpp:1665-1680 for the previous solve): the operand is
        // in a register here, which saves the backtrack a gather per node.
        // selected in binary32 (one select instead of the two a binary64 select
        // costs); the empty asm keeps the compiler from moving the select behind
        // the conversion
        float th_sel = (xown == b) ? th_in : 0.0f;
        asm("" : "+v"(th_sel));
        unary_acc += (double)th_sel;
        const float la = lam_in + p;                          // Tree.cpp:70
        if constexpr (LAM16) {
          const int n_out = n_in + (int)(xown == b) - (int)(xoth == b);
          if (FULL || tg + s < T) count_store(n_out, (unsigned)tg);
        } else {
          if (FULL || tg + s < T) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(la), lamo_rs, lane_off_st, goff, kStoreNT);
        }
        th_own = rho * (th_in + la) + p;                      // Tree.cpp:85-86
      }
      if constexpr (ROLE == 1) s_ring[rev][slot * GPB + j][lane] = th_own;
      static_for<0, NS>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value;
        const bool live = FULL || tg + u < T;
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        // (a) this step's messages leave LDS, with the next group's labels
        //     and the operands of the group one block ahead behind them ...
        f4 mv[(A + 3) / 4];
        // MVLATE: four wavefronts per SIMD leave 128 registers; the second half of the step's
        // messages then leaves LDS only after the arg-min work of (b), whose registers it reuses
        // (the other wavefronts of the SIMD cover its latency)
        // (64 source labels per lane at three wavefronts per SIMD, 168 registers: four pieces, each
        // read while the one before it is worked on)
        constexpr int MVP = !RECOMP ? 1 : (A == 32 && MINW >= 4) ? 2 : (A == 64 && MINW >= 3) ? 4 : 1;  // pieces
        constexpr bool MVLATE = MVP > 1;
        if (live && !POTTS && REC && TRUNC == 0) {
          if constexpr (A % 4 == 0) {
            const f4 *mp = reinterpret_cast<const f4 *>(Mrow + (RECOMP ? ((j * NS + u) & 1) * MROW : 0) + s * A);
#pragma unroll
            for (int q = 0; q < A / 4 / MVP; q++) mv[q] = mp[q];
          } else {
#pragma unroll
            for (int q = 0; q < A; q++) mv[q / 4][q % 4] = Mrow[s * A + q];
          }
        }
        if (u == NS - 1 && SHELL) {
          if constexpr (PAIRED) {
            if (!XPRE && j % 2 == 1 && j + 1 < GPB) {
              xown = s_x[wib][cs][tt0 + j + 1];
              xoth = s_xo[wib][cs][tt0 + j + 1];
              xown1 = s_x[wib][cs][tt0 + j + 2];
              xoth1 = s_xo[wib][cs][tt0 + j + 2];
            }
          } else if (!DUAL && !XPRE && j + 1 < GPB) {
            xown = s_x[wib][cs][tt0 + (j + 1) * NS + s];
            xoth = s_xo[wib][cs][tt0 + (j + 1) * NS + s];
          }
          // The operands of the next block: one load per step, or (PFB) all of them behind the
          // block's last step.  Back to back, the loads of consecutive steps that share a
          // 128-byte line (8 labels: four steps of a row chain, 16 labels: two; the int8 counts
          // of 32 labels: four) are served by one fetch of it, and consecutive lines reach the
          // memory system together.  The first steps of the next block then wait for their
          // operands unless other wavefronts of the SIMD have work, so it pays where the SIMDs
          // are full - 256 grids of 256 x 256, k = 8: 0.433 -> 0.378 ms per sweep (with blocks of
          // 16 steps); 2048 x 2048, k = 32: 0.348 -> 0.345 ms - and loses where they are not
          // (two strips of that grid, two wavefronts per SIMD: 0.231 -> 0.245 ms).  Up to 16
          // labels the kernel always fetches in bursts; with 32 and 64 the instance with path
          // compression does, which the host launches on a full chip only.  The general
          // kernel keeps a load per step (32 labels, full chip: + 1 %; 64 labels: - 3 %; 16: no
          // change).
          // A shell wavefront (SPLIT) that has its SIMD to itself gains by the burst as well
          // (one strip of eight of the 2048 x 2048 grid: 0.115 -> 0.110 ms per sweep); sharing
          // the SIMD with another chain's it loses (one strip of four: 0.140 -> 0.146, and as
          // much with the shell's issue priority raised).  The
          // host launches the instance built for two wavefronts per SIMD (MINW = 2) in the
          // first case, and that is the one that fetches in bursts.
                    constexpr int PFB = (POTTS && !DUAL && (SPLIT ? MINW == 2 : (KP <= 16 || PCOMP))) ? GPB : 1;
          if constexpr (PFB == 1) {
            const unsigned poff = goff + (unsigned)BLK * sstep;
            pth[j] = buf_load(th_rs, poff);
            plam[j] = lam_load((unsigned)(tg + BLK));
          } else if constexpr (j % PFB == PFB - 1) {
#pragma unroll
            for (int jj = j - PFB + 1; jj <= j; jj++) {
              const int tgj = tb + jj * NS;
              pth[jj] = buf_load(th_rs, soff0 + (unsigned)(tgj + BLK) * sstep);
            }
#pragma unroll
            for (int jj = j - PFB + 1; jj <= j; jj++) plam[jj] = lam_load((unsigned)(tb + jj * NS + BLK));
          }
        }
        // memory instructions stay where they are written; with one wavefront
        // per SIMD (a split chain) letting ALU work cross pays, with two it does not
        if constexpr (POTTS) __builtin_amdgcn_sched_barrier(0);  // every step issues its own prefetch
        else if constexpr (NS > 1) __builtin_amdgcn_sched_barrier(0x6);
        else __builtin_amdgcn_sched_barrier(0);
        // (b) ... while the previous step's back-pointer is worked out
        if constexpr (!REC) {
        } else if constexpr (RECOMP) {
          // compare for the step two back, then locate for the step one back
          constexpr int P = j * NS + u;
          constexpr int Q2 = (P + BLK - 2) % BLK, Q1 = (P + BLK - 1) % BLK;
          if (P >= 2) finish(std::integral_constant<int, Q2>{}, tb, FULL || tb + Q2 < T);
          else if (tb > 0) finish(std::integral_constant<int, Q2>{}, tb - BLK, true);
          if (P >= 1) locate(std::integral_constant<int, Q1>{}, FULL || tb + Q1 < T);
          else if (tb > 0) locate(std::integral_constant<int, Q1>{}, true);
          __builtin_amdgcn_sched_barrier(0x6);
        } else if constexpr (PIPE) {
          if (j == 0 && u == 0) {
            if (tb > 0) finish(std::integral_constant<int, BLK - 1>{}, tb - BLK, true);
          } else {
            finish(std::integral_constant<int, (j * NS + u + BLK - 1) % BLK>{}, tb, FULL || tg + u - 1 < T);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // (c) min-plus of this step
        if constexpr (!REC) {
        } else if constexpr (POTTS) {
          if (live) {
            if constexpr (PIPEB && FULL) {
              const int ls = tt0 + j;
              float c, C1, r, t, Mnew;
              unsigned long long m_lt, m_le, m_eq;
              // c = M + w and the first two stages of its minimum over the chain's lanes, with
              // the three compares of the step before, the copy of eq and eq - (lowest bit of
              // every field) between them
#define ADMM_PB_HEAD                                                                                  \
  "v_add_f32_e32 %[c], %[M], %[wp]\n\t"                                                               \
  "v_cmp_eq_f32_e32 vcc, %[pc], %[pC]\n\t"                                                            \
  "v_cmp_lt_f32_e64 %[LT], %[pM], %[pC]\n\t"                                                          \
  "v_min_f32_dpp %[r], %[c], %[c] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                 \
  "v_cmp_le_f32_e64 %[LE], %[pM], %[pC]\n\t"                                                          \
  "s_mov_b64 %[E], vcc\n\t"                                                                           \
  "v_min_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                 \
  "s_sub_u32 vcc_lo, vcc_lo, %[l0]\n\t"                                                               \
  "s_subb_u32 vcc_hi, vcc_hi, %[l1]\n\t"
#define ADMM_PB_DPP(dst, how) "v_min_f32_dpp %[" dst "], %[r], %[r] " how " row_mask:0xf bank_mask:0xf\n\t"
              // below = (eq - lsb) & ~eq, keep = lt | (le & below) ...
#define ADMM_PB_MASKS "s_andn2_b64 vcc, vcc, %[E]\n\ts_and_b64 vcc, vcc, %[LE]\n\t"
#define ADMM_PB_KEEP "s_or_b64 %[LE], vcc, %[LT]\n\t"
              // ... and both masks into lane ls of their register pairs
#define ADMM_PB_WRITE "s_bfm_b64 exec, 1, %[ls]\n\tv_mov_b64 %[vk], %[LE]\n\tv_mov_b64 %[ve], %[E]\n\ts_mov_b64 exec, -1\n\t"
#define ADMM_PB_TAIL "v_med3_f32 %[t], %[M], %[C], %[pad]\n\tv_add_f32_e32 %[Mo], %[th], %[t]"
#define ADMM_PB_OPERANDS                                                                                                                          \
  : [c] "=&v"(c), [r] "=&v"(r), [t] "=&v"(t), [C] "=&v"(C1), [Mo] "=&v"(Mnew), [LT] "=&s"(m_lt), [LE] "=&s"(m_le), [E] "=&s"(m_eq),               \
    [vk] "+v"(vk64), [ve] "+v"(ve64)                                                                                                              \
  : [M] "v"(Mreg), [wp] "v"(wpot), [pc] "v"(pend_c), [pC] "v"(pend_C1), [pM] "v"(pend_M), [th] "v"(th_own), [pad] "v"(potts_pad),                  \
    [ls] "s"(pend_ls), [l0] "n"((unsigned)kFieldLsb<KP>), [l1] "n"((unsigned)(kFieldLsb<KP> >> 32))                                               \
  : "vcc", "scc"
              if constexpr (KP == 8) {
                asm(ADMM_PB_HEAD ADMM_PB_DPP("C", "row_half_mirror") ADMM_PB_MASKS ADMM_PB_KEEP ADMM_PB_WRITE ADMM_PB_TAIL ADMM_PB_OPERANDS);
              } else if constexpr (KP == 16) {
                asm(ADMM_PB_HEAD ADMM_PB_DPP("r", "row_half_mirror") ADMM_PB_MASKS ADMM_PB_DPP("C", "row_mirror") ADMM_PB_KEEP ADMM_PB_WRITE ADMM_PB_TAIL
                    ADMM_PB_OPERANDS);
              } else if constexpr (KP == 32) {
                // (the lane write sits between the copy and the row swap that reads it: its two wait states)
                asm(ADMM_PB_HEAD ADMM_PB_DPP("r", "row_half_mirror") ADMM_PB_MASKS ADMM_PB_DPP("r", "row_mirror") ADMM_PB_KEEP
                    "v_mov_b32 %[t], %[r]\n\t" ADMM_PB_WRITE "v_permlane16_swap_b32 %[r], %[t]\n\tv_min_f32_e32 %[C], %[r], %[t]\n\t" ADMM_PB_TAIL
                    ADMM_PB_OPERANDS);
              } else {
                static_assert(KP == 64, "lanes per chain");
                asm(ADMM_PB_HEAD ADMM_PB_DPP("r", "row_half_mirror") ADMM_PB_MASKS ADMM_PB_DPP("r", "row_mirror") ADMM_PB_KEEP
                    "v_mov_b32 %[t], %[r]\n\t" ADMM_PB_WRITE "v_permlane16_swap_b32 %[r], %[t]\n\tv_min_f32_e32 %[r], %[r], %[t]\n\t"
                    "v_mov_b32 %[t], %[r]\n\ts_nop 1\n\tv_permlane32_swap_b32 %[r], %[t]\n\tv_min_f32_e32 %[C], %[r], %[t]\n\t" ADMM_PB_TAIL
                    ADMM_PB_OPERANDS);
              }
#undef ADMM_PB_HEAD
#undef ADMM_PB_DPP
#undef ADMM_PB_MASKS
#undef ADMM_PB_KEEP
#undef ADMM_PB_WRITE
#undef ADMM_PB_TAIL
#undef ADMM_PB_OPERANDS
              if constexpr (PCOMP) {
                if (pend) origin_step(m_eq, m_le);  // of the step before, whose masks these are
                if (j == 0) origin_block(tb);
              }
              pend_c = c;
              pend_C1 = C1;
              pend_M = Mreg;
              pend_ls = ls;
              pend = true;
              Mreg = Mnew;
            } else {
            if constexpr (PIPEB) flush_pending();
            const float c = Mreg + wpot;
            const float C1 = chain_lanes_min<KP>(c);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(c == C1);
            const unsigned long long lt = __builtin_amdgcn_ballot_w64(Mreg < C1);
            const unsigned long long le = __builtin_amdgcn_ballot_w64(Mreg <= C1);
            // lanes below the chain's lowest lane with c == C1 (every chain's
            // field of eq has a bit set, so the borrow stays inside the field)
            const unsigned long long below = (eq - kFieldLsb<KP>) & ~eq;
            const unsigned long long keepm = lt | (le & below);
            const int ls = tt0 + j;  // the step's place in its chunk (uniform)
            if constexpr (PIPEB) {
              mask_write(ls, keepm, eq);
            } else {
            // into lane ls of the four registers: moves with only that lane enabled (a
            // lane write takes one scalar operand, and the lane would be a second)
            // (s_bfm: one set bit at the offset, and unlike a shift it leaves the
            // scalar condition code alone)
            asm("s_bfm_b64 exec, 1, %8\n\t"
                "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t"
                "s_mov_b64 exec, -1"
                : "+v"(vk0), "+v"(vk1), "+v"(ve0), "+v"(ve1)
                : "s"((unsigned)keepm), "s"((unsigned)(keepm >> 32)), "s"((unsigned)eq), "s"((unsigned)(eq >> 32)), "s"(ls));
            }
            Mreg = th_own + __builtin_amdgcn_fmed3f(Mreg, C1, potts_pad);
            if constexpr (PCOMP) origin_step(eq, keepm);
            }
          }
        } else if constexpr (TRUNC > 0) {
          if (live) {
            // candidates of the near sources b - n .. b + n (DPP: lane b reads lane b -+ d of its
            // row, 0 where the row ends, which the +inf weight covers) and of the far ones
            static_assert(TRUNC == 0 || (TRUNC == 3 && KP == 16), "written out for three neighbours on either side and chains of one DPP row");
            // A wavefront alone on its SIMD issues one instruction of any kind per turn, a wait
            // state included, so the step is written out with the cross-lane minimum of the far
            // term threaded through the near candidates: every DPP read then has its two
            // instructions of distance from the write it depends on and no wait state is needed
            // (the messages were written by the add that precedes this statement; the first two
            // instructions read them without DPP).
            float cm0, cm1, cm2, cp0, cp1, cp2, c0, cf, C, best, tmin, nlo, nhi;
            asm("v_add_f32_e32 %[cf], %[M], %[wfar]\n\t"
                "v_add_f32_e32 %[c0], %[M], %[w0]\n\t"
                "v_add_f32_dpp %[cm0], %[M], %[wm0] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_add_f32_dpp %[cm1], %[M], %[wm1] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_min_f32_dpp %[t], %[cf], %[cf] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[cm2], %[M], %[wm2] row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_add_f32_dpp %[cp0], %[M], %[wp0] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_min_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[cp1], %[M], %[wp1] row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_add_f32_dpp %[cp2], %[M], %[wp2] row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                "v_min_f32_dpp %[t], %[t], %[t] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_min3_f32 %[nlo], %[c0], %[cm0], %[cm1]\n\t"
                "v_min3_f32 %[nhi], %[cm2], %[cp0], %[cp1]\n\t"
                "v_min_f32_dpp %[C], %[t], %[t] row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_min3_f32 %[best], %[nlo], %[nhi], %[cp2]\n\t"
                "v_min_f32_e32 %[best], %[best], %[C]"
                : [cf] "=&v"(cf), [c0] "=&v"(c0), [cm0] "=&v"(cm0), [cm1] "=&v"(cm1), [cm2] "=&v"(cm2), [cp0] "=&v"(cp0), [cp1] "=&v"(cp1),
                  [cp2] "=&v"(cp2), [t] "=&v"(tmin), [nlo] "=&v"(nlo), [nhi] "=&v"(nhi), [C] "=&v"(C), [best] "=&v"(best)
                : [M] "v"(Mreg), [wfar] "v"(tfar), [w0] "v"(tg0), [wm0] "v"(tgm[0]), [wm1] "v"(tgm[NWT > 1 ? 1 : 0]), [wm2] "v"(tgm[NWT > 2 ? 2 : 0]),
                  [wp0] "v"(tgp[0]), [wp1] "v"(tgp[NWT > 1 ? 1 : 0]), [wp2] "v"(tgp[NWT > 2 ? 2 : 0]));
            // lowest near source attaining the minimum, as an offset from b (64: none); a
            // select reads the mask of the compare three instructions back, which covers the
            // two wait states between a vector write of a scalar register and its use.  eqf:
            // the lanes whose far term attains C; eqc: the lanes whose minimum is the far one.
            int off;
            unsigned long long eqf, eqc, ma, mb;
            // (Every third compare goes into vcc: a compare into any other scalar pair costs a lone
            // wavefront 9.9 clock ticks, one into vcc 4.8.)
            asm("v_cmp_eq_f32_e32 vcc, %[cp2], %[best]\n\t"
                "v_cmp_eq_f32_e64 %[ma], %[cp1], %[best]\n\t"
                "v_cmp_eq_f32_e64 %[mb], %[cp0], %[best]\n\t"
                "v_cndmask_b32_e32 %[off], 64, %[three], vcc\n\t"
                "v_cmp_eq_f32_e32 vcc, %[c0], %[best]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], 2, %[ma]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], 1, %[mb]\n\t"
                "v_cmp_eq_f32_e64 %[ma], %[cm0], %[best]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], 0, vcc\n\t"
                "v_cmp_eq_f32_e32 vcc, %[cm1], %[best]\n\t"
                "v_cmp_eq_f32_e64 %[mb], %[cm2], %[best]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], -1, %[ma]\n\t"
                "v_cmp_eq_f32_e64 %[eqf], %[cf], %[C]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], -2, vcc\n\t"
                "v_cmp_eq_f32_e64 %[eqc], %[best], %[C]\n\t"
                "v_cndmask_b32_e64 %[off], %[off], -3, %[mb]"
                : [off] "=&v"(off), [ma] "=&s"(ma), [mb] "=&s"(mb), [eqf] "=&s"(eqf), [eqc] "=&s"(eqc)
                : [cm0] "v"(cm0), [cm1] "v"(cm1), [cm2] "v"(cm2), [c0] "v"(c0), [cp0] "v"(cp0), [cp1] "v"(cp1), [cp2] "v"(cp2), [cf] "v"(cf),
                  [C] "v"(C), [best] "v"(best), [three] "v"(3)
                : "vcc");
            // i_far: the lowest lane of this lane's chain whose far term attains C
            const int ifar = __builtin_ctz((unsigned)(eqf >> (cs * LPC)) | (1u << (KP - 1)));
            const int nearest = b + off;  // >= k when no near source attains the minimum (then it equals C)
            const int lowest = min(nearest, ifar);
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(trunc_arg) : "v"(nearest), "v"(lowest), "s"(eqc));
            // M_0 = t_0; a lane past k starts at +inf and stays there (its weights are +inf)
            if (j == 0 && u == 0) best = (tb == 0) ? (bvalid ? 0.0f : INF) : best;
            Mreg = th_own + best;
          }
        } else if (MVLATE && live) {
          // piece p + 1 leaves LDS, piece p is added to the lane's pairwise column and reduced to
          // the minima of its groups of four
          const f4 *mp = reinterpret_cast<const f4 *>(Mrow + ((j * NS + u) & 1) * MROW + s * A);
          constexpr int PF = A / 4 / MVP;  // groups per piece
          static_for<0, MVP>([&](auto p_c) {
            constexpr int pi = decltype(p_c)::value;
            if constexpr (pi + 1 < MVP) {
#pragma unroll
              for (int q = (pi + 1) * PF; q < (pi + 2) * PF; q++) mv[q] = mp[q];
              if constexpr (MVP > 2) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = pi * PF; q < (pi + 1) * PF; q++) {
              const f2 lo = mv[q].xy + f2{Wr[4 * q + 0], Wr[4 * q + 1]};
              const f2 hi = mv[q].zw + f2{Wr[4 * q + 2], Wr[4 * q + 3]};
              const float c4[4] = {lo.x, lo.y, hi.x, hi.y};
              c_prev[q] = min_of<4>(c4);
            }
            if constexpr (MVP > 2 && pi + 1 < MVP) __builtin_amdgcn_sched_barrier(0);
          });
          float best = __builtin_fminf(min_of<NG / 2>(c_prev), min_of<NG / 2>(c_prev + NG / 2));
          best_prev = best;
          if (j == 0 && u == 0) best = (tb == 0) ? (bvalid ? 0.0f : INF) : best;
          const float Mnew = th_own + best;
          wave_lds_fence();
          s_M[wib][((((j * NS + u) & 1) ^ 1) * MROW) + m_addr[u]] = Mnew;
          wave_lds_fence();
        } else if (live) {
          float c[A];
          if constexpr (A % 4 == 0) {
#pragma unroll
            for (int q = 0; q < A / 4; q++) {
              const f2 lo = mv[q].xy + f2{Wr[4 * q + 0], Wr[4 * q + 1]};
              const f2 hi = mv[q].zw + f2{Wr[4 * q + 2], Wr[4 * q + 3]};
              c[4 * q + 0] = lo.x; c[4 * q + 1] = lo.y;
              c[4 * q + 2] = hi.x; c[4 * q + 3] = hi.y;
            }
          } else {
#pragma unroll
            for (int q = 0; q < A; q++) c[q] = mv[q / 4][q % 4] + Wr[q];
          }
          float best;
          if constexpr (RECOMP) {
#pragma unroll
            for (int q = 0; q < NG; q++) c_prev[q] = min_of<4>(c + 4 * q);
            best = __builtin_fminf(min_of<NG / 2>(c_prev), min_of<NG / 2>(c_prev + NG / 2));
          } else {
            if constexpr (A >= 8) best = __builtin_fminf(min_of<A / 2>(c), min_of<A / 2>(c + A / 2));
            else best = min_of<A>(c);
#pragma unroll
            for (int q = 0; q < A; q++) c_prev[q] = c[q];
          }
          best_prev = best;
          if constexpr (NS == 2 && KP == 32) {
            // lanes l and l^32: one swap gives both halves' minima to every lane
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
            blo_prev = __uint_as_float(rb[0]);
            bhi_prev = __uint_as_float(rb[1]);
            // plain v_min_f32: through the builtin the compiler first canonicalises
            // both operands (it cannot see that the swapped bits are ordinary floats)
            asm("v_min_f32_e32 %0, %1, %2" : "=v"(best) : "v"(blo_prev), "v"(bhi_prev));
          } else if constexpr (NS == 4 && KP == 16) {
            // two register swaps instead of two shuffles through LDS
            const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
            qlo_prev = __uint_as_float(q[0]);
            qhi_prev = __uint_as_float(q[1]);
            asm("v_min_f32_e32 %0, %1, %2" : "=v"(best) : "v"(qlo_prev), "v"(qhi_prev));
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
            blo_prev = __uint_as_float(rb[0]);
            bhi_prev = __uint_as_float(rb[1]);
            asm("v_min_f32_e32 %0, %1, %2" : "=v"(best) : "v"(blo_prev), "v"(bhi_prev));
          } else {
#pragma unroll
            for (int d = KP; d < LPC; d <<= 1) best = __builtin_fminf(best, __shfl_xor(best, d));
          }
          // M_0 = t_0; a lane past k starts at +inf and stays there (its pairwise column is +inf)
          if (j == 0 && u == 0) best = (tb == 0) ? (bvalid ? 0.0f : INF) : best;
          const float Mnew = th_own + best;
          wave_lds_fence();
          s_M[wib][(RECOMP ? (((j * NS + u) & 1) ^ 1) * MROW : 0) + m_addr[u]] = Mnew;
          wave_lds_fence();
        }
        if constexpr (!PIPE && !POTTS && REC) finish(std::integral_constant<int, j * NS + u>{}, tb, live);
      });
    });
  };

  // The wavefronts of a SIMD do not advance at the same pace (the workgroups
  // handed out first finish a quarter earlier than the last ones, and the end
  // of the launch runs on a quarter of the wavefronts).  Issue priorities that
  // rotate every 64 steps, offset by the workgroup's place in the launch, even
  // that out a little: 6 to 7 % with four wavefronts per SIMD (2048 x 2048, k = 32, Potts
  // and dense table), nothing either way with one or two, and with the three wavefronts
  // per SIMD of the 64-label general kernel it costs 2.6 % (1024 x 1024: 1698 against
  // 1742 it/s), so that kernel leaves the priorities alone.  A period of 32 or of 128 steps
  // is no better than 64.  (s_setprio takes an immediate, hence the switch.)
  const int prio_slot = (int)((long long)blockIdx.x * 4 / gridDim.x);
  auto set_prio = [&](int p) {
    switch (p & 3) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  };
  if constexpr (!SPLIT) {
  // One chunk of 64 steps.  Whole chunks run in a loop of their own and the chain's last,
  // partial one after it: in one loop the two forms of the body meet again at its end with
  // the prefetched operands in different registers, and the copies that reconcile them wait
  // for every load in flight - the memory latency once per chunk, which a wavefront that is
  // alone on its SIMD has nothing to cover with.
  auto chunk = [&](const int t0, auto whole_tag) {
    constexpr bool WHOLE = decltype(whole_tag)::value;
    if constexpr (!(RECOMP && A == 64)) set_prio(prio_slot + t0 / kChunk);
    if (!DUAL) {
      // labels of chunk t0 are in LDS; fetch the next chunk's while this one runs
      if constexpr (XNEXT) labels_fetch(t0 + kChunk);
    }
    if constexpr (WHOLE) {
#pragma unroll 2
      for (int tt = 0; tt < kChunk; tt += BLK) block(t0 + tt, tt, std::true_type{}, std::integral_constant<int, 0>{}, 0);
    } else {
#pragma unroll 1
      for (int tt = 0; tt < kChunk && t0 + tt < T; tt += BLK) block(t0 + tt, tt, std::false_type{}, std::integral_constant<int, 0>{}, 0);
    }
    if constexpr (POTTS) {
      uint32_t *o = bpwave + (t0 / kChunk) * 256 + lane;
      if constexpr (PIPEB) {
        flush_pending();
        o[0] = (uint32_t)vk64; o[64] = (uint32_t)(vk64 >> 32); o[128] = (uint32_t)ve64; o[192] = (uint32_t)(ve64 >> 32);
      } else {
        o[0] = (uint32_t)vk0; o[64] = (uint32_t)vk1; o[128] = (uint32_t)ve0; o[192] = (uint32_t)ve1;
      }
    }
    if (!DUAL && WHOLE && t0 + kChunk < T) {
      wave_lds_fence();
      if constexpr (XNEXT) {
        labels_to_lds();
      } else {
        for (int i = within; i < kChunk; i += LPC) {
          const bool ok = t0 + kChunk + i < T;
          s_x[wib][cs][i] = ok ? xinchain[xdir * (t0 + kChunk + i)] : -1;
          s_xo[wib][cs][i] = ok ? xochain[xostep * (t0 + kChunk + i)] : -1;
        }
      }
      wave_lds_fence();
    }
  };
  int t0 = 0;
  for (; t0 + kChunk <= T; t0 += kChunk) chunk(t0, std::true_type{});
  if (t0 < T) chunk(t0, std::false_type{});

  } else {
    // Two wavefronts per half chain.  In round ib the shell works on block ib and the
    // recursion on block ib - 2 with the adjusted unaries it fetched from the ring in
    // round ib - 1; a workgroup barrier ends every round.  All four wavefronts of the
    // workgroup run the same number of rounds (the longer half's).
    const int nblk = (Lhalf + BLK - 1) / BLK;
    // Where two chains' wavefronts share a SIMD the shell goes first: it runs a block ahead of
    // the recursion and its loads are what everything else waits for (one strip of four of the
    // 2048 x 2048 grid: 0.140 -> 0.131 ms per sweep; alone on its SIMD it makes no difference).
    // (priority 3 instead of 2: the same)
    if (role == 1) __builtin_amdgcn_s_setprio(2);
    typedef std::integral_constant<int, 1> shell_t;
    typedef std::integral_constant<int, 2> rec_t;
    // Each role has its rounds in loops of its own, whole blocks apart from the last, partial
    // one: where two forms of a round meet again the compiler reconciles their registers with
    // copies, and copies of operands still in flight wait for the loads - the memory latency
    // of the prefetch once per round.
    auto round_end = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const int nwhole = T / BLK;  // whole blocks of this half
    int ib = 0;
    if (role == 1) {
      auto shell_round = [&](const int ib, auto full_tag) {
        const int tb = ib * BLK, tt = tb & (kChunk - 1);
        if (tt == 0) {
          // labels of this chunk are in LDS; fetch the next chunk's while this one runs
          if constexpr (XNEXT) labels_fetch(tb + kChunk);
        }
        block(tb, tt, full_tag, shell_t{}, ib % 3);
        if (tt + BLK == kChunk && tb + BLK < T) {
          const int t0 = tb - tt;
          wave_lds_fence();
          if constexpr (XNEXT) {
            labels_to_lds();
          } else {
            for (int i = within; i < kChunk; i += LPC) {
              const bool ok = t0 + kChunk + i < T;
              s_x[wib][cs][i] = ok ? xinchain[xdir * (t0 + kChunk + i)] : -1;
              s_xo[wib][cs][i] = ok ? xochain[xostep * (t0 + kChunk + i)] : -1;
            }
          }
          wave_lds_fence();
        }
      };
      // (every load issued so far has landed before the loop starts: the waits inside it
      // then count the loop's own loads only, instead of waiting for all of them at its head)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      for (; ib < nwhole; ib++) {
        shell_round(ib, std::true_type{});
        round_end();
      }
      if (ib * BLK < T) {
        shell_round(ib, std::false_type{});
        round_end();
        ib++;
      }
    } else {
      auto rec_round = [&](const int ib, auto full_tag) {  // block ib - 2, then the ring's block ib - 1
        const int tb = (ib - 2) * BLK, tt = tb & (kChunk - 1);
        block(tb, tt, full_tag, rec_t{}, 0);
        if constexpr (POTTS) {
          if (tt + BLK == kChunk || tb + BLK >= T) {
            uint32_t *o = bpwave + (tb / kChunk) * 256 + lane;
            if constexpr (PIPEB) {
              flush_pending();
              o[0] = (uint32_t)vk64; o[64] = (uint32_t)(vk64 >> 32); o[128] = (uint32_t)ve64; o[192] = (uint32_t)(ve64 >> 32);
            } else {
              o[0] = (uint32_t)vk0; o[64] = (uint32_t)vk1; o[128] = (uint32_t)ve0; o[192] = (uint32_t)ve1;
            }
          }
        }
      };
      auto ring_fetch = [&](const int blk) {  // the block the shell finished a round ago, for the next round
        if (blk * BLK < T) {
#pragma unroll
          for (int j = 0; j < GPB; j++) thr[SPLIT ? j : 0] = s_ring[rev][(blk % 3) * GPB + j][lane];
        }
      };
      round_end();  // round 0: the shell's first block
      ib = 1;
      ring_fetch(0);
      round_end();
      ib = 2;
      for (; ib < nwhole + 2; ib++) {
        rec_round(ib, std::true_type{});
        ring_fetch(ib - 1);
        round_end();
      }
      if ((ib - 2) * BLK < T) {
        rec_round(ib, std::false_type{});
        round_end();
        ib++;
      }
    }
    for (; ib < nblk + 2; ib++) round_end();
    if (role == 1) {
      // the shell's share of the results: the unary term of the objective; then it is done
      // (the barriers of the join do not wait for wavefronts that have ended)
#pragma unroll
      for (int d = 1; d < LPC; d <<= 1) unary_acc += __shfl_xor(unary_acc, d);
      if (cvalid && within == 0) sg.chain_unary[2 * chain + rev] = unary_acc;
      return;
    }
  }

  if constexpr (PCOMP) {
    // the last block's record (steps past the chain's end left F as it was), in its place in
    // a whole dword
    origin_settle(keep_pend);
    if (T > 0) {
      rec = __builtin_amdgcn_alignbit((unsigned)org, rec, 8);
      int r = ((T + 7) >> 3) - 1;
      for (; (r & 3) != 3; r++) rec >>= 8;
      bprec[(r >> 2) * 64 + lane] = rec;
    }
  }
  // the last block's final position is still pending
  if (!DUAL && PIPE) {
    const int tb_last = ((T - 1) / BLK) * BLK;
    if constexpr (RECOMP) {
      finish(std::integral_constant<int, BLK - 2>{}, tb_last, tb_last + BLK - 2 < T);
      locate(std::integral_constant<int, BLK - 1>{}, tb_last + BLK - 1 < T);
    }
    finish(std::integral_constant<int, BLK - 1>{}, tb_last, tb_last + BLK - 1 < T);
  }

  // ---- the two halves meet ----
  // The left half publishes its last message; the right half adds one more
  // min-plus step towards it (no unary), takes the lowest argmin of the sum
  // and hands the left half its last label.  Every wavefront of the
  // workgroup passes the same two barriers (one that has already returned is
  // not waited for).
  if constexpr (POTTS || TRUNC > 0) Mrow[b] = Mreg;
  wave_lds_fence();
  const float *Mlast = Mrow + (RECOMP ? (T & 1) * MROW : 0);  // step T-1 wrote row T & 1
  float bv = Mlast[b];
  int bi = b;
  const bool two = Tfull > Lhalf;  // the chain has a right half
  if (!rev) s_join[pair][cs * KP + b] = bv;  // the NS lanes of a label write the same value
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // lowest-index argmin of (bv, bi) inside each chain's KP lanes
  auto chain_argmin = [&]() {
#pragma unroll
    for (int d = 1; d < KP; d <<= 1) {
      const float ov = __shfl_xor(bv, d);
      const int oi = __shfl_xor(bi, d);
      const bool take = (ov < bv) || (ov == bv && oi < bi);
      bv = take ? ov : bv;
      bi = take ? oi : bi;
    }
  };
  int xjoin = 0;  // right half: the label the join gives the left half's last node
  if (rev && two) {
    // g = min over this lane's sources of (B_L[src] + W[b][src]), lowest src first
    float gbest = INF;
    int garg = 0;
#pragma unroll
    for (int j = 0; j < A; j++) {
      float wj;
      if constexpr (POTTS) wj = (j < k && bvalid) ? (j == b ? 0.0f : wpot) : INF;
      else if constexpr (TRUNC > 0) wj = (j < k && bvalid) ? Wuse[j * k + b] : INF;
      else wj = Wr[j];
      const float cnd = Mlast[s * A + j] + wj;
      if (cnd < gbest) { gbest = cnd; garg = s * A + j; }
    }
#pragma unroll
    for (int d = KP; d < LPC; d <<= 1) {
      const float ob = __shfl_xor(gbest, d);
      const int oa = __shfl_xor(garg, d);
      const bool take = (ob < gbest) || (ob == gbest && oa < garg);
      gbest = take ? ob : gbest;
      garg = take ? oa : garg;
    }
    bv = bvalid ? s_join[pair][cs * KP + b] + gbest : INF;  // J[b]
    chain_argmin();
    xjoin = bi;
    if (within == 0) s_joinx[pair][cs] = xjoin;
    bi = __shfl(garg, cs * LPC + xjoin);  // label of the right half's last node: where its trace starts
  } else {
    chain_argmin();  // a chain of one node: the left half's own argmin decides
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (!rev && two) bi = s_joinx[pair][cs];
  if (DUAL) {
    // the chain minimum is the join's (or, for one node, the left half's)
    if (cvalid && within == 0 && (two ? rev : !rev)) sg.chain_min[chain] = bv;
    return;
  }

  if constexpr (!SPLIT) {
#pragma unroll
    for (int d = 1; d < LPC; d <<= 1) unary_acc += __shfl_xor(unary_acc, d);
    if (cvalid && within == 0) sg.chain_unary[2 * chain + rev] = unary_acc;
  }
  if (T == 0) {  // right half of a one-node chain: nothing to trace
    if (cvalid && within == 0) sg.chain_lin[2 * chain + rev] = 0.0;
    return;
  }
  const float *__restrict__ Wsp = rev ? args.WsplitT : args.Wsplit;
  const int Tedge = T + rev;  // the right half also owns the edge to the left half's last node
  if (args.stamps && lane == 0) {
    args.stamps[3 * (size_t)wave] = st0;
    // where the wavefront ran: die (XCC_ID, register 20) and the HW_ID word (register 4), after the three stamps of all wavefronts
    args.stamps[3 * (size_t)args.wave_begin[args.nseg] + wave] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    args.stamps[3 * (size_t)wave + 1] = __builtin_readcyclecounter();
  }
  // backtrack, chunk by chunk from the end; one lane per chain walks the LDS
  // copy while the chunk before it is already being fetched into registers
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // this wavefront's own back-pointer stores
  const int nchunks = Tpad / kChunk;
  if constexpr (PCOMP) {
    // Trace-back of the Potts step with path compression.  The forward pass left, per step, the two lane masks
    // keep and eq (lane = step of a 64-step chunk) and, per block of eight steps, the record
    // F of the block (see "path compression" above).  In pieces of 512 steps, from the end:
    //   1. hop over the piece's 64 records on the scalar unit,
    //        x <- byte (record j) of lane x's dword,
    //      which leaves the label at the last node of every block (kept as the lane that
    //      holds it, lane j of one register per chain);
    //   2. in rounds of TB_RSTEPS steps, lane (chain, block) walks its block's eight steps,
    //        x <- bit x of keep ? x : i1,
    //      on records staged through LDS in transposed order ([step of the block][lane]), and
    //      leaves the eight labels in LDS;
    //   3. the lanes pick the labels up step by step again, store them and gather the
    //      pairwise terms of the linear objective.
    // The serial part is a step of two scalar instructions per eight nodes plus eight
    // steps per round, instead of one step per node.
    uint32_t *tbs = &s_tb[wib][0];
    int cur[CPW], carry[CPW];
    double linc[CPW];
    float wp[CPW];
    label_t *xcb[CPW];  // uniform base pointers of the chains of this wavefront
    bool cval[CPW];
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      cur[q] = q * KP + __builtin_amdgcn_readlane(bi, q * LPC);  // lane holding the chain's last label
      carry[q] = __builtin_amdgcn_readlane(xjoin, q * LPC);      // used by the right half only
      linc[q] = 0.0;
      wp[q] = 0.0f;
      const int craw = (wlocal >> 1) * CPW + q;
      cval[q] = craw < sg.nchains;
      const int cq = cval[q] ? craw : sg.nchains - 1;
      xcb[q] = sg.x + (long long)cq * Tfull + (rev ? Tfull - 1 : 0);
    }
    const int nrec = (T + 7) >> 3;
    constexpr int NCK = TB_RSTEPS / 64;
    // Records and masks are fetched one stage ahead, into the registers the stage before has
    // just finished with: the next piece's records behind the hop over this piece's, the
    // next round's masks behind the staging of this round's.
    uint32_t recs[16];
    uint32_t mk[NCK][4];
    auto load_recs = [&](const int sc) {
      const int rec0 = sc << 6;
      const int nr = min(64, nrec - rec0);
#pragma unroll
      for (int d = 0; d < 16; d++) recs[d] = (4 * d < nr) ? bprec[((rec0 >> 2) + d) * 64 + lane] : 0u;
    };
    auto load_masks = [&](const int tr0) {
#pragma unroll
      for (int ck = 0; ck < NCK; ck++) {
        const int chunk = (tr0 >> 6) + ck;
        const uint32_t *src = bpwave + (long long)chunk * 256 + lane;
#pragma unroll
        for (int i = 0; i < 4; i++) mk[ck][i] = (chunk * kChunk < T) ? src[i * 64] : 0u;
      }
    };
    const int nrounds = (T + TB_RSTEPS - 1) / TB_RSTEPS;
    load_recs((nrec - 1) >> 6);
    load_masks((nrounds - 1) * TB_RSTEPS);
    __builtin_amdgcn_sched_barrier(0);
    int xe[CPW];
#pragma unroll
    for (int q = 0; q < CPW; q++) xe[q] = 0;
    for (int g = nrounds - 1; g >= 0; g--) {
      const int tr0 = g * TB_RSTEPS;
      const int sc = tr0 >> 9;
      const int rec0 = sc << 6;
      if (g == nrounds - 1 || ((tr0 + TB_RSTEPS) & 511) == 0) {
        // 1. a new piece: labels at its blocks' last nodes
        const int nr = min(64, nrec - rec0);
        static_for<0, 64>([&](auto r_c) {
          constexpr int j = 63 - decltype(r_c)::value;
          if (j < nr) {
#pragma unroll
            for (int q = 0; q < CPW; q++) {
              const int cu = __builtin_amdgcn_readfirstlane(cur[q]);  // already uniform: keeps it scalar
              asm("v_writelane_b32 %0, %1, %2" : "+v"(xe[q]) : "s"(cu), "n"(j));
              const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)recs[j >> 2], cu);
              cur[q] = (int)((w >> ((j & 3) * 8)) & 0xffu);
            }
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        if (sc > 0) load_recs(sc - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        // 2. and 3. for the round
        wave_lds_fence();
#pragma unroll
        for (int ck = 0; ck < NCK; ck++) {
#pragma unroll
          for (int q = 0; q < CPW; q++) {
            const int at = (lane & 7) * TB_SW + q * TB_NBR + ck * 8 + (lane >> 3);
            if constexpr (KP == 64) {
              const unsigned long long ef = mk[ck][2] | ((unsigned long long)mk[ck][3] << 32);
              tbs[at] = mk[ck][0];
              tbs[8 * TB_SW + at] = mk[ck][1];
              // at the chain's first node the masks are not used (whatever they hold gives a label below KP)
              tbs[16 * TB_SW + at] = (uint32_t)__builtin_ctzll(ef | (1ull << 63));
            } else {
              constexpr uint32_t FM = KP >= 32 ? 0xFFFFFFFFu : (1u << (KP & 31)) - 1u;
              const uint32_t kf = ((q * KP < 32 ? mk[ck][0] : mk[ck][1]) >> ((q * KP) & 31)) & FM;
              const uint32_t ef = ((q * KP < 32 ? mk[ck][2] : mk[ck][3]) >> ((q * KP) & 31)) & FM;
              const uint32_t i1 = (uint32_t)__builtin_ctz(ef | (1u << (KP - 1)));
              if constexpr (KP == 32) {
                tbs[at] = kf;
                tbs[8 * TB_SW + at] = i1;
              } else {
                tbs[at] = kf | (i1 << 16);
              }
            }
          }
        }
        wave_lds_fence();
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0) load_masks(tr0 - TB_RSTEPS);
        __builtin_amdgcn_sched_barrier(0);
        // the walking lane's chain, block and first label (the shuffles need every lane)
        const int wq = lane / TB_NBR, wblk = lane % TB_NBR;
        const int gb = (tr0 >> 3) + wblk;  // block of the half chain
        int xs = 0;
#pragma unroll
        for (int q = 0; q < CPW; q++) {
          const int v = __shfl(xe[q], (gb - rec0) & 63) - q * KP;
          xs = (wq == q) ? v : xs;
        }
        uint32_t lab[8];
#pragma unroll
        for (int e = 0; e < 8; e++) lab[e] = 0u;
        if (lane < TB_NW) {
          const int s0 = T - 1 - 8 * gb;  // the block's last live step; negative: the block lies past the chain's end
          uint32_t ra[8], rb[8], rc[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            ra[e] = tbs[e * TB_SW + lane];
            rb[e] = TB_NA >= 2 ? tbs[8 * TB_SW + e * TB_SW + lane] : 0u;
            rc[e] = TB_NA >= 3 ? tbs[16 * TB_SW + e * TB_SW + lane] : 0u;
          }
          int x = xs;
#pragma unroll
          for (int e = 7; e >= 0; e--) {
            const bool on = e <= s0;
            lab[e] = on ? (uint32_t)x : 0u;
            if (e > 0) {
              // x <- bit x of keep ? x : i1, branch-free
              int kept, i1;
              if constexpr (KP == 64) {
                kept = (int)(((ra[e] | ((unsigned long long)rb[e] << 32)) >> x) & 1ull);
                i1 = (int)rc[e];
              } else if constexpr (KP == 32) {
                kept = (int)((ra[e] >> x) & 1u);
                i1 = (int)rb[e];
              } else {
                kept = (int)((ra[e] >> x) & 1u);
                i1 = (int)(ra[e] >> 16);
              }
              const int nx = kept ? x : i1;
              x = on ? nx : x;
            }
          }
        }
        wave_lds_fence();
        if (lane < TB_NW) {
          uint4 *dst = reinterpret_cast<uint4 *>(tbs + wq * TB_RSTEPS + wblk * 8);
          dst[0] = uint4{lab[0], lab[1], lab[2], lab[3]};
          dst[1] = uint4{lab[4], lab[5], lab[6], lab[7]};
        }
        wave_lds_fence();
        // labels out, and the round's terms of the linear objective
#pragma unroll
        for (int ck = NCK - 1; ck >= 0; ck--) {
          const int t0 = tr0 + ck * kChunk;
          if (t0 >= T) continue;
          const int n = min(kChunk, T - t0);
#pragma unroll
          for (int q = 0; q < CPW; q++) {
            linc[q] += (double)wp[q];
            const int xi = (int)tbs[q * TB_RSTEPS + ck * kChunk + lane];
            int xn = __shfl_down(xi, 1);
            xn = (lane == n - 1) ? carry[q] : xn;
            const int t = t0 + lane;
            wp[q] = 0.0f;
            if (lane < n) {
              if (cval[q]) xcb[q][xdir * t] = (label_t)xi;
              if (t + 1 < Tedge) wp[q] = Wsp[xi * k + xn];
            }
            carry[q] = __builtin_amdgcn_readlane(xi, 0);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      double l = linc[q] + (double)wp[q];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) l += __shfl_xor(l, d);
      if (cval[q] && lane == 0) sg.chain_lin[2 * ((wlocal >> 1) * CPW + q) + rev] = l;
    }
    if (args.stamps && lane == 0) args.stamps[3 * (size_t)wave + 2] = __builtin_readcyclecounter();
    return;
  }
  if constexpr (POTTS && CPW >= 2) {
    // Two or more chains per wavefront: one lane per chain walks.  Lane tt
    // (= step tt of the chunk) cuts every chain's field out of the two masks
    // and leaves  keep field | i1 << 16  (32 labels: the two as a pair of
    // words) in LDS, row = chain; lane q then reads its chain's row 16 bytes at
    // a time, walks it in registers
    //   x_{t-1} = bit x_t of keep ? x_t : i1
    // (a signed one-bit field extract and a bit-field insert per step) and puts
    // the labels where the words were, for the lanes to pick up step by step
    // again.  Rows are padded by four dwords so that the 16-byte accesses of the
    // walking lanes fall into different banks.  With two chains of 32 labels
    // this replaces a walk on the scalar unit that cost ten lane reads, lane
    // writes and scalar instructions per step and wavefront.
    constexpr int WPS = KP > 16 ? 2 : 1;  // words per step
    constexpr int RS = kChunk * WPS + 4;
    static_assert(CPW * RS <= CPW * 16 * KP, "the staging rows fit the back-pointer buffer");
    uint32_t *stg = &s_bp[wib][0][0];
    uint32_t mr[4], mn[4], mf[4];  // this chunk, the next one, in flight
    auto load_chunk = [&](uint32_t (&dst)[4], int ci) {
      const uint32_t *src = bpwave + (long long)ci * 256 + lane;
#pragma unroll
      for (int i = 0; i < 4; i++) dst[i] = src[i * 64];
    };
    int cur = __shfl(bi, (lane < CPW ? lane : 0) * LPC);  // lane q: the last label of chain q
    int carry[CPW];
    double linc[CPW];
    float wp[CPW];
    label_t *xcb[CPW];  // uniform base pointers of the chains of this wavefront
    bool cval[CPW];
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      carry[q] = __builtin_amdgcn_readlane(xjoin, q * LPC);   // used by the right half only
      linc[q] = 0.0;
      wp[q] = 0.0f;
      const int craw = (wlocal >> 1) * CPW + q;
      cval[q] = craw < sg.nchains;
      const int cq = cval[q] ? craw : sg.nchains - 1;
      xcb[q] = sg.x + (long long)cq * Tfull + (rev ? Tfull - 1 : 0);
    }
    load_chunk(mr, nchunks - 1);
    if (nchunks > 1) load_chunk(mn, nchunks - 2);
    for (int ci = nchunks - 1; ci >= 0; ci--) {
      const int t0 = ci * kChunk;
      const int n = min(kChunk, T - t0);
      if (ci > 1) load_chunk(mf, ci - 2);
#pragma unroll
      for (int q = 0; q < CPW; q++) {
        constexpr uint32_t FM = KP >= 32 ? 0xFFFFFFFFu : (1u << (KP & 31)) - 1u;
        const uint32_t kf = ((q * KP < 32 ? mr[0] : mr[1]) >> ((q * KP) & 31)) & FM;
        const uint32_t ef = ((q * KP < 32 ? mr[2] : mr[3]) >> ((q * KP) & 31)) & FM;
        // at the chain's first node the masks are not used (whatever they hold
        // gives a label below KP)
        const uint32_t i1 = (uint32_t)__builtin_ctz(ef | (1u << (KP - 1)));
        if constexpr (WPS == 2) *reinterpret_cast<uint2 *>(stg + q * RS + 2 * lane) = uint2{kf, i1};
        else stg[q * RS + lane] = kf | (i1 << 16);
      }
      wave_lds_fence();
      if (lane < CPW) {
        uint4 *row = reinterpret_cast<uint4 *>(stg + lane * RS);
        // x <- bit x of keep ? x : i1, branch-free: m = all ones or zero, then a bitwise select
        auto step = [&](const uint32_t kw, const uint32_t i1) {
          const int m = -(int)((kw >> cur) & 1u);
          cur = (cur & m) | ((int)i1 & ~m);
        };
        // four steps (one or two 16-byte reads), labels written back to each step's first word
        auto walk4 = [&](auto g_c, const bool guarded) {
          constexpr int g = decltype(g_c)::value;
          uint32_t o[4] = {0u, 0u, 0u, 0u};
          if constexpr (WPS == 2) {
            const uint4 hi = row[2 * g + 1], lo = row[2 * g];
            const uint32_t kw[4] = {lo.x, lo.z, hi.x, hi.z}, iw[4] = {lo.y, lo.w, hi.y, hi.w};
#pragma unroll
            for (int e = 3; e >= 0; e--) {
              if (!guarded || 4 * g + e < n) {
                o[e] = (uint32_t)cur;
                step(kw[e], iw[e]);
              }
            }
            row[2 * g] = uint4{o[0], 0u, o[1], 0u};
            row[2 * g + 1] = uint4{o[2], 0u, o[3], 0u};
          } else {
            const uint4 w4 = row[g];
            const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int e = 3; e >= 0; e--) {
              if (!guarded || 4 * g + e < n) {
                o[e] = (uint32_t)cur;
                step(w[e], w[e] >> 16);
              }
            }
            row[g] = uint4{o[0], o[1], o[2], o[3]};
          }
        };
        if (n == kChunk) {
          static_for<0, kChunk / 4>([&](auto r_c) { walk4(std::integral_constant<int, kChunk / 4 - 1 - decltype(r_c)::value>{}, false); });
        } else {
          static_for<0, kChunk / 4>([&](auto r_c) {
            constexpr int g = kChunk / 4 - 1 - decltype(r_c)::value;
            if (4 * g < n) walk4(std::integral_constant<int, g>{}, true);
          });
        }
      }
      wave_lds_fence();
      int xr[CPW];
#pragma unroll
      for (int q = 0; q < CPW; q++) xr[q] = (int)stg[q * RS + WPS * lane];
      wave_lds_fence();
      // labels out, and this chunk's terms of the linear objective
#pragma unroll
      for (int q = 0; q < CPW; q++) {
        linc[q] += (double)wp[q];
        const int xi = xr[q];
        int xn = __shfl_down(xi, 1);
        xn = (lane == n - 1) ? carry[q] : xn;
        const int t = t0 + lane;
        wp[q] = 0.0f;
        if (lane < n) {
          if (cval[q]) xcb[q][xdir * t] = (label_t)xi;
          if (t + 1 < Tedge) wp[q] = Wsp[xi * k + xn];
        }
        carry[q] = __builtin_amdgcn_readlane(xi, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        mr[i] = mn[i];
        mn[i] = mf[i];
      }
    }
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      double l = linc[q] + (double)wp[q];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) l += __shfl_xor(l, d);
      if (cval[q] && lane == 0) sg.chain_lin[2 * ((wlocal >> 1) * CPW + q) + rev] = l;
    }
    if (args.stamps && lane == 0) args.stamps[3 * (size_t)wave + 2] = __builtin_readcyclecounter();
    return;
  } else if constexpr (POTTS) {
    // One chain of 64 labels per wavefront.
    // Lane tt holds the two masks of step tt of the chunk; the walk runs on the
    // scalar unit: x_{t-1} = x_t when bit x_t of the chain's field of keep is set,
    // else the lowest set bit of the chain's field of eq.  Labels are collected
    // in one register per chain (lane tt = step tt of the chunk).
    uint32_t mr[4], mn[4], mf[4];  // this chunk, the next one, in flight
    auto load_chunk = [&](uint32_t (&dst)[4], int ci) {
      const uint32_t *src = bpwave + (long long)ci * 256 + lane;
#pragma unroll
      for (int i = 0; i < 4; i++) dst[i] = src[i * 64];
    };
    int curs[CPW], carry[CPW];
    double linc[CPW];
    float wp[CPW];
    label_t *xcb[CPW];  // uniform base pointers of the chains of this wavefront
    bool cval[CPW];
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      curs[q] = __builtin_amdgcn_readlane(bi, q * LPC);       // the chain's last label
      carry[q] = __builtin_amdgcn_readlane(xjoin, q * LPC);   // used by the right half only
      linc[q] = 0.0;
      wp[q] = 0.0f;
      const int craw = (wlocal >> 1) * CPW + q;
      cval[q] = craw < sg.nchains;
      const int cq = cval[q] ? craw : sg.nchains - 1;
      xcb[q] = sg.x + (long long)cq * Tfull + (rev ? Tfull - 1 : 0);
    }
    load_chunk(mr, nchunks - 1);
    if (nchunks > 1) load_chunk(mn, nchunks - 2);
    for (int ci = nchunks - 1; ci >= 0; ci--) {
      const int t0 = ci * kChunk;
      const int n = min(kChunk, T - t0);
      if (ci > 1) load_chunk(mf, ci - 2);
      __builtin_amdgcn_sched_barrier(0);
      int xr[CPW];
#pragma unroll
      for (int q = 0; q < CPW; q++) xr[q] = 0;
      auto walk_step = [&](auto tt_c) {
        constexpr int tt = decltype(tt_c)::value;
        const unsigned long long keepm = (unsigned)__builtin_amdgcn_readlane((int)mr[0], tt) |
                                         ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mr[1], tt) << 32);
        const unsigned long long eqm = (unsigned)__builtin_amdgcn_readlane((int)mr[2], tt) |
                                       ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mr[3], tt) << 32);
#pragma unroll
        for (int q = 0; q < CPW; q++) {
          const int cu = __builtin_amdgcn_readfirstlane(curs[q]);  // already uniform: keeps it scalar
          asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(xr[q]) : "s"(cu), "n"(tt));
          // at the chain's first node the masks are not used (whatever they hold
          // gives a label below KP)
          const int i1 = __builtin_ctzll((eqm >> (q * KP)) | (1ull << (KP - 1))) & (KP - 1);
          curs[q] = ((keepm >> (q * KP + cu)) & 1) ? cu : i1;
        }
      };
      if (n == kChunk) {
        static_for<0, kChunk>([&](auto r_c) { walk_step(std::integral_constant<int, kChunk - 1 - decltype(r_c)::value>{}); });
      } else {
        static_for<0, kChunk>([&](auto r_c) {
          constexpr int tt = kChunk - 1 - decltype(r_c)::value;
          if (tt < n) walk_step(std::integral_constant<int, tt>{});
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      // labels out, and this chunk's terms of the linear objective
#pragma unroll
      for (int q = 0; q < CPW; q++) {
        linc[q] += (double)wp[q];
        const int xi = xr[q];
        int xn = __shfl_down(xi, 1);
        xn = (lane == n - 1) ? carry[q] : xn;
        const int t = t0 + lane;
        if (n == kChunk && cval[q]) {
          xcb[q][xdir * t] = (label_t)xi;
          const float wv = Wsp[xi * k + xn];
          wp[q] = (t + 1 < Tedge) ? wv : 0.0f;
        } else {
          wp[q] = 0.0f;
          if (lane < n) {
            if (cval[q]) xcb[q][xdir * t] = (label_t)xi;
            if (t + 1 < Tedge) wp[q] = Wsp[xi * k + xn];
          }
        }
        carry[q] = __builtin_amdgcn_readlane(xi, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        mr[i] = mn[i];
        mn[i] = mf[i];
      }
    }
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      double l = linc[q] + (double)wp[q];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) l += __shfl_xor(l, d);
      if (cval[q] && lane == 0) sg.chain_lin[2 * ((wlocal >> 1) * CPW + q) + rev] = l;
    }
    if (args.stamps && lane == 0) args.stamps[3 * (size_t)wave + 2] = __builtin_readcyclecounter();
    return;
  }
  if constexpr (SWALK) {
    // One or two chains per wavefront: the walk runs on the scalar unit.  Lane
    // b keeps its label's sixteen back-pointer dwords of the chunk in
    // registers; a step is one v_readlane (lane = current label) and one bit
    // field extract, with no LDS round trip.  The labels are collected in one
    // register per chain (lane tt = step tt of the chunk).
    uint32_t bpr[16], bpn[16];
    auto load_chunk = [&](uint32_t (&dst)[16], int ci) {
      const uint32_t *src = bpchain + (long long)ci * 16 * KP + b;
#pragma unroll
      for (int i = 0; i < 16; i++) dst[i] = src[i * KP];
    };
    int curs[CPW], carry[CPW], wtmp[2] = {0, 0};
    double linc[CPW];
    float wp[CPW];
    label_t *xcb[CPW];  // uniform base pointers of the chains of this wavefront
    bool cval[CPW];
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      curs[q] = q * LPC + __builtin_amdgcn_readlane(bi, q * LPC);  // lane holding the chain's last label
      carry[q] = __builtin_amdgcn_readlane(xjoin, q * LPC);  // used by the right half only
      linc[q] = 0.0;
      wp[q] = 0.0f;
      const int craw = (wlocal >> 1) * CPW + q;
      cval[q] = craw < sg.nchains;
      const int cq = cval[q] ? craw : sg.nchains - 1;
      xcb[q] = sg.x + (long long)cq * Tfull + (rev ? Tfull - 1 : 0);
    }
    // back-pointers are fetched two chunks ahead of the walk (bpr: this chunk,
    // bpn: the next one, bpf: in flight); scheduling barriers keep the loads
    // where they are written
    uint32_t bpf[16];
    load_chunk(bpr, nchunks - 1);
    if (nchunks > 1) load_chunk(bpn, nchunks - 2);
    for (int ci = nchunks - 1; ci >= 0; ci--) {
      const int t0 = ci * kChunk;
      const int n = min(kChunk, T - t0);
      if (ci > 1) load_chunk(bpf, ci - 2);
      __builtin_amdgcn_sched_barrier(0);
      int xr[CPW];
#pragma unroll
      for (int q = 0; q < CPW; q++) {
        xr[q] = 0;
        curs[q] = __builtin_amdgcn_readfirstlane(curs[q]);  // uniform already; pins it to a scalar register
      }
      // a lane select written by a vector instruction needs four wait states,
      // and nothing inserts them inside an asm statement
      asm volatile("s_nop 3");
      if (n == kChunk) {
        // whole chunk: four steps per back-pointer dword, hand-written so that
        // the walk stays on the scalar unit: record the lane, fetch that
        // lane's dword, extract the step's byte (= the previous lane)
        static_for<0, 16>([&](auto r_c) {
          constexpr int i = 15 - decltype(r_c)::value;
          // named references: operands of an asm statement alone do not make
          // a generic lambda capture a variable
          int &x0 = xr[0], &x1 = xr[CPW - 1], &c0 = curs[0], &c1 = curs[CPW - 1], &w0 = wtmp[0], &w1 = wtmp[1];
          const uint32_t bpw = bpr[i];
          if constexpr (CPW == 2) {
            asm volatile(
                "v_writelane_b32 %[x0], %[c0], %[t3]\n\tv_writelane_b32 %[x1], %[c1], %[t3]\n\t"
                "v_readlane_b32 %[w0], %[bp], %[c0]\n\tv_readlane_b32 %[w1], %[bp], %[c1]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80018\n\ts_bfe_u32 %[c1], %[w1], 0x80018\n\ts_nop 1\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t2]\n\tv_writelane_b32 %[x1], %[c1], %[t2]\n\t"
                "v_readlane_b32 %[w0], %[bp], %[c0]\n\tv_readlane_b32 %[w1], %[bp], %[c1]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80010\n\ts_bfe_u32 %[c1], %[w1], 0x80010\n\ts_nop 1\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t1]\n\tv_writelane_b32 %[x1], %[c1], %[t1]\n\t"
                "v_readlane_b32 %[w0], %[bp], %[c0]\n\tv_readlane_b32 %[w1], %[bp], %[c1]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80008\n\ts_bfe_u32 %[c1], %[w1], 0x80008\n\ts_nop 1\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t0]\n\tv_writelane_b32 %[x1], %[c1], %[t0]\n\t"
                "v_readlane_b32 %[w0], %[bp], %[c0]\n\tv_readlane_b32 %[w1], %[bp], %[c1]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80000\n\ts_bfe_u32 %[c1], %[w1], 0x80000\n\ts_nop 1"
                : [x0] "+v"(x0), [x1] "+v"(x1), [c0] "+s"(c0), [c1] "+s"(c1), [w0] "=&s"(w0), [w1] "=&s"(w1)
                : [bp] "v"(bpw), [t3] "n"(4 * i + 3), [t2] "n"(4 * i + 2), [t1] "n"(4 * i + 1), [t0] "n"(4 * i));
          } else {
            asm volatile(
                "v_writelane_b32 %[x0], %[c0], %[t3]\n\tv_readlane_b32 %[w0], %[bp], %[c0]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80018\n\ts_nop 3\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t2]\n\tv_readlane_b32 %[w0], %[bp], %[c0]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80010\n\ts_nop 3\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t1]\n\tv_readlane_b32 %[w0], %[bp], %[c0]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80008\n\ts_nop 3\n\t"
                "v_writelane_b32 %[x0], %[c0], %[t0]\n\tv_readlane_b32 %[w0], %[bp], %[c0]\n\t"
                "s_bfe_u32 %[c0], %[w0], 0x80000\n\ts_nop 3"
                : [x0] "+v"(x0), [c0] "+s"(c0), [w0] "=&s"(w0)
                : [bp] "v"(bpw), [t3] "n"(4 * i + 3), [t2] "n"(4 * i + 2), [t1] "n"(4 * i + 1), [t0] "n"(4 * i));
          }
        });
      } else {
        static_for<0, kChunk>([&](auto r_c) {
          constexpr int tt = kChunk - 1 - decltype(r_c)::value;
          if (tt < n) {
#pragma unroll
            for (int q = 0; q < CPW; q++) {
              const int cu = __builtin_amdgcn_readfirstlane(curs[q]);  // already uniform: keeps it scalar
              asm("v_writelane_b32 %0, %1, %2" : "+v"(xr[q]) : "s"(cu), "n"(tt));
              // at the chain's first node the value read is not used
              const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)bpr[tt >> 2], cu);
              curs[q] = (int)((w >> ((tt & 3) * 8)) & 0xffu);
            }
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
      // labels out, and this chunk's terms of the linear objective.  No
      // divergent branch around the memory operations of a whole chunk, so
      // that the wait for the next chunk's back-pointers can be a counted one
      // and does not drain these gathers.
#pragma unroll
      for (int q = 0; q < CPW; q++) {
        linc[q] += (double)wp[q];
        const int xi = xr[q] & (KP - 1);  // lane -> label
        int xn = __shfl_down(xi, 1);
        xn = (lane == n - 1) ? carry[q] : xn;
        const int t = t0 + lane;
        if (n == kChunk && cval[q]) {
          xcb[q][xdir * t] = (label_t)xi;
          const float wv = Wsp[xi * k + xn];
          wp[q] = (t + 1 < Tedge) ? wv : 0.0f;
        } else {
          wp[q] = 0.0f;
          if (lane < n) {
            if (cval[q]) xcb[q][xdir * t] = (label_t)xi;
            if (t + 1 < Tedge) wp[q] = Wsp[xi * k + xn];
          }
        }
        carry[q] = __builtin_amdgcn_readlane(xi, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        bpr[i] = bpn[i];
        bpn[i] = bpf[i];
      }
    }
#pragma unroll
    for (int q = 0; q < CPW; q++) {
      double l = linc[q] + (double)wp[q];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) l += __shfl_xor(l, d);
      if (cval[q] && lane == 0) sg.chain_lin[2 * ((wlocal >> 1) * CPW + q) + rev] = l;
    }
    if (args.stamps && lane == 0) args.stamps[3 * (size_t)wave + 2] = __builtin_readcyclecounter();
    return;
  }
  constexpr int VW = DW >= 4 ? 4 : DW;  // dwords per load
  constexpr int NV = DW / VW;
  uint32_t stage[NV][VW];
  auto fetch_chunk = [&](int ci) {
    const uint32_t *src = bpchain + (long long)ci * 16 * KP;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int o = (v * LPC + within) * VW;
      if constexpr (VW == 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(src + o);
        stage[v][0] = q.x; stage[v][1] = q.y; stage[v][2] = q.z; stage[v][3] = q.w;
      } else if constexpr (VW == 2) {
        const uint2 q = *reinterpret_cast<const uint2 *>(src + o);
        stage[v][0] = q.x; stage[v][1] = q.y;
      } else {
        stage[v][0] = src[o];
      }
    }
  };
  int cur = bi;
  // The pairwise term of the chain's energy (Grid.cpp:1665-1680): sum of
  // W[x_t][x_t+1] over the new labels, gathered chunk by chunk by all lanes.
  double lin = 0.0;
  constexpr int NI = kChunk / LPC;   // label positions per lane per chunk
  constexpr bool DEFER = NI <= 4;    // consume a chunk's gathers one chunk later
  float wp[DEFER ? NI : 1];
#pragma unroll
  for (int q = 0; q < (DEFER ? NI : 1); q++) wp[q] = 0.0f;
  int carry = xjoin;  // label of the first node after the current chunk (right half: the left half's last node)
  fetch_chunk(nchunks - 1);
  for (int ci = nchunks - 1; ci >= 0; ci--) {
    const int t0 = ci * kChunk;
    const int n = min(kChunk, T - t0);
    wave_lds_fence();
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int o = (v * LPC + within) * VW;
#pragma unroll
      for (int e = 0; e < VW; e++) s_bp[wib][cs][o + e] = stage[v][e];
    }
    wave_lds_fence();
    if (ci > 0) fetch_chunk(ci - 1);
    // Whole chunks, 8 or 16 lanes per chain: the walk in three short phases instead of 64
    // dependent LDS reads by one lane.  (A) every lane walks each of the chunk's eight blocks
    // of eight steps from its own label as the block's last one - eight independent walks
    // per lane, their reads overlap - which leaves, per block, the label before the block as
    // a function of the label at its end; (B) one lane per chain hops over the eight blocks
    // (8 dependent reads) and leaves every block's last label; (C) lane j of the chain walks
    // block j (8 dependent reads).  The tables of (A) and the labels of (B) live in the
    // wavefront's row of s_xo, which the forward pass is done with.
    constexpr bool PWALK = NS == 1 && (KP == 8 || KP == 16);
    if (PWALK && n == kChunk) {
      const uint8_t *bytes = reinterpret_cast<const uint8_t *>(&s_bp[wib][cs][0]);
      uint8_t *gtab = reinterpret_cast<uint8_t *>(&s_xo[wib][cs][0]);  // [block][label]
      int *xend = &s_xo[wib][cs][(8 * KP) / 4];                         // [block]
      int v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = b;
#pragma unroll
      for (int st = 7; st >= 0; st--) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int tt = 8 * j + st;
          v[j] = bytes[(v[j] << 2) + (((tt >> 2) * KP) << 2) + (tt & 3)];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) gtab[j * KP + b] = (uint8_t)v[j];
      wave_lds_fence();
      if (within == 0) {
#pragma unroll
        for (int j = 7; j >= 0; j--) {
          xend[j] = cur;
          cur = gtab[j * KP + cur];  // (the chain's first chunk: the value after block 0 is not used)
        }
      }
      wave_lds_fence();
      if (within < 8) {
        int c = xend[within];
        const int base = within * 2 * KP * 4;  // block `within` starts at step 8 * within: dword 2 * within of every label
#pragma unroll
        for (int st = 7; st >= 0; st--) {
          s_x[wib][cs][8 * within + st] = c;
          if (st > 0) c = bytes[base + (c << 2) + (((st >> 2) * KP) << 2) + (st & 3)];
        }
      }
    } else if (within == 0) {
      const uint8_t *bytes = reinterpret_cast<const uint8_t *>(&s_bp[wib][cs][0]);
      if (n == kChunk && ci > 0) {
        // whole chunk, not the chain's first: every offset but the label is static
#pragma unroll
        for (int tt = kChunk - 1; tt >= 0; tt--) {
          s_x[wib][cs][tt] = cur;
          cur = bytes[(cur << 2) + (((tt >> 2) * KP) << 2) + (tt & 3)];
        }
      } else {
        for (int tt = n - 1; tt >= 0; tt--) {
          s_x[wib][cs][tt] = cur;
          if (t0 + tt > 0) cur = bytes[(((tt >> 2) * KP + cur) << 2) + (tt & 3)];
        }
      }
    }
    cur = __shfl(cur, cs * LPC);
    wave_lds_fence();
    if (DEFER) {
#pragma unroll
      for (int q = 0; q < NI; q++) {
        lin += (double)wp[q];
        wp[q] = 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < NI; q++) {
      const int i = within + q * LPC;
      if (i < n) {
        const int xi = s_x[wib][cs][i];
        if (cvalid) xchain[xdir * (t0 + i)] = (label_t)xi;
        const int t = t0 + i;
        float wv = 0.0f;
        if (t + 1 < Tedge) {
          const int xn = (i + 1 < n) ? s_x[wib][cs][i + 1] : carry;
          wv = Wsp[xi * k + xn];
        }
        if (DEFER) wp[q] = wv;
        else lin += (double)wv;
      }
    }
    carry = s_x[wib][cs][0];
    wave_lds_fence();
  }
  if (DEFER) {
#pragma unroll
    for (int q = 0; q < NI; q++) {
      lin += (double)wp[q];
    }
  }
#pragma unroll
  for (int d = 1; d < LPC; d <<= 1) lin += __shfl_xor(lin, d);
  if (cvalid && within == 0) sg.chain_lin[2 * chain + rev] = lin;
  if (args.stamps && lane == 0) args.stamps[3 * (size_t)wave + 2] = __builtin_readcyclecounter();
}

// Consensus of a strip of nr x nc nodes from the two label arrays:
//   m[r][c][l] = (mu(xa[r][c])[l] + mu(xbT[c][r])[l]) / 2      Grid.cpp:163-181
// xa is row-major (labels of the row chains), xbT is column-major (labels of
// the column chains, chain-major as the sweep writes them).  A workgroup
// handles a 32 x 32 tile; the transposed read goes through LDS.
template <typename TB>
__global__ __launch_bounds__(256) void chain_consensus_kernel(float *__restrict__ m,
                                                              const label_t *__restrict__ xa,
                                                              const TB *__restrict__ xbT, int nr,
                                                              int nc, int k, float inv_k,
                                                              int tiles_c, int tiles_per_grid) {
  __shared__ int s_a[32][33];
  __shared__ int s_b[32][33];
  const int g = blockIdx.x / tiles_per_grid;
  const int tile = blockIdx.x % tiles_per_grid;
  const int r0 = (tile / tiles_c) * 32, c0 = (tile % tiles_c) * 32;
  const long long gnode = (long long)g * nr * nc;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int hi = i >> 5, lo = i & 31;
    // xa[r0+hi][c0+lo]
    s_a[hi][lo] = (r0 + hi < nr && c0 + lo < nc) ? xa[gnode + (long long)(r0 + hi) * nc + c0 + lo] : -1;
    // xbT[c0+hi][r0+lo]
    s_b[lo][hi] = (c0 + hi < nc && r0 + lo < nr) ? xbT[gnode + (long long)(c0 + hi) * nr + r0 + lo] : -1;
  }
  __syncthreads();
  if ((k & 3) == 0) {
    const int q4 = k >> 2;
    const int per_row = 32 * q4;
    for (int i = threadIdx.x; i < 32 * per_row; i += 256) {
      const int rl = i / per_row, rem = i % per_row;
      const int cl = rem / q4, q = rem % q4;
      if (r0 + rl >= nr || c0 + cl >= nc) continue;
      const int x1 = s_a[rl][cl], x2 = s_b[rl][cl];
      float4 v;
      float *pv = reinterpret_cast<float *>(&v);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float sum = 0.0f;
        sum += chain_mu(x1, 4 * q + e, inv_k);
        sum += chain_mu(x2, 4 * q + e, inv_k);
        pv[e] = sum / 2.0f;
      }
      *reinterpret_cast<float4 *>(m + ((gnode + (long long)(r0 + rl) * nc + c0 + cl) * k + 4 * q)) = v;
    }
  } else {
    const int per_row = 32 * k;
    for (int i = threadIdx.x; i < 32 * per_row; i += 256) {
      const int rl = i / per_row, rem = i % per_row;
      const int cl = rem / k, l = rem % k;
      if (r0 + rl >= nr || c0 + cl >= nc) continue;
      float sum = 0.0f;
      sum += chain_mu(s_a[rl][cl], l, inv_k);
      sum += chain_mu(s_b[rl][cl], l, inv_k);
      m[(gnode + (long long)(r0 + rl) * nc + c0 + cl) * k + l] = sum / 2.0f;
    }
  }
}

__device__ __forceinline__ int chain_decode(int xr, int xc) {
  // argmax of the consensus, lowest index on ties
  if (xr < 0 || xc < 0) return 0;
  return xr < xc ? xr : xc;
}

// Objective terms of one set of chains, one thread per node, on chain-major
// label arrays (x_own: the chains' own labels, x_oth: the labels the crossing
// chains gave the same nodes).  d = decoded label = argmax of the consensus.
//   primal += [mode & 1] 2 * theta_share[d]  +  W[d][d_next]
//   linear +=            theta_share[x_own]  +  W[x_own][x_own_next]
// mode & 2: only linear += theta_share[x_own] (primal untouched).
// "next" is the next node along the chain.  The row chains are run with
// mode 1 and the column chains with mode 0, so every unary and every edge is
// counted once.  linear is the chain analogue of Grid.cpp:1665-1680 (every
// tree's own solution under its share of the potentials).  Each workgroup
// writes one (primal, linear) partial; binary64 accumulation throughout.
__global__ __launch_bounds__(256) void chain_objective_kernel(
    double *__restrict__ partial, const float *__restrict__ theta, const float *__restrict__ Wsplit,
    const label_t *__restrict__ x_own, const label_t *__restrict__ x_oth, int nchains, int T, int cpg,
    long long gstride, long long cstride, long long sstride, int k, int mode) {
  __shared__ float s_W[64 * 64];
  __shared__ double s_red[2][4];
  for (int i = threadIdx.x; i < k * k; i += 256) s_W[i] = Wsplit[i];
  __syncthreads();
  const long long total = (long long)nchains * T;
  double p = 0.0, lin = 0.0;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int chain = (int)(idx / T), t = (int)(idx % T);
    const int xo = x_own[idx], xt = x_oth[idx];
    const int d = chain_decode(xo, xt);
    const int xo0 = xo < 0 ? 0 : xo;
    const long long node = (long long)(chain / cpg) * gstride + (long long)(chain % cpg) * cstride + (long long)t * sstride;
    const float *th = theta + node * k;
    if (mode & 2) {  // unary term of the chains' own labels only
      lin += (double)th[xo0];
      continue;
    }
    if (mode & 1) {
      p += (double)th[d];
      p += (double)th[d];
    }
    lin += (double)th[xo0];
    if (t + 1 < T) {
      const int xo1 = x_own[idx + 1], xt1 = x_oth[idx + 1];
      p += (double)s_W[d * k + chain_decode(xo1, xt1)];
      lin += (double)s_W[xo0 * k + (xo1 < 0 ? 0 : xo1)];
    }
  }
  p = wave_sum(p);
  lin = wave_sum(lin);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = p; s_red[1][threadIdx.x >> 6] = lin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    partial[2 * blockIdx.x + 1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
  }
}

// One transposition: out[g][c][r] = in[g][r][c] for g < batch, r < nr, c < nc.
struct TransposeJob {
  label_t *out;
  const void *in;  // label_t, or int32 (labels as they arrive from other strips)
  int nr, nc, tiles_c, tiles_per_grid;
  int block_begin;  // first workgroup of this job
};
struct TransposeArgs {
  TransposeJob job[2];
  int njobs;
};

// Up to two transpositions in one launch (Update needs both label arrays laid
// out along the crossing chains); 64 x 64 tiles through LDS, a wavefront reads
// and writes 64 consecutive labels.
template <typename TI>
__global__ __launch_bounds__(256) void transpose_labels_kernel(const TransposeArgs args) {
  __shared__ label_t s_t[64][64 + 4];
  const int ji = (args.njobs > 1 && (int)blockIdx.x >= args.job[1].block_begin) ? 1 : 0;
  const TransposeJob &jb = args.job[ji];
  label_t *__restrict__ out = jb.out;
  const TI *__restrict__ in = static_cast<const TI *>(jb.in);
  const int nr = jb.nr, nc = jb.nc, tiles_c = jb.tiles_c, tiles_per_grid = jb.tiles_per_grid;
  const int blk = blockIdx.x - jb.block_begin;
  const int g = blk / tiles_per_grid;
  const int tile = blk % tiles_per_grid;
  const int r0 = (tile / tiles_c) * 64, c0 = (tile % tiles_c) * 64;
  const long long base = (long long)g * nr * nc;
  // every load of the tile in flight at once: no branch around a load (rows and
  // columns past the edge read the edge again and are never written out)
  label_t v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    const int r = min(r0 + (i >> 6), nr - 1), cc = min(c0 + (i & 63), nc - 1);
    v[j] = (label_t)in[base + (long long)r * nc + cc];
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    s_t[i >> 6][i & 63] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    const int hi = i >> 6, lo = i & 63;
    if (c0 + hi < nc && r0 + lo < nr) out[base + (long long)(c0 + hi) * nr + r0 + lo] = s_t[lo][hi];
  }
}

// Splits n items over G ranks as admm_partition does and returns the rank that
// owns item i, with that rank's first item and count.
__device__ __forceinline__ int owner_of(int i, int n, int G, int *start, int *count) {
  const int base = n / G, extra = n % G;
  const int cut = extra * (base + 1);
  int h;
  if (i < cut) { h = i / (base + 1); *start = h * (base + 1); *count = base + 1; }
  else { h = extra + (i - cut) / base; *start = cut + (h - extra) * base; *count = base; }
  return h;
}

// All send blocks of the strip exchange in one launch.  a-blocks: my
// row-chain labels xrow[nr][cols] cut by the peers' column strips, block h =
// [nr][nc_h]; b-blocks: my column-chain labels xcolT[nc][rows] cut by the
// peers' row strips, block h = [nc][nr_h].  The blocks of a kind lie in rank
// order, so block h of the a-kind starts at nr * c0_h and of the b-kind at
// nc * r0_h.  With own_a / own_b set, the blocks this rank would send to
// itself (rank self) go straight to where it would receive them.
__global__ void pack_strip_labels_kernel(int *__restrict__ send_a, int *__restrict__ send_b,
                                         const label_t *__restrict__ xrow, const label_t *__restrict__ xcolT,
                                         int nr, int nc, int rows, int cols, int G, int self,
                                         int *__restrict__ own_a, int *__restrict__ own_b) {
  const long long na = (long long)nr * cols, nb = (long long)nc * rows;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < na + nb;
       i += (long long)gridDim.x * blockDim.x) {
    int st, cnt;
    if (i < na) {
      const int r = (int)(i / cols), cidx = (int)(i % cols);
      const int h = owner_of(cidx, cols, G, &st, &cnt);
      int *dst = (own_a && h == self) ? own_a : send_a + (long long)nr * st;
      dst[(long long)r * cnt + (cidx - st)] = xrow[i];
    } else {
      const long long j = i - na;
      const int cc = (int)(j / rows), ridx = (int)(j % rows);
      const int h = owner_of(ridx, rows, G, &st, &cnt);
      int *dst = (own_b && h == self) ? own_b : send_b + (long long)nc * st;
      dst[(long long)cc * cnt + (ridx - st)] = xcolT[j];
    }
  }
}

// The strip exchange and Update's transposition in one launch, for strips whose label arrays
// this device can read directly (all strips of a problem held by one process): strip h PULLS.
// Job a, for its column chains: the labels of every strip's row chains on h's columns,
//   v = xrow_g[r - r0_g][c0_h + c]  ->  recv_a[r][c] (int32, as the exchange leaves them) and
//   xo_c[c][r] (bytes along h's column chains, as Update's transposition leaves them);
// job b, for its row chains: the labels of every strip's column chains on h's rows,
//   v = xcolT_g[c - c0_g][r0_h + r]  ->  recv_b[c][r] and xo_r[r][c].
// 64 x 64 tiles through LDS like transpose_labels_kernel; the workgroups of job b follow
// those of job a.
constexpr int kPullMaxStrips = 64;
struct PullArgs {
  const label_t *src_row[kPullMaxStrips];   // xrow of strip g: [nr_g][cols]
  const label_t *src_colT[kPullMaxStrips];  // xcolT of strip g: [nc_g][rows]
  int G, rows, cols;
  int r0, nr, c0, nc;                       // the pulling strip
  int *recv_a;     // [rows][nc]
  int *recv_b;     // [cols][nr]
  label_t *xo_c;   // [nc][rows]
  label_t *xo_r;   // [nr][cols]
  int tiles_a, tiles_a_minor, tiles_b_minor;  // workgroups of job a; tiles along the minor axis of each job
};

__global__ __launch_bounds__(256) void pull_labels_kernel(const PullArgs a) {
  __shared__ label_t s_t[64][64 + 4];
  const bool jb = (int)blockIdx.x >= a.tiles_a;
  const int blk = jb ? blockIdx.x - a.tiles_a : blockIdx.x;
  // job a: major = grid row (all of them), minor = my columns; job b: major = grid column, minor = my rows
  const int nmaj = jb ? a.cols : a.rows, nmin = jb ? a.nr : a.nc, off_min = jb ? a.r0 : a.c0;
  const int pitch = jb ? a.rows : a.cols;  // row pitch of a source array
  const int tiles_minor = jb ? a.tiles_b_minor : a.tiles_a_minor;
  const int m0 = (blk / tiles_minor) * 64, n0 = (blk % tiles_minor) * 64;
  int *__restrict__ recv = jb ? a.recv_b : a.recv_a;
  label_t *__restrict__ xo = jb ? a.xo_r : a.xo_c;
  label_t v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    const int mj = min(m0 + (i >> 6), nmaj - 1), mn = min(n0 + (i & 63), nmin - 1);
    int st, cnt;
    const int g = owner_of(mj, nmaj, a.G, &st, &cnt);
    const label_t *__restrict__ src = jb ? a.src_colT[g] : a.src_row[g];
    v[j] = src[(long long)(mj - st) * pitch + off_min + mn];
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    s_t[i >> 6][i & 63] = v[j];
    if (m0 + (i >> 6) < nmaj && n0 + (i & 63) < nmin) recv[(long long)(m0 + (i >> 6)) * nmin + n0 + (i & 63)] = (int)v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int i = threadIdx.x + 256 * j;
    const int hi = i >> 6, lo = i & 63;
    if (n0 + hi < nmin && m0 + lo < nmaj) xo[(long long)(n0 + hi) * nmaj + m0 + lo] = s_t[lo][hi];
  }
}

// Dense form of the strip exchange: the marginals of the chains' own
// solutions as float k-vectors (one-hot of the label; 1/k everywhere before
// the first solve), which is what the reference moves between ranks
// (Grid.cpp:1646-1656 sends rows of vertMuArr).
__global__ void labels_to_marginals_kernel(float *__restrict__ mu, const int *__restrict__ lab,
                                           long long n, int k, float inv_k) {
  const long long total = n * k;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
    mu[i] = chain_mu(lab[i / k], (int)(i % k), inv_k);
}

// The inverse: the label whose marginal is 1, or -1 for the uniform vector.
__global__ void marginals_to_labels_kernel(int *__restrict__ lab, const float *__restrict__ mu,
                                           long long n, int k) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float *v = mu + i * k;
    int x = -1;
    for (int l = k - 1; l >= 0; l--) x = (v[l] == 1.0f) ? l : x;
    lab[i] = x;
  }
}

}  // namespace admm
