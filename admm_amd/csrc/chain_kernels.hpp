// Chain (min-sum / Viterbi) solver kernels for gfx950.
//
// Trees are the row chains and the column chains of the grid.  One wavefront
// sweeps one chain with the label dimension across lanes (several short-label
// chains share a wavefront when k is small).  The ADMM shell around the sweep
// keeps the reference's formulas and operation order:
//   dual update      lambda += penalty * (mu - m)                  Tree.cpp:70
//   potential adjust t = rho*(theta + lambda) - penalty*(m - mu)   Tree.cpp:85-86
//   consensus        m = (mu_row + mu_col) / 2                     Grid.cpp:176-180
// The sweep itself (forward min-plus recursion with lowest-index
// back-pointers, backtrack from the lowest argmin) has no counterpart in the
// reference, whose only subproblem solver is the per-edge one in Tree.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace admm {

// One set of chains that share a stride pattern (all row chains, or all
// column chains).  Node t of chain c is
//   (c / cpg) * gstride + (c % cpg) * cstride + t * sstride,
// the k floats of a node are contiguous, and the labels of a chain are
// contiguous in x (chain-major).
struct ChainSeg {
  const float *theta;  // per-copy share of the unaries
  float *lam;          // Lagrange multipliers of these chains
  const float *m;      // consensus
  int *x;              // [nchains][T] labels of these chains (-1: uniform mu)
  uint8_t *bp;         // [nchains][T][KP] back-pointer scratch
  float *chain_min;    // [nchains] (dual evaluation only)
  int nchains;
  int T;
  int cpg;
  long long gstride, cstride, sstride;
};

struct ChainArgs {
  ChainSeg seg[2];
  int nseg;
  int wave_begin[3];  // waves [wave_begin[i], wave_begin[i+1]) work on seg[i]
  int k;
  float penalty, rho, inv_k;
  const float *Wt;     // k*k, rho * (W / rho)
};

constexpr int kChainBlock = 256;  // 4 independent wavefronts per workgroup
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

// Cross-lane exchange between the NS lanes that share a target label.
// Returns the partner's value for lane ^ d.
template <int D>
__device__ __forceinline__ float xor_lane_f(float v) { return __shfl_xor(v, D); }
template <int D>
__device__ __forceinline__ int xor_lane_i(int v) { return __shfl_xor(v, D); }

// KP   : labels padded to a power of two (lanes b >= k idle with M = +inf)
// NS   : lanes that share one target label; lane s of them scans the source
//        labels [s*KP/NS, (s+1)*KP/NS) and holds the node operands of the
//        steps t with t % NS == s, so one 64-lane load covers NS steps
// GPB  : load groups (of NS steps) per unrolled block = prefetch distance
// MINW : wavefronts per SIMD the register allocation must leave room for
// DUAL : evaluate the chain minimum under theta + lambda only (no dual
//        update, no back-pointers, no labels)
//
// Back-pointers: byte (t, b) = lowest source label attaining the minimum for
// target b at step t.  They are packed four steps to a dword and written
// straight from registers, dword (t/4)*KP + b of the chain's scratch; the
// backtrack streams them back through LDS one 64-step chunk at a time.
template <int KP, int NS, int GPB, int MINW, bool DUAL>
__global__ __launch_bounds__(kChainBlock, MINW) void chain_sweep_kernel(const ChainArgs args) {
  constexpr int LPC = KP * NS;   // lanes per chain
  constexpr int CPW = 64 / LPC;  // chains per wavefront
  constexpr int A = KP / NS;     // source labels scanned per lane
  constexpr int BLK = GPB * NS;  // steps per unrolled block
  constexpr int WIN = 4 * NS;    // steps per back-pointer store (one dword per lane)
  constexpr int DW = 16 / NS;    // back-pointer dwords per lane per chunk
  static_assert(LPC <= 64 && A >= 1 && NS <= 8, "bad lane layout");
  static_assert(BLK % WIN == 0 && kChunk % BLK == 0, "blocks must tile windows and chunks");
  constexpr int WPB = kChainBlock / 64;

  __shared__ __attribute__((aligned(16))) float s_M[WPB][CPW * KP + 64];  // messages + a dump row
  __shared__ int s_x[WPB][CPW][kChunk];
  __shared__ __attribute__((aligned(16))) uint32_t s_bp[WPB][CPW][16 * KP];

  const int lane = threadIdx.x & 63;
  // the wavefront index is the same in all 64 lanes: keep it (and what
  // derives from it) in scalar registers
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * WPB + wib;
  if (wave >= args.wave_begin[args.nseg]) return;
  const int si = (args.nseg > 1 && wave >= args.wave_begin[1]) ? 1 : 0;
  const ChainSeg &sg = args.seg[si];

  const int cs = lane / LPC;
  const int within = lane % LPC;
  const int s = within / KP;
  const int b = within % KP;
  const int k = args.k;
  const int chain_raw = (wave - args.wave_begin[si]) * CPW + cs;
  const bool cvalid = chain_raw < sg.nchains;
  const int chain = cvalid ? chain_raw : sg.nchains - 1;
  const int T = sg.T;
  const bool bvalid = b < k;
  const bool lvalid = cvalid && bvalid;
  const long long first = (long long)(chain / sg.cpg) * sg.gstride + (long long)(chain % sg.cpg) * sg.cstride;
  const float INF = __builtin_inff();
  const float pen = args.penalty, rho = args.rho, inv_k = args.inv_k;

  // pairwise column of this lane's target label, kept in registers
  float Wr[A];
#pragma unroll
  for (int j = 0; j < A; j++) {
    const int a = s * A + j;
    Wr[j] = (a < k && bvalid) ? args.Wt[a * k + b] : INF;
  }

  // Addresses: a per-chain base pointer (wave-uniform when the wavefront holds
  // one chain) plus a 32-bit element offset.  Operand loads are never
  // predicated: steps past the chain's end and labels past k read the slack
  // that follows every operand array, and their values are never used.
  const unsigned nstride = (unsigned)(sg.sstride * (long long)k);
  const float *__restrict__ th_c = sg.theta + first * k;
  const float *__restrict__ m_c = sg.m + first * k;
  float *lam_c = sg.lam + first * k;
  const unsigned lane_off = (unsigned)s * nstride + (unsigned)b;  // own step of group 0
  int *xchain = sg.x + (long long)chain * T;
  const int Tpad = (T + kChunk - 1) / kChunk * kChunk;
  uint32_t *bpchain = DUAL ? nullptr : reinterpret_cast<uint32_t *>(sg.bp) + (long long)chain * (Tpad / 4) * KP;

  float *Mrow = &s_M[wib][cs * KP];
  // where this lane's M_new goes at sub-step u: the real slot when the lane
  // holds that step's operands, a private dump slot otherwise
  int m_addr[NS];
#pragma unroll
  for (int u = 0; u < NS; u++) m_addr[u] = (s == u) ? (cs * KP + b) : (CPW * KP + lane);

  float pth[GPB], plam[GPB], pm[GPB];
#pragma unroll
  for (int j = 0; j < GPB; j++) {
    const unsigned off = lane_off + (unsigned)(j * NS) * nstride;
    pth[j] = th_c[off];
    plam[j] = lam_c[off];
    pm[j] = DUAL ? 0.0f : m_c[off];
  }
  int xnext = -1;  // previous labels of the next chunk, one 64-step chunk per chain
  if (!DUAL) {
    // lanes of a chain load chunk 0 into LDS now and keep chunk 1 in flight
    for (int i = within; i < kChunk; i += LPC) s_x[wib][cs][i] = (i < T) ? xchain[i] : -1;
  }

  uint32_t acc = 0, keep = 0;

  // One unrolled block of BLK steps starting at step tb (tt0 = tb - chunk start).
  auto block = [&](const int tb, const int tt0, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int j = 0; j < GPB; j++) {
      const int tg = tb + j * NS;
      // past the chain's end only the back-pointer bookkeeping goes on, so
      // that a partly filled window is still stored
      const bool glive = FULL || tg < T;
      const float th_in = pth[j], lam_in = plam[j], m_in = pm[j];
      const size_t goff = (size_t)(unsigned)tg * nstride;  // uniform
      {  // operands of the group one block ahead
        const size_t poff = goff + (size_t)BLK * nstride;
        pth[j] = (th_c + poff)[lane_off];
        plam[j] = (lam_c + poff)[lane_off];
        if (!DUAL) pm[j] = (m_c + poff)[lane_off];
      }
      float th_own = INF;
      if (DUAL) {
        th_own = th_in + lam_in;
      } else if (glive) {
        const int xo = s_x[wib][cs][tt0 + j * NS + s];
        const float mu = chain_mu(xo, b, inv_k);
        float la = lam_in;
        la = la + pen * (mu - m_in);                          // Tree.cpp:70
        if (lvalid && (FULL || tg + s < T)) (lam_c + goff)[lane_off] = la;
        th_own = rho * (th_in + la) - pen * (m_in - mu);      // Tree.cpp:85-86
      }
      if (!bvalid) th_own = INF;
#pragma unroll
      for (int u = 0; u < NS; u++) {
        int arg = 0;
        if (FULL || tg + u < T) {
        float c[A];
        if constexpr (A % 4 == 0) {
          const float4 *mp = reinterpret_cast<const float4 *>(Mrow + s * A);
#pragma unroll
          for (int q = 0; q < A / 4; q++) {
            const float4 v = mp[q];
            c[4 * q + 0] = v.x + Wr[4 * q + 0];
            c[4 * q + 1] = v.y + Wr[4 * q + 1];
            c[4 * q + 2] = v.z + Wr[4 * q + 2];
            c[4 * q + 3] = v.w + Wr[4 * q + 3];
          }
        } else {
#pragma unroll
          for (int q = 0; q < A; q++) c[q] = Mrow[s * A + q] + Wr[q];
        }
        float best = c[0];
#pragma unroll
        for (int q = 1; q < A; q++) best = __builtin_fminf(best, c[q]);
        if (!DUAL) {
          int aj = A - 1;
#pragma unroll
          for (int q = A - 2; q >= 0; q--) aj = (c[q] == best) ? q : aj;
          arg = s * A + aj;
        }
        if constexpr (NS == 2 && KP == 32) {
          // lanes l and l^32: one swap gives both halves' values to every lane
          const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
          const float blo = __uint_as_float(rb[0]), bhi = __uint_as_float(rb[1]);
          if (!DUAL) {
            const auto ra = __builtin_amdgcn_permlane32_swap((unsigned)arg, (unsigned)arg, false, false);
            arg = (bhi < blo) ? (int)ra[1] : (int)ra[0];  // ties keep the lower half's (lower) label
          }
          best = __builtin_fminf(blo, bhi);
        } else {
#pragma unroll
          for (int d = KP; d < LPC; d <<= 1) {
            const float ob = __shfl_xor(best, d);
            if (!DUAL) {
              const int oa = __shfl_xor(arg, d);
              const bool take = (ob < best) || (ob == best && oa < arg);
              arg = take ? oa : arg;
            }
            best = __builtin_fminf(best, ob);
          }
        }
        if (j == 0 && u == 0) best = (tb == 0) ? 0.0f : best;  // M_0 = t_0
        const float Mnew = th_own + best;
        wave_lds_fence();
        s_M[wib][m_addr[u]] = Mnew;
        wave_lds_fence();
        }
        if (!DUAL) {
          acc = __builtin_amdgcn_alignbit((unsigned)arg, acc, 8);
          const int sb = j * NS + u;  // step within the block
          if ((sb & 3) == 3) {
            if (s == ((sb >> 2) % NS)) keep = acc;
            if (((sb >> 2) % NS) == NS - 1) {
              const int w0 = (tb + sb - (WIN - 1)) >> 2;  // first dword of the window
              if (lvalid) bpchain[(long long)(w0 + s) * KP + b] = keep;
            }
          }
        }
      }
    }
  };

  for (int t0 = 0; t0 < T; t0 += kChunk) {
    if (!DUAL) {
      // labels of chunk t0 are in LDS; fetch the next chunk's while this one runs
      if (within < kChunk && LPC >= kChunk) xnext = (t0 + kChunk + within < T) ? xchain[t0 + kChunk + within] : -1;
    }
    if (t0 + kChunk <= T) {
#pragma unroll 1
      for (int tt = 0; tt < kChunk; tt += BLK) block(t0 + tt, tt, std::true_type{});
    } else {
#pragma unroll 1
      for (int tt = 0; tt < kChunk && t0 + tt < T; tt += BLK) block(t0 + tt, tt, std::false_type{});  // tail
    }
    if (!DUAL && t0 + kChunk < T) {
      wave_lds_fence();
      if (LPC >= kChunk) {
        s_x[wib][cs][within] = xnext;
      } else {
        for (int i = within; i < kChunk; i += LPC) s_x[wib][cs][i] = (t0 + kChunk + i < T) ? xchain[t0 + kChunk + i] : -1;
      }
      wave_lds_fence();
    }
  }

  // lowest-index argmin of the last message, inside each chain's KP lanes
  wave_lds_fence();
  float bv = Mrow[b];
  int bi = b;
#pragma unroll
  for (int d = 1; d < KP; d <<= 1) {
    const float ov = __shfl_xor(bv, d);
    const int oi = __shfl_xor(bi, d);
    const bool take = (ov < bv) || (ov == bv && oi < bi);
    bv = take ? ov : bv;
    bi = take ? oi : bi;
  }
  if (DUAL) {
    if (cvalid && within == 0) sg.chain_min[chain] = bv;
    return;
  }

  // backtrack, chunk by chunk from the end; one lane per chain walks the LDS
  // copy while the chunk before it is already being fetched into registers
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // this wavefront's own back-pointer stores
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
  const int nchunks = Tpad / kChunk;
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
    if (within == 0) {
      const uint8_t *bytes = reinterpret_cast<const uint8_t *>(&s_bp[wib][cs][0]);
      for (int tt = n - 1; tt >= 0; tt--) {
        s_x[wib][cs][tt] = cur;
        if (t0 + tt > 0) cur = bytes[(((tt >> 2) * KP + cur) << 2) + (tt & 3)];
      }
    }
    cur = __shfl(cur, cs * LPC);
    wave_lds_fence();
    if (cvalid)
      for (int i = within; i < n; i += LPC) xchain[t0 + i] = s_x[wib][cs][i];
    wave_lds_fence();
  }
}

// Consensus of a strip of nr x nc nodes from the two label arrays:
//   m[r][c][l] = (mu(xa[r][c])[l] + mu(xbT[c][r])[l]) / 2      Grid.cpp:163-181
// xa is row-major (labels of the row chains), xbT is column-major (labels of
// the column chains, chain-major as the sweep writes them).  A workgroup
// handles a 32 x 32 tile; the transposed read goes through LDS.
__global__ __launch_bounds__(256) void chain_consensus_kernel(float *__restrict__ m,
                                                              const int *__restrict__ xa,
                                                              const int *__restrict__ xbT, int nr,
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// Objective partials of a strip of nr x nc nodes, one wavefront per row.
// mode bit 0 (row part, run on the strip that owns whole rows):
//   primal += 2 * theta_share[x] + W[x][x_right]      (x = decoded label)
//   linear += theta_share[x_row] + W[x_row][x_row_right]
// mode bit 1 (column part, run on the strip that owns whole columns):
//   primal += W[x][x_down]
//   linear += theta_share[x_col] + W[x_col][x_col_down]
// linear is the chain analogue of Grid.cpp:1665-1680 (every tree's own
// solution under its share of the potentials).  out[2*row] = primal partial,
// out[2*row+1] = linear partial; the host adds the partials in binary64.
__global__ __launch_bounds__(256) void chain_objective_kernel(
    double *__restrict__ out, const float *__restrict__ theta, const float *__restrict__ Wsplit,
    const int *__restrict__ xa, const int *__restrict__ xbT, int nr, int nc, int k,
    int rows_total, int mode) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows_total) return;
  const int g = row / nr, r = row % nr;
  const long long abase = (long long)g * nr * nc + (long long)r * nc;  // xa: [g][nr][nc]
  const long long bbase = (long long)g * nr * nc;                      // xbT: [g][nc][nr]
  double p = 0.0, lin = 0.0;
  for (int c = lane; c < nc; c += 64) {
    const int xr = xa[abase + c], xc = xbT[bbase + (long long)c * nr + r];
    const int x = chain_decode(xr, xc);
    const int xr0 = xr < 0 ? 0 : xr, xc0 = xc < 0 ? 0 : xc;
    const float *th = theta + (abase + c) * k;
    if (mode & 1) {
      p += (double)th[x];
      p += (double)th[x];
      lin += (double)th[xr0];
      if (c + 1 < nc) {
        const int xr1 = xa[abase + c + 1], xc1 = xbT[bbase + (long long)(c + 1) * nr + r];
        p += (double)Wsplit[x * k + chain_decode(xr1, xc1)];
        lin += (double)Wsplit[xr0 * k + (xr1 < 0 ? 0 : xr1)];
      }
    }
    if (mode & 2) {
      lin += (double)th[xc0];
      if (r + 1 < nr) {
        const int xr2 = xa[abase + nc + c], xc2 = xbT[bbase + (long long)c * nr + r + 1];
        p += (double)Wsplit[x * k + chain_decode(xr2, xc2)];
        lin += (double)Wsplit[xc0 * k + (xc2 < 0 ? 0 : xc2)];
      }
    }
  }
  p = wave_sum(p);
  lin = wave_sum(lin);
  if (lane == 0) {
    out[2 * row] = p;
    out[2 * row + 1] = lin;
  }
}

// Strided gather / scatter of label blocks for the strip exchange.
// dst[i][j] = src[(r0+i)*ld + c0 + j], i < nr, j < nc (pack), or the inverse.
__global__ void label_block_copy_kernel(int *__restrict__ dst, long long dst_ld,
                                        const int *__restrict__ src, long long src_ld, int nr,
                                        int nc) {
  const long long total = (long long)nr * nc;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / nc, c = i % nc;
    dst[r * dst_ld + c] = src[r * src_ld + c];
  }
}

}  // namespace admm
