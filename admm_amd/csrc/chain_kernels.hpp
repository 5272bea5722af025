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

// KP : labels padded to a power of two (lanes b >= k idle with M = +inf)
// NS : lanes that share one target label, each scanning KP/NS source labels
// U  : steps whose operands are fetched together (software prefetch group)
// DUAL: evaluate the chain minimum under theta + lambda only (no dual update,
//       no back-pointers, no labels)
template <int KP, int NS, int U, bool DUAL>
__global__ __launch_bounds__(kChainBlock) void chain_sweep_kernel(const ChainArgs args) {
  constexpr int LPC = KP * NS;   // lanes per chain
  constexpr int CPW = 64 / LPC;  // chains per wavefront
  constexpr int A = KP / NS;     // source labels scanned per lane
  static_assert(LPC <= 64 && A >= 1, "bad lane layout");
  static_assert(kChunk % (2 * U) == 0, "prefetch groups must tile a chunk");
  constexpr int WPB = kChainBlock / 64;

  __shared__ float s_M[WPB][CPW][KP];
  __shared__ __attribute__((aligned(16))) uint8_t s_bp[WPB][CPW][kChunk][KP];
  __shared__ int s_x[WPB][CPW][kChunk];

  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
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
  const bool writer = cvalid && bvalid && s == 0;
  const long long first = (long long)(chain / sg.cpg) * sg.gstride + (long long)(chain % sg.cpg) * sg.cstride;
  const float INF = __builtin_inff();

  // pairwise column of this lane's target label, kept in registers
  float Wr[A];
#pragma unroll
  for (int j = 0; j < A; j++) {
    const int a = s * A + j;
    Wr[j] = (a < k && bvalid) ? args.Wt[a * k + b] : INF;
  }

  const float *th_base = sg.theta + first * k + b;
  const float *m_base = sg.m + first * k + b;
  float *lam_base = sg.lam + first * k + b;
  const long long nstride = sg.sstride * (long long)k;
  int *xchain = sg.x + (long long)chain * T;
  const int Tpad = (T + kChunk - 1) / kChunk * kChunk;  // whole chunks, so 16-byte flushes never cross chains
  uint8_t *bpchain = DUAL ? nullptr : sg.bp + (long long)chain * Tpad * KP;

  float Mcur = INF;  // M_t[b] of the step just finished (same in all NS lanes)

  float pth[2][U], plam[2][U], pm[2][U];
  auto fetch = [&](int buf, int tbase) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int t = tbase + u;
      const bool ok = bvalid && t < T;
      const long long off = (long long)t * nstride;
      pth[buf][u] = ok ? th_base[off] : 0.0f;
      plam[buf][u] = ok ? lam_base[off] : 0.0f;
      if (!DUAL) pm[buf][u] = ok ? m_base[off] : 0.0f;
    }
  };

  auto step = [&](int t, int tt, float th_in, float lam_in, float m_in) {
    float th;
    if (DUAL) {
      th = th_in + lam_in;
    } else {
      const int xo = s_x[wib][cs][tt];
      const float mu = chain_mu(xo, b, args.inv_k);
      float la = lam_in;
      la = la + args.penalty * (mu - m_in);
      if (writer) lam_base[(long long)t * nstride] = la;
      th = args.rho * (th_in + la) - args.penalty * (m_in - mu);
    }
    if (!bvalid) th = INF;
    float Mnew;
    int arg = 0;
    if (t == 0) {
      Mnew = th;
    } else {
      float c[A];
#pragma unroll
      for (int j4 = 0; j4 < A; j4++) c[j4] = s_M[wib][cs][s * A + j4] + Wr[j4];
      float best = c[0];
#pragma unroll
      for (int j = 1; j < A; j++) best = __builtin_fminf(best, c[j]);
      if (!DUAL) {
        int aj = A - 1;
#pragma unroll
        for (int j = A - 2; j >= 0; j--) aj = (c[j] == best) ? j : aj;
        arg = s * A + aj;
      }
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
      Mnew = th + best;
    }
    Mcur = Mnew;
    wave_lds_fence();
    if (s == 0) {
      s_M[wib][cs][b] = Mnew;
      if (!DUAL) s_bp[wib][cs][tt][b] = (uint8_t)arg;
    }
    wave_lds_fence();
  };

  fetch(0, 0);
  for (int t0 = 0; t0 < T; t0 += kChunk) {
    const int n = min(kChunk, T - t0);
    if (!DUAL) {
      // previous labels of this chunk -> LDS (one coalesced load per chain)
      wave_lds_fence();
      for (int i = within; i < n; i += LPC) s_x[wib][cs][i] = xchain[t0 + i];
      wave_lds_fence();
    }
    for (int g = 0; g < n; g += 2 * U) {
      // operands of the next group are requested before this group is used
      fetch(1, t0 + g + U);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (g + u < n) step(t0 + g + u, g + u, pth[0][u], plam[0][u], pm[0][u]);
      fetch(0, t0 + g + 2 * U);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (g + U + u < n) step(t0 + g + U + u, g + U + u, pth[1][u], plam[1][u], pm[1][u]);
    }
    if (!DUAL) {
      // flush the chunk's back-pointers, 16 bytes per lane
      wave_lds_fence();
      const int bytes = (n * KP + 15) & ~15;
      const uint8_t *src = &s_bp[wib][cs][0][0];
      uint8_t *dst = bpchain + (long long)t0 * KP;
      if (cvalid)
        for (int o = within * 16; o < bytes; o += LPC * 16)
          *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
      wave_lds_fence();
    }
  }

  // lowest-index argmin of the last message, inside each chain's KP lanes
  float bv = Mcur;
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
  int cur = bi;
  const int nchunks = (T + kChunk - 1) / kChunk;
  for (int ci = nchunks - 1; ci >= 0; ci--) {
    const int t0 = ci * kChunk;
    const int n = min(kChunk, T - t0);
    if (ci != nchunks - 1) {  // the last chunk is still in LDS
      wave_lds_fence();
      const int bytes = (n * KP + 15) & ~15;
      uint8_t *dst = &s_bp[wib][cs][0][0];
      const uint8_t *src = bpchain + (long long)t0 * KP;
      for (int o = within * 16; o < bytes; o += LPC * 16)
        *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(src + o);
      wave_lds_fence();
    }
    if (within == 0) {
      for (int tt = n - 1; tt >= 0; tt--) {
        s_x[wib][cs][tt] = cur;
        if (t0 + tt > 0) cur = s_bp[wib][cs][tt][cur];
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
