// Memory-rate probe for the chain sweep's access patterns on one GPU: what rate do two
// streamed reads and one streamed write reach when they are issued the way the sweep
// issues them (one node = 128 B per chain and step), against a plain streaming kernel?
//   hipcc --offload-arch=gfx950 -O3 -o access_patterns tools/access_patterns.hip && ./access_patterns
// Prints one line per pattern: bytes moved, milliseconds, TB/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

constexpr int R = 2048, C = 2048, K = 32;  // nodes of 32 floats, row-major
constexpr long long N = (long long)R * C * K;

// plain streaming: lam[i] = th[i] + lam[i], 16 B per lane
__global__ __launch_bounds__(256) void stream_kernel(const float4 *__restrict__ th, float4 *__restrict__ lam, long long n4) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 a = th[i], b = lam[i];
    lam[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// streaming with a mix of NR reads per write from separate buffers (NR = 0: write only); NT:
// non-temporal loads and stores (the data is not touched again before it leaves the caches)
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NR, bool WR, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const f4v *__restrict__ a, const f4v *__restrict__ b,
                                                  const f4v *__restrict__ c, const f4v *__restrict__ d,
                                                  f4v *__restrict__ out, float *__restrict__ sink, long long n4) {
  float acc = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f4v v = {1.f, 2.f, 3.f, 4.f};
    const f4v *src[4] = {a, b, c, d};
#pragma unroll
    for (int r = 0; r < NR; r++) {
      f4v x;
      if (NT) x = __builtin_nontemporal_load(src[r] + i);
      else x = src[r][i];
      v += x;
    }
    if (WR) {
      if (NT) __builtin_nontemporal_store(v, out + i);
      else out[i] = v;
    } else {
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (!WR && acc == 123.456f) sink[0] = acc;
}

// The sweep's pattern.  A wavefront walks two chains, half a chain each way as the sweep
// does (left half forwards, right half backwards), one node (32 lanes x 4 B) per chain and
// step, operands fetched D steps ahead.  COL = false: chains are rows (a step advances 128 B,
// the two chains lie 256 KiB apart); COL = true: chains are columns (a step advances a whole
// row, the two chains are adjacent).  WRITE = false leaves out the store.
template <bool COL, int D, bool WRITE>
__global__ __launch_bounds__(256) void walk_kernel(const float *__restrict__ th, float *__restrict__ lam, float *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rev = wave & 1;
  const int chain = (wave >> 1) * 2 + (lane >> 5);
  const int T = (COL ? R : C) / 2;
  const long long step = COL ? (long long)C * K : K;
  const long long base = (COL ? (long long)chain * K : (long long)chain * C * K) + (lane & 31);
  long long pos = rev ? base + (2LL * T - 1) * step : base;
  const long long dir = rev ? -step : step;
  float a[D], b[D];
#pragma unroll
  for (int j = 0; j < D; j++) {
    a[j] = th[pos + j * dir];
    b[j] = lam[pos + j * dir];
  }
  float acc = 0.f;
  for (int t = 0; t < T; t += D) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      const float v = a[j] + b[j];
      acc += v;
      if (WRITE) lam[pos + j * dir] = v;
      if (t + D < T) {
        a[j] = th[pos + (j + D) * dir];
        b[j] = lam[pos + (j + D) * dir];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += D * dir;
  }
  if (acc == 123.456f) sink[0] = acc;
}

// Rows again, 16 B per lane: a wavefront load covers four consecutive nodes of each of its
// two chains (512 B contiguous per chain), fetched D groups ahead.
template <int D, bool WRITE>
__global__ __launch_bounds__(256) void walk_rows_x4_kernel(const float4 *__restrict__ th, float4 *__restrict__ lam, float *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rev = wave & 1;
  const int chain = (wave >> 1) * 2 + (lane >> 5);
  const int G = C / 2 / 4;  // groups of four nodes per half chain
  const long long step = 32;  // float4 per group
  const long long base = (long long)chain * C * K / 4 + (lane & 31);
  long long pos = rev ? base + (2LL * G - 1) * step : base;
  const long long dir = rev ? -step : step;
  float4 a[D], b[D];
#pragma unroll
  for (int j = 0; j < D; j++) {
    a[j] = th[pos + j * dir];
    b[j] = lam[pos + j * dir];
  }
  float acc = 0.f;
  for (int t = 0; t < G; t += D) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      const float4 v = make_float4(a[j].x + b[j].x, a[j].y + b[j].y, a[j].z + b[j].z, a[j].w + b[j].w);
      acc += v.x + v.y + v.z + v.w;
      if (WRITE) lam[pos + j * dir] = v;
      if (t + D < G) {
        a[j] = th[pos + (j + D) * dir];
        b[j] = lam[pos + (j + D) * dir];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += D * dir;
  }
  if (acc == 123.456f) sink[0] = acc;
}

// Streaming with 4 B per lane: the same mixes as mix_kernel, every load and store a dword
// (a wavefront touches 256 contiguous bytes per instruction instead of 1 KiB).
template <int NR, bool WR>
__global__ __launch_bounds__(256) void mix4_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                   const float *__restrict__ c, const float *__restrict__ d,
                                                   float *__restrict__ out, float *__restrict__ sink, long long n) {
  float acc = 0.f;
  const float *src[4] = {a, b, c, d};
  // four independent elements per trip, so that as many bytes are in flight per lane as with 16-B accesses
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 4LL * gridDim.x * 256) {
    float v[4] = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long long j = i + (long long)u * gridDim.x * 256;
      if (j < n) {
#pragma unroll
        for (int r = 0; r < NR; r++) v[u] += src[r][j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long long j = i + (long long)u * gridDim.x * 256;
      if (j < n) {
        if (WR) out[j] = v[u];
        else acc += v[u];
      }
    }
  }
  if (!WR && acc == 123.456f) sink[0] = acc;
}

// The sweep's memory operations and nothing else: what the Potts sweep of the 2048 x 2048,
// k = 32 grid loads and stores, in its launch shape (1024 workgroups of four wavefronts, the
// first two the left and right half of two row chains, the other two of two column chains).
// Per step and chain: 128 B of theta (binary32) and 64 B of multiplier counts (int16) loaded D
// steps ahead; the row chains store 64 B of counts into a second buffer; every 64 steps a
// wavefront stores 1 KiB of back-pointer masks.  No arithmetic between them, so the rate it
// reaches is what this access pattern can get out of the memory system.
template <int D>
__global__ __launch_bounds__(256, 4) void sweep_skeleton_kernel(const float *__restrict__ th, const short *__restrict__ lam,
                                                                 short *__restrict__ lam_out, unsigned *__restrict__ bp,
                                                                 float *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const bool col = wib >= 2;
  const int rev = wib & 1;
  const int chain = blockIdx.x * 2 + (lane >> 5);
  const int T = (col ? R : C) / 2;
  const long long step = col ? (long long)C * K : K;
  const long long base = (col ? (long long)chain * K : (long long)chain * C * K) + (lane & 31);
  long long pos = rev ? base + (2LL * T - 1) * step : base;
  const long long dir = rev ? -step : step;
  float a[D];
  short b[D];
#pragma unroll
  for (int j = 0; j < D; j++) {
    a[j] = th[pos + j * dir];
    b[j] = lam[pos + j * dir];
  }
  float acc = 0.f;
  unsigned *bpw = bp + ((long long)(blockIdx.x * 4 + wib) * (T / 64)) * 256 + lane;
  for (int t = 0; t < T; t += D) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      const float v = a[j] + (float)b[j];
      acc += v;
      if (!col) lam_out[pos + j * dir] = (short)(b[j] + 1);
      if (t + D < T) {
        a[j] = th[pos + (j + D) * dir];
        b[j] = lam[pos + (j + D) * dir];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pos += D * dir;
    if (((t + D) & 63) == 0) {
      unsigned *o = bpw + ((t + D) / 64 - 1) * 256;
      o[0] = (unsigned)t; o[64] = (unsigned)t; o[128] = (unsigned)t; o[192] = (unsigned)t;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <typename F>
void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) launch();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-44s %8.3f GB  %7.3f ms  %6.2f TB/s\n", name, bytes / 1e9, ms, bytes / ms / 1e9);
  fflush(stdout);
}

int main() {
  float *th, *lam, *sink;
  CK(hipMalloc(&th, N * sizeof(float)));
  CK(hipMalloc(&lam, N * sizeof(float)));
  CK(hipMalloc(&sink, 256));
  CK(hipMemset(th, 0, N * sizeof(float)));
  CK(hipMemset(lam, 0, N * sizeof(float)));
  const double rd = 2.0 * N * 4, wr = 1.0 * N * 4;
  {
    // five separate 512 MiB buffers: up to four read, one written
    float *bufs[5];
    for (auto &p : bufs) {
      CK(hipMalloc(&p, N * sizeof(float)));
      CK(hipMemset(p, 0, N * sizeof(float)));
    }
    const double one = 1.0 * N * 4;
    const dim3 g(256 * 16), t(256);
#define MIX(NR, WR, NT, label)                                                                                     \
  timeit(label, one * (NR + (WR ? 1 : 0)), [&] {                                                                   \
    hipLaunchKernelGGL((mix_kernel<NR, WR, NT>), g, t, 0, 0, (const f4v *)bufs[0], (const f4v *)bufs[1],     \
                       (const f4v *)bufs[2], (const f4v *)bufs[3], (f4v *)bufs[4], sink, N / 4);          \
  })
    MIX(1, false, false, "stream: 1 read");
    MIX(4, false, false, "stream: 4 reads");
    MIX(4, false, true, "stream: 4 reads, non-temporal");
    MIX(0, true, false, "stream: 1 write");
    MIX(0, true, true, "stream: 1 write, non-temporal");
    MIX(1, true, false, "stream: 1 read + 1 write");
    MIX(1, true, true, "stream: 1 read + 1 write, non-temporal");
    MIX(2, true, false, "stream: 2 reads + 1 write");
    MIX(4, true, false, "stream: 4 reads + 1 write");
    MIX(4, true, true, "stream: 4 reads + 1 write, non-temporal");
#undef MIX
    // the same at the sweep's occupancy: 1024 workgroups of 256, four wavefronts per SIMD
    const dim3 gs(1024);
#define MIXS(NR, WR, label)                                                                                        \
  timeit(label, one * (NR + (WR ? 1 : 0)), [&] {                                                                   \
    hipLaunchKernelGGL((mix_kernel<NR, WR, false>), gs, t, 0, 0, (const f4v *)bufs[0], (const f4v *)bufs[1],   \
                       (const f4v *)bufs[2], (const f4v *)bufs[3], (f4v *)bufs[4], sink, N / 4);          \
  })
    MIXS(1, true, "1024 groups: copy (1 read + 1 write), 16 B/lane");
    MIXS(4, true, "1024 groups: 4 reads + 1 write, 16 B/lane");
    MIXS(4, false, "1024 groups: 4 reads, 16 B/lane");
#undef MIXS
#define MIX4(NR, WR, grid, label)                                                                                  \
  timeit(label, one * (NR + (WR ? 1 : 0)), [&] {                                                                   \
    hipLaunchKernelGGL((mix4_kernel<NR, WR>), dim3(grid), t, 0, 0, (const float *)bufs[0], (const float *)bufs[1], \
                       (const float *)bufs[2], (const float *)bufs[3], bufs[4], sink, N);                       \
  })
    MIX4(1, true, 4096, "copy (1 read + 1 write), 4 B/lane");
    MIX4(4, true, 4096, "4 reads + 1 write, 4 B/lane");
    MIX4(1, true, 1024, "1024 groups: copy, 4 B/lane");
    MIX4(4, true, 1024, "1024 groups: 4 reads + 1 write, 4 B/lane");
    MIX4(4, false, 1024, "1024 groups: 4 reads, 4 B/lane");
#undef MIX4
    for (auto &p : bufs) CK(hipFree(p));
  }
  const int rows_blocks = R / 2 * 2 / 4, cols_blocks = C / 2 * 2 / 4;  // two chains per wavefront, two halves, four wavefronts per block
  timeit("stream 2 reads + 1 write, 16 B/lane", rd + wr, [&] {
    hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)th, (float4 *)lam, N / 4);
  });
  timeit("rows, 4 B/lane, 8 ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_kernel<false, 8, true>), dim3(rows_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("rows, 4 B/lane, 8 ahead, read only", rd, [&] {
    hipLaunchKernelGGL((walk_kernel<false, 8, false>), dim3(rows_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("rows, 4 B/lane, 16 ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_kernel<false, 16, true>), dim3(rows_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("columns, 4 B/lane, 8 ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_kernel<true, 8, true>), dim3(cols_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("columns, 4 B/lane, 8 ahead, read only", rd, [&] {
    hipLaunchKernelGGL((walk_kernel<true, 8, false>), dim3(cols_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("columns, 4 B/lane, 16 ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_kernel<true, 16, true>), dim3(cols_blocks), dim3(256), 0, 0, th, lam, sink);
  });
  timeit("rows, 16 B/lane, 4 groups ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_rows_x4_kernel<4, true>), dim3(rows_blocks), dim3(256), 0, 0, (const float4 *)th, (float4 *)lam, sink);
  });
  timeit("rows, 16 B/lane, 8 groups ahead, read+write", rd + wr, [&] {
    hipLaunchKernelGGL((walk_rows_x4_kernel<8, true>), dim3(rows_blocks), dim3(256), 0, 0, (const float4 *)th, (float4 *)lam, sink);
  });
  timeit("rows, 16 B/lane, 4 groups ahead, read only", rd, [&] {
    hipLaunchKernelGGL((walk_rows_x4_kernel<4, false>), dim3(rows_blocks), dim3(256), 0, 0, (const float4 *)th, (float4 *)lam, sink);
  });
  {
    // the Potts sweep's own loads and stores (theta binary32, multiplier counts int16, masks)
    short *l16, *l16o;
    unsigned *bp;
    CK(hipMalloc(&l16, N * sizeof(short)));
    CK(hipMalloc(&l16o, N * sizeof(short)));
    CK(hipMalloc(&bp, 1024LL * 4 * (C / 2 / 64) * 256 * sizeof(unsigned)));
    CK(hipMemset(l16, 0, N * sizeof(short)));
    const double skel = 2.0 * N * 4 + 2.0 * N * 2 + 1.0 * N * 2 + 1024.0 * 4 * (C / 2 / 64) * 1024;
    timeit("sweep skeleton (theta x2, int16 lam x2 + 1 write), 8 ahead", skel, [&] {
      hipLaunchKernelGGL((sweep_skeleton_kernel<8>), dim3(1024), dim3(256), 0, 0, th, l16, l16o, bp, sink);
    });
    timeit("sweep skeleton, 16 ahead", skel, [&] {
      hipLaunchKernelGGL((sweep_skeleton_kernel<16>), dim3(1024), dim3(256), 0, 0, th, l16, l16o, bp, sink);
    });
    timeit("sweep skeleton, 4 ahead", skel, [&] {
      hipLaunchKernelGGL((sweep_skeleton_kernel<4>), dim3(1024), dim3(256), 0, 0, th, l16, l16o, bp, sink);
    });
  }
  CK(hipDeviceSynchronize());
  return 0;
}
