// Calibration of the FETCH_SIZE counter on the chain sweep's read patterns.
//
// rocprofv3's FETCH_SIZE is derived from the L2's memory-side request counters
// ((BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32) / 1024) and on gfx950 tallies
// a 128-byte request as 64 bytes, so what it reports per byte really fetched depends on
// the width of the requests a kernel causes.  Every kernel below reads a buffer of known
// size exactly once in one of the sweep's patterns; run under
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calibration
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -- ./fetch_calibration
// the ratio known bytes / (FETCH_SIZE * 1024) is the factor for that pattern
// (tools/distill_profiles.py reads them from profiles/*_fetch_calibration.json).
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calibration tools/fetch_calibration.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

constexpr int R = 2048, C = 2048, K = 32;  // the headline grid: nodes of 32 labels

// wide coalesced stream, 16 B per lane (128-byte requests)
__global__ __launch_bounds__(256) void cal_wide_16B_per_lane(const float4 *__restrict__ a, float *__restrict__ sink, long long n4) {
  float acc = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}

// The sweep's walk: a wavefront holds two chains (lanes 0-31 and 32-63), lane b reads entry b
// of the chain's node at every step.  EL = bytes per entry (4: the unaries, a 128-byte node;
// 1: int8 multiplier counts, a 32-byte node).  COL = false: a row chain, consecutive steps are
// consecutive nodes; the two chains of a wavefront are two grid rows.  COL = true: a column
// chain, consecutive steps are a grid row apart; the two chains are adjacent columns, so their
// nodes are adjacent in memory (2 x 32 bytes = one 64-byte piece for int8 counts).
template <typename T, bool COL>
__device__ __forceinline__ void walk(const T *__restrict__ a, float *__restrict__ sink) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int chain = 2 * wave + (lane >> 5), b = lane & 31;
  if (chain >= (COL ? C : R)) return;
  const long long step = COL ? (long long)C * K : K;
  const T *p = a + (COL ? (long long)chain * K : (long long)chain * C * K) + b;
  const int Tn = COL ? R : C;
  float acc = 0.f;
  for (int t = 0; t < Tn; t += 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[(long long)(t + u) * step];
#pragma unroll
    for (int u = 0; u < 8; u++) acc += (float)v[u];
  }
  if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void cal_row_walk_128B_node(const float *a, float *sink) { walk<float, false>(a, sink); }
__global__ __launch_bounds__(256) void cal_col_walk_128B_node(const float *a, float *sink) { walk<float, true>(a, sink); }
__global__ __launch_bounds__(256) void cal_row_walk_32B_node(const signed char *a, float *sink) { walk<signed char, false>(a, sink); }
__global__ __launch_bounds__(256) void cal_col_walk_2x32B_nodes(const signed char *a, float *sink) { walk<signed char, true>(a, sink); }
// the same with int16 counts (64-byte nodes; two adjacent columns = 128 bytes)
__global__ __launch_bounds__(256) void cal_row_walk_64B_node(const short *a, float *sink) { walk<short, false>(a, sink); }
__global__ __launch_bounds__(256) void cal_col_walk_2x64B_nodes(const short *a, float *sink) { walk<short, true>(a, sink); }

// byte labels: every 64 steps a wavefront reads the 64 labels of the next chunk of each of its
// two chains, 32 bytes per chain and load instruction, chunk after chunk along the chain
__global__ __launch_bounds__(256) void cal_label_chunks_32B(const signed char *__restrict__ x, float *__restrict__ sink) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int chain = 2 * wave + (lane >> 5);
  if (chain >= R) return;
  float acc = 0.f;
  for (int t = 0; t < C; t += 64) {
    const signed char a = x[(long long)chain * C + t + (lane & 31)], b = x[(long long)chain * C + t + 32 + (lane & 31)];
    acc += (float)a + (float)b;
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const long long n = (long long)R * C * K;
  float *th = nullptr, *sink = nullptr;
  signed char *c8 = nullptr, *lab = nullptr;
  short *c16 = nullptr;
  CK(hipMalloc(&th, n * sizeof(float)));
  CK(hipMalloc(&c8, n));
  CK(hipMalloc(&c16, n * sizeof(short)));
  CK(hipMalloc(&lab, (long long)R * C));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(th, 0, n * sizeof(float)));
  CK(hipMemset(c8, 0, n));
  CK(hipMemset(c16, 0, n * sizeof(short)));
  CK(hipMemset(lab, 0, (long long)R * C));
  CK(hipDeviceSynchronize());
  const int walk_blocks = (R / 2 * 64 + 255) / 256;
  // every kernel twice: the second run's numbers are the ones to read (the first also pays for
  // whatever the memset left in the caches)
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(cal_wide_16B_per_lane, dim3(2048), dim3(256), 0, 0, (const float4 *)th, sink, n / 4);
    hipLaunchKernelGGL(cal_row_walk_128B_node, dim3(walk_blocks), dim3(256), 0, 0, th, sink);
    hipLaunchKernelGGL(cal_col_walk_128B_node, dim3(walk_blocks), dim3(256), 0, 0, th, sink);
    hipLaunchKernelGGL(cal_row_walk_32B_node, dim3(walk_blocks), dim3(256), 0, 0, c8, sink);
    hipLaunchKernelGGL(cal_col_walk_2x32B_nodes, dim3(walk_blocks), dim3(256), 0, 0, c8, sink);
    hipLaunchKernelGGL(cal_row_walk_64B_node, dim3(walk_blocks), dim3(256), 0, 0, c16, sink);
    hipLaunchKernelGGL(cal_col_walk_2x64B_nodes, dim3(walk_blocks), dim3(256), 0, 0, c16, sink);
    hipLaunchKernelGGL(cal_label_chunks_32B, dim3(walk_blocks), dim3(256), 0, 0, lab, sink);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
  }
  printf("{\"cal_wide_16B_per_lane\": %lld, \"cal_row_walk_128B_node\": %lld, \"cal_col_walk_128B_node\": %lld, "
         "\"cal_row_walk_32B_node\": %lld, \"cal_col_walk_2x32B_nodes\": %lld, \"cal_row_walk_64B_node\": %lld, "
         "\"cal_col_walk_2x64B_nodes\": %lld, \"cal_label_chunks_32B\": %lld}\n",
         n * 4, n * 4, n * 4, n, n, n * 2, n * 2, (long long)R * C);
  return 0;
}
