// Which SIMD the wavefronts of a 256-thread workgroup land on (HW_ID bits 5:4), for a launch
// of one workgroup per compute unit: the wavefronts 0..3 of every workgroup, tallied.
//   hipcc --offload-arch=gfx950 -O2 -o tools/gpu/simd_map tools/gpu/simd_map.hip && tools/gpu/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  // keep the workgroup resident for a while so that all workgroups coexist
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 200000) {}
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
}

int main() {
  const int nb = 256;
  unsigned *d;
  if (hipMalloc((void **)&d, nb * 4 * sizeof(unsigned)) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d);
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  int tally[4][4] = {};
  int same01 = 0, distinct = 0;
  for (int b = 0; b < nb; b++) {
    int simd[4];
    for (int w = 0; w < 4; w++) { simd[w] = (h[b * 4 + w] >> 4) & 3; tally[w][simd[w]]++; }
    same01 += simd[0] == simd[1];
    distinct += (simd[0] != simd[1] && simd[0] != simd[2] && simd[0] != simd[3] && simd[1] != simd[2] && simd[1] != simd[3] && simd[2] != simd[3]);
  }
  for (int w = 0; w < 4; w++) printf("wavefront %d of its workgroup: SIMD 0..3 = %d %d %d %d\n", w, tally[w][0], tally[w][1], tally[w][2], tally[w][3]);
  printf("workgroups whose wavefronts 0 and 1 share a SIMD: %d of %d; with four distinct SIMDs: %d\n", same01, nb, distinct);
  for (int b = 0; b < 4; b++) printf("workgroup %d: hw_id %08x %08x %08x %08x\n", b, h[b * 4], h[b * 4 + 1], h[b * 4 + 2], h[b * 4 + 3]);
  hipFree(d);
  return 0;
}
