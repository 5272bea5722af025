// What the DPP row shifts do on this part: which lane a lane reads, and what an out-of-row read gives.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out) {
  const int lane = threadIdx.x;
  float m = (float)lane, w = 100.0f, a = -1.0f, b = -1.0f, c = -1.0f;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(a) : "v"(m), "v"(w));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(b) : "v"(m), "v"(w));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(m), "v"(w));
  out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
}
int main() {
  float *d; hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *n[3] = {"row_shr:1 bound_ctrl:1", "row_shl:3 bound_ctrl:1", "row_shr:2 (no bound_ctrl, dst preset -1)"};
  for (int r = 0; r < 3; r++) { printf("%s:", n[r]); for (int i = 0; i < 20; i++) printf(" %g", h[64 * r + i]); printf("\n"); }
  return 0;
}
