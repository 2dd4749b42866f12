// Effective shader clock inside short and long kernels: delta s_memtime / delta s_memrealtime x 100 MHz.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/build/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ __launch_bounds__(256) void spin(unsigned *out, long long *t, int iters) {
  unsigned a = threadIdx.x, b = blockIdx.x + 3;
  const long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(b));
  }
  const long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = m1 - m0; t[1] = r1 - r0; }
}
int main() {
  unsigned *o; long long *t, h[2];
  (void)hipMalloc(&o, 2048 * 256 * 4); (void)hipMalloc(&t, 16);
  const int iters_list[] = { 20, 200, 2000, 20000 };
  for (int pass = 0; pass < 2; ++pass)
    for (int k = 0; k < 4; ++k) {
      if (pass == 1) usleep(20000);  // idle gap before a single launch
      const int reps = pass == 0 ? 20 : 1;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, 0, o, t, iters_list[k]);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
      printf("%s iters %6d: %8lld memtime ticks in %7.1f us realtime -> %.0f MHz; %.2f ticks per v_mad (2 waves/SIMD... 8 waves/CU x8 WG)\n",
             pass == 0 ? "20 back-to-back" : "after 20 ms idle", iters_list[k], h[0], h[1] / 100.0, h[0] / (h[1] / 100.0),
             (double)h[0] / (iters_list[k] * 64.0));
    }
  return 0;
}
