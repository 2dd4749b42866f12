// clock_probe: what shader clock does the GPU run at under (a) one tiny wave, (b) a full chip?
// shader clock = delta(s_memtime) / delta(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1664525 + 1013904223;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = v; }
}
__global__ void empty() {}
int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    probe<<<1, 64>>>(d, 20000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("1 wave, short  : %.0f MHz (%llu cycles)\n", 100.0 * h[0] / h[1], h[0]);
  }
  // 1000 back-to-back tiny kernels (like the loop-filter diagonal launches)
  hipEventRecord(a); for (int i = 0; i < 1000; ++i) probe<<<20, 64>>>(d, 2000); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("1000 tiny launches: %.3f us each, clock in last %.0f MHz\n", ms, 100.0 * h[0] / h[1]);
  hipEventRecord(a); for (int i = 0; i < 1000; ++i) empty<<<20, 64>>>(); hipEventRecord(b); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b); printf("1000 empty launches: %.3f us each\n", ms);
  probe<<<4096, 256>>>(d, 2000000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("full chip, long: %.0f MHz\n", 100.0 * h[0] / h[1]);
  probe<<<1, 64>>>(d, 2000000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("1 wave, long   : %.0f MHz\n", 100.0 * h[0] / h[1]);
  return 0;
}
