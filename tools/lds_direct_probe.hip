// Probe (tools only): layout of global_load_lds_dwordx3 on gfx950 — lane l's 12 bytes land at LDS base + 16 * l
// (a 16-byte slot per lane, the fourth dword untouched; measured: a 12-byte stride does not match).  Prints the number
// of mismatches against plain loads; unaligned global addresses included.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const unsigned char *g, unsigned *out, int stride) {
  __shared__ unsigned buf[4][15 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned *my = buf[wave];
#pragma unroll
  for (int i = 0; i < 15; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)((lane * 7 + i + wave) % 97) * stride + lane * 4 + wave),
                                     (__attribute__((address_space(3))) void *)(my + i * 64 * 4), 12, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    unsigned d[3];
    __builtin_memcpy(d, my + i * 256 + lane * 4, 12);
    for (int q = 0; q < 3; ++q) out[((blockIdx.x * 256 + threadIdx.x) * 15 + i) * 3 + q] = d[q];
  }
}
int main() {
  const int stride = 1024, rows = 100;
  std::vector<unsigned char> h(stride * rows);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)(i * 2654435761u >> 13);
  unsigned char *d; unsigned *o;
  hipMalloc(&d, h.size()); hipMalloc(&o, 256 * 15 * 3 * 4);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, stride);
  std::vector<unsigned> r(256 * 15 * 3);
  hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) for (int i = 0; i < 15; ++i) for (int q = 0; q < 3; ++q) {
    const int lane = t & 63, wave = t >> 6;
    unsigned want; std::memcpy(&want, &h[(size_t)((lane * 7 + i + wave) % 97) * stride + lane * 4 + wave + 4 * q], 4);
    bad += want != r[(t * 15 + i) * 3 + q];
  }
  printf("global_load_lds_dwordx3: %d mismatches of %zu dwords (unaligned global addresses included)\n", bad, r.size());
  return bad != 0;
}
