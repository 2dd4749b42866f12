// Issue cost of the integer VALU instructions the reconstruction kernels lean on (gfx950), in shader
// cycles per wave64 instruction on one SIMD, at 1, 2 and 4 waves per SIMD (s_memtime around a loop of
// 8 independent chains x 64 repeats).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/build/valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, long long *cyc, int iters) {
  unsigned a[8], b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x * 40503u + 7u;
  for (int i = 0; i < 8; ++i) a[i] = b * (i + 3) + c;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define CHAIN(i)                                                                                          \
  if (OP == 0) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                 \
  if (OP == 1) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
  if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
  if (OP == 3) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));                        \
  if (OP == 4) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                  \
  if (OP == 5) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                 \
  if (OP == 6) asm volatile("v_ashr_pk_u8_i32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));                       \
  if (OP == 7) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                \
  if (OP == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 \
  if (OP == 9) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
  if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b));                         \
  if (OP == 11) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                             \
  if (OP == 12) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&a[i & 6]) : "v"(b), "v"(c) : "vcc"); \
  if (OP == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
  if (OP == 14) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                     \
  if (OP == 15) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                             \
  if (OP == 16) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)); \
  if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                       \
  if (OP == 18) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                \
  if (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP8(CHAIN)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  unsigned x = 0;
  for (int i = 0; i < 8; ++i) x ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

typedef void (*kern_t)(unsigned *, long long *, int);
template <int OP> void run(const char *name, unsigned *d_out, long long *d_cyc) {
  const int iters = 64;
  printf("%-18s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {
    // wps waves per SIMD on every CU: 256 CUs, one workgroup of 4*wps waves each (a workgroup's waves spread over the SIMDs)
    const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps, blocks = 256 * (64 * 4 * wps / threads);
    const int waves = blocks * threads / 64;
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(waves);
    hipMemcpy(h.data(), d_cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[waves / 2] / (iters * 64.0);  // cycles one wave spends per instruction
    printf("  %d w/SIMD: %5.2f cyc/instr/wave = %5.2f cyc/instr/SIMD", wps, per_wave, per_wave / wps);
  }
  printf("\n");
}

int main() {
  unsigned *d_out; long long *d_cyc;
  hipMalloc(&d_out, 1024 * 1024 * 4); hipMalloc(&d_cyc, 65536 * 8);
  run<13>("v_add_u32", d_out, d_cyc);
  run<8>("v_xor_b32", d_out, d_cyc);
  run<0>("v_dot4_i32_i8", d_out, d_cyc);
  run<1>("v_dot4c_i32_i8", d_out, d_cyc);
  run<18>("v_dot4_u32_u8", d_out, d_cyc);
  run<7>("v_dot2_i32_i16", d_out, d_cyc);
  run<2>("v_perm_b32", d_out, d_cyc);
  run<3>("v_alignbyte_b32", d_out, d_cyc);
  run<4>("v_pk_mad_i16", d_out, d_cyc);
  run<11>("v_pk_add_u16", d_out, d_cyc);
  run<15>("v_pk_max_i16", d_out, d_cyc);
  run<5>("v_mad_i32_i24", d_out, d_cyc);
  run<19>("v_mul_lo_u32", d_out, d_cyc);
  run<12>("v_mad_u64_u32", d_out, d_cyc);
  run<6>("v_ashr_pk_u8_i32", d_out, d_cyc);
  run<9>("v_med3_i32", d_out, d_cyc);
  run<10>("v_lshl_or_b32", d_out, d_cyc);
  run<14>("v_sad_u8", d_out, d_cyc);
  run<16>("v_mov_b32 dpp", d_out, d_cyc);
  run<17>("v_cndmask_b32", d_out, d_cyc);
  return 0;
}
