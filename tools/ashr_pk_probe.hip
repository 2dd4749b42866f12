// What v_ashr_pk_u8_i32 (new in gfx950) leaves in the half of the destination it does not write.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/build/ashr_pk_probe tools/ashr_pk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const int *in, unsigned *out) {
  const int a = in[0], b = in[1], c = in[2], e = in[3];
  unsigned lo = 0xDEADBEEFu, both = 0xDEADBEEFu, hi = 0xDEADBEEFu;
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 7" : "+v"(lo) : "v"(a), "v"(b));
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 7 op_sel:[0,0,0,1]" : "+v"(hi) : "v"(c), "v"(e));
  asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 7\n\tv_ashr_pk_u8_i32 %0, %3, %4, 7 op_sel:[0,0,0,1]"
               : "+v"(both) : "v"(a), "v"(b), "v"(c), "v"(e));
  out[0] = lo; out[1] = hi; out[2] = both;
}
int main() {
  int h[4] = { 100 << 7, -5 << 7, 300 << 7, (17 << 7) + 127 };  // -> 100, 0 (sat), 255 (sat), 17
  int *d; unsigned *o, r[3];
  hipMalloc(&d, 16); hipMalloc(&o, 12);
  hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
  printf("low-half write into 0xDEADBEEF:  %08x (expect ....0064 -> bytes 64,00)\n", r[0]);
  printf("high-half write into 0xDEADBEEF: %08x (expect 11ff....)\n", r[1]);
  printf("both:                            %08x (expect 11ff0064)\n", r[2]);
  return 0;
}
