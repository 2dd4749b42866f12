// fetch_calibrate.hip — known-byte-count reads/writes in the access widths our kernels use, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM: FETCH_SIZE halves wide
// coalesced reads; "other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern").   hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o tools/fetch_calibrate
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/fetch_calibrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void read_b8(const uint8_t *p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 0xdeadbeef) *out = s;
}
__global__ void read_b32(const uint32_t *p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 0xdeadbeef) *out = s;
}
__global__ void read_b64(const uint2 *p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint2 v = p[i]; s += v.x + v.y; }
  if (s == 0xdeadbeef) *out = s;
}
__global__ void read_b128(const uint4 *p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 0xdeadbeef) *out = s;
}
// 16-byte chunks at a 2560-byte row pitch, 2 chunks per row, 23 rows: the convolve window pattern
__global__ void read_window(const uint8_t *p, size_t pitch, size_t rows, unsigned *out) {
  const size_t tile = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
  const int l = threadIdx.x % 64;
  unsigned s = 0;
  if (l < 46) {
    const size_t tx = tile % (pitch / 16 - 2), ty = (tile / (pitch / 16 - 2)) * 16;
    if (ty + 23 <= rows) {
      uint4 v;
      __builtin_memcpy(&v, p + (ty + l / 2) * pitch + tx * 16 + (l & 1) * 16 + 5, 16);
      s = v.x + v.y + v.z + v.w;
    }
  }
  if (s == 0xdeadbeef) *out = s;
}
__global__ void write_b8(uint8_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)i;
}
__global__ void write_b32(uint32_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void write_b128(uint4 *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, i, i, i);
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: well past the 256 MiB Infinity Cache
  uint8_t *buf;
  unsigned *out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 1, bytes));
  const int grid = 256 * 16, blk = 256;
  hipLaunchKernelGGL(read_b8, dim3(grid), dim3(blk), 0, 0, buf, bytes, out);
  hipLaunchKernelGGL(read_b32, dim3(grid), dim3(blk), 0, 0, (const uint32_t *)buf, bytes / 4, out);
  hipLaunchKernelGGL(read_b64, dim3(grid), dim3(blk), 0, 0, (const uint2 *)buf, bytes / 8, out);
  hipLaunchKernelGGL(read_b128, dim3(grid), dim3(blk), 0, 0, (const uint4 *)buf, bytes / 16, out);
  {
    const size_t pitch = 2560, rows = bytes / pitch;
    const size_t tiles = (pitch / 16 - 2) * (rows / 16);
    hipLaunchKernelGGL(read_window, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf, pitch, rows, out);
    printf("read_window: %zu tiles, %zu bytes requested, %zu distinct bytes\n", tiles, tiles * 46 * 16, bytes);
  }
  hipLaunchKernelGGL(write_b8, dim3(grid), dim3(blk), 0, 0, buf, bytes);
  hipLaunchKernelGGL(write_b32, dim3(grid), dim3(blk), 0, 0, (uint32_t *)buf, bytes / 4);
  hipLaunchKernelGGL(write_b128, dim3(grid), dim3(blk), 0, 0, (uint4 *)buf, bytes / 16);
  CK(hipDeviceSynchronize());
  printf("each read_/write_ kernel touches %zu bytes once\n", bytes);
  return 0;
}
