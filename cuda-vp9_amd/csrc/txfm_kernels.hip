// txfm_kernels.hip — batched inverse transform + add (SURVEY §8 a1–a3).
//
// Mapping: N lanes per N×N block (lane = row in the row pass, = column in the column pass),
// 256/N blocks per 256-thread workgroup.  Coefficients are read from HBM once, coalesced
// (lane t of a block reads element [i][t]), staged in LDS with an odd row pitch (N+1 dwords:
// row reads and column reads are both conflict-free), transformed in registers, and the
// rounded residual is added to the destination rows (N consecutive samples per block-row).
// Algorithmic bytes per block: N*N*4 (coefficients) + 2*N*N*bps (dest read+write) + 16.
#include "txfm_device.h"
#include "vp9hip_internal.h"

namespace {

template <typename Pix>
__device__ __forceinline__ int clip_pix(int v, int maxv) {
  return v < 0 ? 0 : (v > maxv ? maxv : v);
}

template <int N, typename Pix, bool HBD>
__global__ __launch_bounds__(256) void idct_add_kernel(const vp9hip_txb *__restrict__ blocks, int n_blocks,
                                                       const int32_t *__restrict__ coeffs, FrameDev f) {
  constexpr int BPW = 256 / N;          // blocks per workgroup
  constexpr int PITCH = N + 1;          // LDS row pitch in dwords
  __shared__ int lds[BPW * N * PITCH];
  const int t = threadIdx.x % N;        // lane inside the block
  const int lb = threadIdx.x / N;       // local block
  const int gb = blockIdx.x * BPW + lb;
  const bool active = gb < n_blocks;
  int *tile = lds + lb * N * PITCH;

  vp9hip_txb blk;
  if (active) blk = blocks[gb];
  const int tx_type = active ? (blk.tx_type & 3) : 0;
  const bool lossless = active && (blk.tx_type & 0x80);
  const int eob = active ? blk.eob : 0;
  // vp9_idct.c:119-204: DCT_DCT (and every 32x32) takes the DC-only shortcut
  const bool dc_path = active && !lossless && (tx_type == 0 || N == 32) && (N == 4 ? eob <= 1 : eob == 1);
  const bool wht_dc = lossless && eob <= 1;
  const int32_t *src = coeffs + (active ? blk.coeff_off : 0);

  int v[N];
  if (active && !dc_path && !wht_dc) {
#pragma unroll
    for (int i = 0; i < N; ++i) tile[i * PITCH + t] = src[i * N + t];
  }
  __syncthreads();
  if (active && !dc_path && !wht_dc) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tile[t * PITCH + k];
    if (lossless) {
      if constexpr (N == 4) txfm::iwht4(v, true);
    } else if (N < 32 && (tx_type & 2)) {  // DCT_ADST / ADST_ADST: adst on rows
      if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
    } else {
      txfm::idct1d<N, HBD>(v);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) tile[t * PITCH + k] = v[k];
  }
  __syncthreads();
  if (!active) return;

  constexpr int shift = N == 4 ? 4 : (N == 8 ? 5 : 6);
  if (dc_path) {
    const int a1 = txfm::dc_only<N, HBD>(src[0]);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = a1;
  } else if (wht_dc) {  // vpx_iwht4x4_1_add_c (inv_txfm.c:71-94)
    if constexpr (N == 4) {
      txfm::i64 a1 = src[0] >> 2, e1 = a1 >> 1;
      a1 -= e1;
      const int ip = t == 0 ? (int)a1 : (int)e1;
      const int e = ip >> 1;
      v[0] = ip - e;
      v[1] = v[2] = v[3] = e;
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tile[k * PITCH + t];
    if (lossless) {
      if constexpr (N == 4) txfm::iwht4(v, false);
    } else {
      if (N < 32 && (tx_type & 1)) {  // ADST_DCT / ADST_ADST: adst on columns
        if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
      } else {
        txfm::idct1d<N, HBD>(v);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = txfm::add32(v[k], 1 << (shift - 1)) >> shift;
    }
  }

  const int pl = blk.plane;
  Pix *dst = (Pix *)f.plane[pl];
  const int stride = f.stride[pl];
  const int maxv = (1 << f.bit_depth) - 1;
  const int x = blk.x + t;
  if (x >= f.awidth[pl]) return;
  const int rows = min(N, f.aheight[pl] - (int)blk.y);
  Pix *p = dst + (size_t)blk.y * stride + x;
  // all destination loads are issued before the first store (the compiler cannot prove that
  // row k+1 does not alias row k and would otherwise serialise N load->store round trips)
  int d[N];
#pragma unroll
  for (int k = 0; k < N; ++k) d[k] = (k < rows) ? (int)p[(size_t)k * stride] : 0;
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k < rows) p[(size_t)k * stride] = (Pix)clip_pix<Pix>(txfm::add32(d[k], v[k]), maxv);
}

template <int N>
int launch_size(vp9hip_ctx *ctx, const vp9hip_txb *blocks, int n, const int32_t *coeffs, const vp9hip_frame *fr) {
  if (n <= 0) return VP9HIP_OK;
  constexpr int BPW = 256 / N;
  const int grid = (n + BPW - 1) / BPW;
  const FrameDev f = to_dev(fr);
  if (fr->hbd)
    hipLaunchKernelGGL((idct_add_kernel<N, uint16_t, true>), dim3(grid), dim3(256), 0, ctx->stream, blocks, n, coeffs, f);
  else
    hipLaunchKernelGGL((idct_add_kernel<N, uint8_t, false>), dim3(grid), dim3(256), 0, ctx->stream, blocks, n, coeffs, f);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}

}  // namespace

extern "C" int vp9hip_idct_add_batch(vp9hip_ctx *ctx, const vp9hip_txb *d_blocks, const int32_t size_count[4],
                                     const int32_t *d_coeffs, const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_blocks || !size_count || !d_coeffs || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_idct_add_batch: bad argument");
  for (int i = 0; i < 4; ++i)
    if (size_count[i] < 0) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_idct_add_batch: negative count");
  int rc;
  const vp9hip_txb *b = d_blocks;
  if ((rc = launch_size<4>(ctx, b, size_count[0], d_coeffs, frame))) return rc;
  b += size_count[0];
  if ((rc = launch_size<8>(ctx, b, size_count[1], d_coeffs, frame))) return rc;
  b += size_count[1];
  if ((rc = launch_size<16>(ctx, b, size_count[2], d_coeffs, frame))) return rc;
  b += size_count[2];
  if ((rc = launch_size<32>(ctx, b, size_count[3], d_coeffs, frame))) return rc;
  return VP9HIP_OK;
}
