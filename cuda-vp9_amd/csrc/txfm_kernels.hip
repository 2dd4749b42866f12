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

constexpr int TX_LDS_INTS = (256 / 32) * 32 * 33;  // the 32x32 class needs the most: 8 blocks x 32 rows x 33 dwords

// A block's N lanes lie inside one wavefront (N <= 32) and its LDS tile is touched by them only; a wave's LDS
// operations execute in issue order: the stages need the compiler pinned and lgkmcnt drained, not a workgroup barrier.
__device__ __forceinline__ void tile_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local"); }

template <int N, typename Pix, bool HBD>
__device__ __forceinline__ void idct_add_body(int *lds, int wg, const vp9hip_txb *__restrict__ blocks, int n_blocks,
                                              const txfm::Coefs &coeffs, const FrameDev &f) {
  constexpr int BPW = 256 / N;          // blocks per workgroup
  constexpr int PITCH = N + 1;          // LDS row pitch in dwords
  static_assert(BPW * N * PITCH <= TX_LDS_INTS, "LDS budget");
  const int t = threadIdx.x % N;        // lane inside the block
  const int lb = threadIdx.x / N;       // local block
  const int gb = wg * BPW + lb;
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
  const txfm::CoefAt src = txfm::at(coeffs, active ? blk.coeff_off : 0);

  int v[N];
  if (active && !dc_path && !wht_dc) {
    // Lane t reads ROW t of the coefficients straight into its registers — N consecutive coefficients, 16 bytes per
    // load (a slot starts at a multiple of N coefficients) — instead of N single loads that went through LDS to be
    // turned from columns into rows.  Rows past rd are zero and, in a compact slot, absent.
    const int rd = txfm::coeff_rows(eob, lossless ? 0 : tx_type, N);
    if (t < rd) {
      if (coeffs.c16) {
        short c[N];
        __builtin_memcpy(c, __builtin_assume_aligned((const short *)coeffs.p + blk.coeff_off + (unsigned)(t * N), 4), N * 2);
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = c[k];
      } else {
        __builtin_memcpy(v, __builtin_assume_aligned((const int *)coeffs.p + blk.coeff_off + (unsigned)(t * N), 4), N * 4);
      }
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = 0;
    }
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
  tile_sync();

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

  // ---- destination.  The column pass left lane t with COLUMN t of the residual; the frame is updated by ROWS: lane t
  // takes row t through LDS once more and adds it to N consecutive samples — one load and one store of N samples per
  // lane instead of N loads and N stores of one sample each (for the 4x4 blocks, two thirds of a frame's blocks, eight
  // single-byte memory instructions per lane became two dword ones).
  const bool uniform = dc_path;  // every sample of the block gets the same value: nothing to transpose
  if (active && !uniform) {
#pragma unroll
    for (int k = 0; k < N; ++k) tile[k * PITCH + t] = v[k];
  }
  tile_sync();
  if (!active) return;
  const int pl = blk.plane;
  const int stride = f.stride[pl];
  const int maxv = (1 << f.bit_depth) - 1;
  const int rows = min(N, f.aheight[pl] - (int)blk.y);
  const int cols = min(N, f.awidth[pl] - (int)blk.x);  // a multiple of 4 (block positions and awidth are)
  if (t >= rows || cols <= 0) return;
  int r[N];
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = uniform ? v[0] : tile[t * PITCH + j];
  Pix *p = (Pix *)f.plane[pl] + (size_t)(blk.y + t) * stride + blk.x;
  constexpr int SPD = 4 / (int)sizeof(Pix);  // samples per dword
  constexpr int ND = N / SPD;                // dwords per row
  unsigned dw[ND];
  const int vd = cols / SPD;                 // valid dwords of this row (cols is a multiple of 4)
  if (vd == ND) {
    __builtin_memcpy(dw, __builtin_assume_aligned(p, 4), ND * 4);
  } else {
#pragma unroll
    for (int q = 0; q < ND; ++q) dw[q] = q < vd ? ((const unsigned *)p)[q] : 0;
  }
#pragma unroll
  for (int q = 0; q < ND; ++q) {
    unsigned o = 0;
#pragma unroll
    for (int e = 0; e < SPD; ++e) {
      const int sh = e * 8 * (int)sizeof(Pix);
      const int smp = (int)((dw[q] >> sh) & (sizeof(Pix) == 1 ? 0xffu : 0xffffu));
      o |= (unsigned)clip_pix<Pix>(txfm::add32(smp, r[q * SPD + e]), maxv) << sh;
    }
    dw[q] = o;
  }
  if (vd == ND) {
    __builtin_memcpy(__builtin_assume_aligned(p, 4), dw, ND * 4);
  } else {
#pragma unroll
    for (int q = 0; q < ND; ++q)
      if (q < vd) ((unsigned *)p)[q] = dw[q];
  }
}

// All four transform sizes in one launch: workgroups [wg_start[k], wg_start[k+1]) serve size class k
// (the four launches were 7-10 us each for a 1440p frame, back to back and each with its own tail).
struct TxPlan {
  int wg_start[5];
  int blk_start[4];
  int count[4];
};

// XCD-aware order inside a size class (same idea as the convolve, inter_kernels.hip): workgroup b runs
// on XCD b % 8 and each XCD has its own L2; records are in decode (superblock-raster) order, so handing
// XCD x the x-th contiguous eighth of a class keeps blocks that share destination cache lines (4x4 and
// 8x8 blocks write 4- and 8-byte row pieces) in one L2 instead of fetching the lines once per XCD.
__device__ __forceinline__ int tx_xcd_order(int b, int s0, int s1) {
  const int x = b & 7;
  int prefix = 0, first_x = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int first = s0 + ((j - s0) & 7);  // first workgroup of [s0, s1) on XCD j
    const int cnt = first < s1 ? (s1 - first + 7) >> 3 : 0;
    if (j < x) prefix += cnt;
    if (j == x) first_x = first;
  }
  return prefix + ((b - first_x) >> 3);
}

template <typename Pix, bool HBD>
__global__ __launch_bounds__(256) void idct_add_all_kernel(const vp9hip_txb *__restrict__ blocks, TxPlan plan,
                                                           txfm::Coefs coeffs, FrameDev f) {
  __shared__ int lds[TX_LDS_INTS];
  const int b = blockIdx.x;
  if (b < plan.wg_start[1])
    idct_add_body<4, Pix, HBD>(lds, tx_xcd_order(b, plan.wg_start[0], plan.wg_start[1]), blocks + plan.blk_start[0], plan.count[0], coeffs, f);
  else if (b < plan.wg_start[2])
    idct_add_body<8, Pix, HBD>(lds, tx_xcd_order(b, plan.wg_start[1], plan.wg_start[2]), blocks + plan.blk_start[1], plan.count[1], coeffs, f);
  else if (b < plan.wg_start[3])
    idct_add_body<16, Pix, HBD>(lds, tx_xcd_order(b, plan.wg_start[2], plan.wg_start[3]), blocks + plan.blk_start[2], plan.count[2], coeffs, f);
  else
    idct_add_body<32, Pix, HBD>(lds, tx_xcd_order(b, plan.wg_start[3], plan.wg_start[4]), blocks + plan.blk_start[3], plan.count[3], coeffs, f);
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
  TxPlan plan;
  int acc_w = 0, acc_b = 0;
  for (int k = 0; k < 4; ++k) {
    const int bpw = 256 / (4 << k);
    plan.wg_start[k] = acc_w;
    plan.blk_start[k] = acc_b;
    plan.count[k] = size_count[k];
    acc_w += (size_count[k] + bpw - 1) / bpw;
    acc_b += size_count[k];
  }
  plan.wg_start[4] = acc_w;
  if (acc_w == 0) return VP9HIP_OK;
  const FrameDev f = to_dev(frame);
  const txfm::Coefs cf = { d_coeffs, ctx->coeff16 };
  // the 32x32 class first in the grid would start the longest workgroups first, but the classes are
  // laid out 4x4 .. 32x32 like the record list; the grid is short enough (a few rounds) for this not to matter
  if (frame->hbd)
    hipLaunchKernelGGL((idct_add_all_kernel<uint16_t, true>), dim3(acc_w), dim3(256), 0, ctx->stream, d_blocks, plan, cf, f);
  else
    hipLaunchKernelGGL((idct_add_all_kernel<uint8_t, false>), dim3(acc_w), dim3(256), 0, ctx->stream, d_blocks, plan, cf, f);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
