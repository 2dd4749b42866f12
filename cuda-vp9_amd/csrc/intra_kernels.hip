// intra_kernels.hip — intra prediction (+ fused residual add), dependency-wave ordered
// (SURVEY §8 a8–a10).
//
// One launch per dependency wave; inside a launch every transform block is independent.
// A block gets a 32-lane slot (8 slots per 256-thread workgroup): the lanes assemble the edge
// (vp9_reconintra.c:262-402 build_intra_predictors: 127/129-style fill, frame-edge replication,
// above-right only for 4x4-with-have_right) into LDS as one array E[-bs..2bs] with
// E[0] = above-left, E[k] = above[k-1], E[-k] = left[k-1]; lane c then evaluates the closed
// form of the predictor for column c (vpx_dsp/intrapred.c, derivations in DESIGN.md §intra),
// runs the inverse transform of the block's coefficients exactly like txfm_kernels.hip and
// stores clip(pred + residual).
//
// Algorithmic bytes per block: bs*bs*bps written + (3*bs+1)*bps edge reads + 16 (+ bs*bs*4
// coefficients when coded).
#include "txfm_device.h"
#include "vp9hip_internal.h"
#include <stdlib.h>

namespace {

constexpr int SLOT = 32;
constexpr int SLOTS = 8;
constexpr int EOFF = 32;          // index of E[0]
constexpr int ESIZE = 32 + 1 + 64;
constexpr int TPITCH = 33;

#define AVG2(a, b) (((a) + (b) + 1) >> 1)
#define AVG3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

__device__ __forceinline__ int clip_to(int v, int maxv) { return v < 0 ? 0 : (v > maxv ? maxv : v); }

// P(r,c) for one column c, rows 0..BS-1, from the LDS edge array.
template <int BS>
__device__ __forceinline__ void predict_column(int mode, int c, const int *E, bool have_top, bool have_left,
                                               int bd, int *p) {
  const int maxv = (1 << bd) - 1;
  const int *A = E + 1;  // A[i] = above[i], A[-1] = above-left
  switch (mode) {
    case 0: {  // DC family: dc_pred[left][up] (vp9_reconintra.c:86-89)
      int sum = 0, cnt = 0;
      if (have_top) {
#pragma unroll
        for (int i = 0; i < BS; ++i) sum += A[i];
        cnt += BS;
      }
      if (have_left) {
#pragma unroll
        for (int i = 0; i < BS; ++i) sum += E[-1 - i];
        cnt += BS;
      }
      const int dc = cnt ? (sum + (cnt >> 1)) / cnt : (128 << (bd - 8));
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] = dc;
      break;
    }
    case 1:  // V
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] = A[c];
      break;
    case 2:  // H
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] = E[-1 - r];
      break;
    case 9: {  // TM
      const int tl = A[-1], a = A[c];
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] = clip_to(E[-1 - r] + a - tl, maxv);
      break;
    }
    case 3:  // D45 (intrapred.c:65-81 generic, :354-373 4x4)
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int i = r + c;
        if (BS == 4)
          p[r] = (i == 6) ? A[7] : AVG3(A[i], A[i + 1], A[i + 2]);
        else
          p[r] = (i < BS - 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : A[BS - 1];
      }
      break;
    case 8:  // D63 (:47-63 generic, :308-329 4x4)
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int k = r >> 1, i = c + k;
        if (BS != 4 && r >= 2 && c >= BS - 1 - k)
          p[r] = A[BS - 1];
        else
          p[r] = (r & 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : AVG2(A[i], A[i + 1]);
      }
      break;
    case 7:  // D207 (:21-45): walks down the left edge, L[j] = E[-1-j]
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int i = r + (c >> 1);
        if (i >= BS - 1)
          p[r] = E[-BS];
        else if (!(c & 1))
          p[r] = AVG2(E[-1 - i], E[-2 - i]);
        else
          p[r] = AVG3(E[-1 - i], E[-2 - i], E[-1 - (i + 2 < BS ? i + 2 : BS - 1)]);
      }
      break;
    // The four "e" / VP8-style 4x4 predictors of the dispatch table (vpx_dsp_rtcd_defs.pl:46, 51, 57, 70;
    // intrapred.c:250-280, 331-350, 375-393).  VP9 itself never selects them; the twins exist for the table.
    case 13:  // D45E: as D45 4x4 but the last sample is AVG3(G, H, H)
      if (BS == 4) {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
          const int i = r + c;
          p[r] = AVG3(A[i], A[i + 1], A[i + 2 > 7 ? 7 : i + 2]);
        }
      }
      break;
    case 14:  // D63E
      if (BS == 4) {
        p[0] = AVG2(A[c], A[c + 1]);
        p[1] = AVG3(A[c], A[c + 1], A[c + 2]);
        p[2] = c < 3 ? AVG2(A[c + 1], A[c + 2]) : AVG3(A[4], A[5], A[6]);
        p[3] = c < 3 ? AVG3(A[c + 1], A[c + 2], A[c + 3]) : AVG3(A[5], A[6], A[7]);
      }
      break;
    case 15:  // HE: rows smoothed along the left edge, L[-1] = above-left, L[4] = L[3]
      if (BS == 4) {
#pragma unroll
        for (int r = 0; r < BS; ++r) p[r] = AVG3(r == 0 ? A[-1] : E[-r], E[-1 - r], r == 3 ? E[-4] : E[-2 - r]);
      }
      break;
    case 16:  // VE: columns smoothed along the above row
      if (BS == 4) {
#pragma unroll
        for (int r = 0; r < BS; ++r) p[r] = AVG3(A[c - 1], A[c], A[c + 1]);
      }
      break;
    case 4:  // D135 (:109-139): constant along d = c - r, AVG3(E[d-1],E[d],E[d+1]) around E[d]
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int d = c - r;
        p[r] = AVG3(E[d - 1], E[d], E[d + 1]);
      }
      break;
    case 5:  // D117 (:83-107): P(r,c) = P(r-2,c-1)
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int k = (r >> 1) < c ? (r >> 1) : c;
        const int rr = r - 2 * k, cc = c - k;
        if (rr == 0)
          p[r] = AVG2(E[cc], E[cc + 1]);
        else if (rr == 1)
          p[r] = AVG3(E[cc - 1], E[cc], E[cc + 1]);
        else
          p[r] = AVG3(E[2 - rr], E[1 - rr], E[-rr]);
      }
      break;
    case 6:  // D153 (:141-165): P(r,c) = P(r-1,c-2)
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int k = r < (c >> 1) ? r : (c >> 1);
        const int rr = r - k, cc = c - 2 * k;
        if (cc == 0)
          p[r] = AVG2(E[-rr], E[-rr - 1]);
        else if (cc == 1)
          p[r] = AVG3(E[1 - rr], E[-rr], E[-rr - 1]);
        else
          p[r] = AVG3(E[cc - 2], E[cc - 1], E[cc]);
      }
      break;
    default:
#pragma unroll
      for (int r = 0; r < BS; ++r) p[r] = 0;
  }
}

template <int N, bool HBD>
__device__ __forceinline__ void row_pass(int *tile, int t, int tx_type, bool lossless) {
  int v[N];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = tile[t * TPITCH + k];
  if (lossless) {
    if constexpr (N == 4) txfm::iwht4(v, true);
  } else if (N < 32 && (tx_type & 2)) {
    if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
  } else {
    txfm::idct1d<N, HBD>(v);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) tile[t * TPITCH + k] = v[k];
}

template <int N, bool HBD>
__device__ __forceinline__ void col_pass(const int *tile, int t, int tx_type, bool lossless, int *v) {
  constexpr int shift = N == 4 ? 4 : (N == 8 ? 5 : 6);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = tile[k * TPITCH + t];
  if (lossless) {
    if constexpr (N == 4) txfm::iwht4(v, false);
    return;
  }
  if (N < 32 && (tx_type & 1)) {
    if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
  } else {
    txfm::idct1d<N, HBD>(v);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = txfm::add32(v[k], 1 << (shift - 1)) >> shift;
}

// Residual column t (rows 0..N-1) of a coded block: DC-only forms, a residual handed in directly, or the
// column pass over the row-pass output in `tile`.
template <int N, bool HBD>
__device__ __forceinline__ void block_residual(const vp9hip_intra_task &tk, int t, const int *tile, int dc_coeff,
                                               int dc_kind, int *v) {
  if (dc_kind == 1) {  // vpx_idctNxN_1_add_c
    const int a1 = txfm::dc_only<N, HBD>(dc_coeff);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = a1;
  } else if (dc_kind == 2) {  // vpx_iwht4x4_1_add_c
    txfm::i64 a1 = dc_coeff >> 2, e1 = a1 >> 1;
    a1 -= e1;
    const int ip = t == 0 ? (int)a1 : (int)e1;
    const int e = ip >> 1;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = e;
    v[0] = ip - e;
  } else if (tk.tx_type & 0x40) {  // residual given directly (residual-plane mode), raster NxN
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tile[k * TPITCH + t];
  } else {
    col_pass<N, HBD>(tile, t, tk.tx_type & 3, tk.tx_type & 0x80, v);
  }
}

// Where intra_residual_kernel leaves the residual of every coded island task and the island walk
// picks it up: one int32 per sample at the sample's position, planes back to back (context scratch).
struct ResidDev {
  int *p;
  int off[3], stride[3];
};

template <int N, typename Pix, bool HBD>
__device__ __forceinline__ void finish_block(const vp9hip_intra_task &tk, int t, const int *E, const int *tile,
                                             bool coded, int dc_coeff, int dc_kind, const FrameDev &f) {
  if (t >= N) return;
  int p[N];
  predict_column<N>(tk.mode, t, E, tk.flags & 1, (tk.flags >> 1) & 1, f.bit_depth, p);
  if (coded) {
    int v[N];
    block_residual<N, HBD>(tk, t, tile, dc_coeff, dc_kind, v);
    const int maxv = (1 << f.bit_depth) - 1;
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = clip_to(txfm::add32(p[k], v[k]), maxv);
  }
  const int pl = tk.plane;
  const int x = tk.x + t;
  if (x >= f.awidth[pl]) return;
  Pix *dst = (Pix *)f.plane[pl] + (size_t)tk.y * f.stride[pl] + x;
  const int rows = min(N, f.aheight[pl] - (int)tk.y);
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k < rows) dst[(size_t)k * f.stride[pl]] = (Pix)p[k];
}

// The walk's form: the residual was computed ahead of the walk (it does not depend on the neighbours)
// and v[] was loaded before the edges were assembled.
template <int N, typename Pix>
__device__ __forceinline__ void finish_block_res(const vp9hip_intra_task &tk, int t, const int *E, const int *v,
                                                 bool coded, const FrameDev &f) {
  if (t >= N) return;
  int p[N];
  predict_column<N>(tk.mode, t, E, tk.flags & 1, (tk.flags >> 1) & 1, f.bit_depth, p);
  if (coded) {
    const int maxv = (1 << f.bit_depth) - 1;
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = clip_to(txfm::add32(p[k], v[k]), maxv);
  }
  const int pl = tk.plane;
  const int x = tk.x + t;
  if (x >= f.awidth[pl]) return;
  Pix *dst = (Pix *)f.plane[pl] + (size_t)tk.y * f.stride[pl] + x;
  const int rows = min(N, f.aheight[pl] - (int)tk.y);
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k < rows) dst[(size_t)k * f.stride[pl]] = (Pix)p[k];
}

// Up to SLOTS (8) independent transform blocks, one per 32-lane slot.  `active` slots predict (and
// add the residual of) tasks[index]; every thread of the workgroup calls this.
// A block's slot (32 lanes) lies inside one wavefront and edge[] / tiles[] of a slot are touched by that
// slot only; a wave's LDS operations execute in issue order.  The stages of a chunk therefore need the
// compiler pinned and lgkmcnt drained, not a workgroup barrier (which made the four waves of an island
// wait for the slowest one, twice per chunk, and drained its global stores as well).
__device__ __forceinline__ void slot_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local"); }

template <typename Pix, bool HBD>
__device__ __forceinline__ void intra_chunk(int (*edge)[ESIZE], int (*tiles)[32 * TPITCH],
                                            const vp9hip_intra_task *__restrict__ tasks, int index, bool active,
                                            const int32_t *__restrict__ coeffs, const FrameDev &f) {
  const int slot = threadIdx.x / SLOT, t = threadIdx.x % SLOT;
  vp9hip_intra_task tk;
  memset(&tk, 0, sizeof(tk));
  if (active) tk = tasks[index];
  const int bs = 4 << tk.tx_size;
  const int pl = tk.plane;
  const bool lossless = tk.tx_type & 0x80;
  const bool identity = tk.tx_type & 0x40;  // the "coefficients" are already the residual
  // coded: residual present; eob<=1 blocks take the DC-only forms (vp9_idct.c:119-204)
  const bool coded = active && coeffs != nullptr && tk.eob > 0;
  int *E = edge[slot] + EOFF;
  int *tile = tiles[slot];
  int dc_kind = 0, dc_coeff = 0;

  if (active) {
    const Pix *plane = (const Pix *)f.plane[pl];
    const int stride = f.stride[pl];
    const int fw = f.awidth[pl], fh = f.aheight[pl];
    const int base = 128 << (f.bit_depth - 8);
    const bool have_top = tk.flags & 1, have_left = (tk.flags >> 1) & 1, have_right = (tk.flags >> 2) & 1;
    const int x = tk.x, y = tk.y;
    // above: entries 0..2bs-1 (+ above-left); AR rule of vp9_reconintra.c:349-393 — entries
    // below bs are identical to the above-only rule (:322-347)
    {
      int take;
      // bit 3 (raw edges, used by the rtcd twins): take all 2*bs above samples verbatim
      const bool ext = (bs == 4 && have_right) || (tk.flags & 8);
      if (x + 2 * bs <= fw)
        take = ext ? 2 * bs : bs;
      else if (x + bs <= fw)
        take = ext ? fw - x : bs;
      else
        take = fw - x;
      for (int i = t; i < 2 * bs; i += SLOT) {
        int v;
        if (have_top)
          v = plane[(size_t)(y - 1) * stride + x + (i < take ? i : take - 1)];
        else
          v = base - 1;
        E[1 + i] = v;
      }
      if (t == 0) E[0] = have_top ? (have_left ? (int)plane[(size_t)(y - 1) * stride + x - 1] : base + 1) : base - 1;
    }
    {
      const int valid = (y + bs <= fh) ? bs : fh - y;
      for (int i = t; i < bs; i += SLOT) {
        int v;
        if (have_left)
          v = plane[(size_t)(y + (i < valid ? i : valid - 1)) * stride + x - 1];
        else
          v = base + 1;
        E[-1 - i] = v;
      }
    }
    if (coded) {
      const int32_t *src = coeffs + tk.coeff_off;
      if (!identity) {
        if (!lossless && ((tk.tx_type & 3) == 0 || bs == 32) && (bs == 4 ? tk.eob <= 1 : tk.eob == 1)) dc_kind = 1;
        if (lossless && tk.eob <= 1) dc_kind = 2;
      }
      if (dc_kind)
        dc_coeff = src[0];  // only the DC term is defined (and read) at eob <= 1
      else if (t < bs) {
        const int rd = identity ? bs : txfm::coeff_rows(tk.eob, lossless ? 0 : (tk.tx_type & 3), bs);
        for (int i = 0; i < bs; ++i) tile[i * TPITCH + t] = i < rd ? src[i * bs + t] : 0;
      }
    }
  }
  slot_sync();
  if (coded && !dc_kind && !identity && t < bs) {
    const int tt = tk.tx_type & 3;
    switch (tk.tx_size) {
      case 0: row_pass<4, HBD>(tile, t, tt, lossless); break;
      case 1: row_pass<8, HBD>(tile, t, tt, false); break;
      case 2: row_pass<16, HBD>(tile, t, tt, false); break;
      default: row_pass<32, HBD>(tile, t, 0, false); break;
    }
  }
  slot_sync();
  if (active) {
    switch (tk.tx_size) {
      case 0: finish_block<4, Pix, HBD>(tk, t, E, tile, coded, dc_coeff, dc_kind, f); break;
      case 1: finish_block<8, Pix, HBD>(tk, t, E, tile, coded, dc_coeff, dc_kind, f); break;
      case 2: finish_block<16, Pix, HBD>(tk, t, E, tile, coded, dc_coeff, dc_kind, f); break;
      default: finish_block<32, Pix, HBD>(tk, t, E, tile, coded, dc_coeff, dc_kind, f); break;
    }
  }
}

// ---- residual ahead of the walk ----------------------------------------------------------------
// The residual of an intra block depends on its coefficients only; the prediction depends on the
// neighbours.  intra_residual_kernel runs the inverse transforms of every coded island task in
// parallel (no waves) into ResidDev; the walk (intra_chunk_res) then has only edge assembly +
// prediction + add on its dependent chain, and its residual loads are issued before the edge loads.
template <int N, bool HBD>
__device__ __forceinline__ void residual_block(const vp9hip_intra_task &tk, int t, int *tile, int dc_coeff, int dc_kind,
                                               const ResidDev &rd, const FrameDev &f) {
  if (t >= N) return;
  int v[N];
  block_residual<N, HBD>(tk, t, tile, dc_coeff, dc_kind, v);
  const int pl = tk.plane;
  const int x = tk.x + t;
  if (x >= f.awidth[pl]) return;
  int *dst = rd.p + rd.off[pl] + (size_t)tk.y * rd.stride[pl] + x;
  const int rows = min(N, f.aheight[pl] - (int)tk.y);
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (k < rows) dst[(size_t)k * rd.stride[pl]] = v[k];
}

template <bool HBD>
__device__ __forceinline__ void residual_chunk(int (*tiles)[32 * TPITCH], const vp9hip_intra_task *__restrict__ tasks,
                                               int index, bool active, const int32_t *__restrict__ coeffs,
                                               const ResidDev &rd, const FrameDev &f) {
  const int slot = threadIdx.x / SLOT, t = threadIdx.x % SLOT;
  vp9hip_intra_task tk;
  memset(&tk, 0, sizeof(tk));
  if (active) tk = tasks[index];
  const int bs = 4 << tk.tx_size;
  const bool lossless = tk.tx_type & 0x80;
  const bool identity = tk.tx_type & 0x40;
  const bool coded = active && tk.eob > 0;
  int *tile = tiles[slot];
  int dc_kind = 0, dc_coeff = 0;
  if (coded) {  // same selection of forms as intra_chunk
    const int32_t *src = coeffs + tk.coeff_off;
    if (!identity) {
      if (!lossless && ((tk.tx_type & 3) == 0 || bs == 32) && (bs == 4 ? tk.eob <= 1 : tk.eob == 1)) dc_kind = 1;
      if (lossless && tk.eob <= 1) dc_kind = 2;
    }
    if (dc_kind)
      dc_coeff = src[0];
    else if (t < bs) {
      const int rd = identity ? bs : txfm::coeff_rows(tk.eob, lossless ? 0 : (tk.tx_type & 3), bs);
      for (int i = 0; i < bs; ++i) tile[i * TPITCH + t] = i < rd ? src[i * bs + t] : 0;
    }
  }
  slot_sync();
  if (coded && !dc_kind && !identity && t < bs) {
    const int tt = tk.tx_type & 3;
    switch (tk.tx_size) {
      case 0: row_pass<4, HBD>(tile, t, tt, lossless); break;
      case 1: row_pass<8, HBD>(tile, t, tt, false); break;
      case 2: row_pass<16, HBD>(tile, t, tt, false); break;
      default: row_pass<32, HBD>(tile, t, 0, false); break;
    }
  }
  slot_sync();
  if (coded) {
    switch (tk.tx_size) {
      case 0: residual_block<4, HBD>(tk, t, tile, dc_coeff, dc_kind, rd, f); break;
      case 1: residual_block<8, HBD>(tk, t, tile, dc_coeff, dc_kind, rd, f); break;
      case 2: residual_block<16, HBD>(tk, t, tile, dc_coeff, dc_kind, rd, f); break;
      default: residual_block<32, HBD>(tk, t, tile, dc_coeff, dc_kind, rd, f); break;
    }
  }
  slot_sync();  // the tile is reused by the next chunk of this slot
}

constexpr int RESID_Y = 8;  // workgroups per island: workgroup (i, j) takes chunks j, j + RESID_Y, ... of island i
template <bool HBD>
__global__ __launch_bounds__(256) void intra_residual_kernel(const vp9hip_intra_task *__restrict__ tasks,
                                                             const vp9hip_intra_island *__restrict__ islands,
                                                             const int32_t *__restrict__ wave_off,
                                                             const int32_t *__restrict__ coeffs, ResidDev rd, FrameDev f) {
  __shared__ int tiles[SLOTS][32 * TPITCH];
  const vp9hip_intra_island isl = islands[blockIdx.x];
  const int n = wave_off[isl.wave_off_start + isl.n_waves];  // tasks of the island
  const int slot = threadIdx.x / SLOT;
  for (int base = blockIdx.y * SLOTS; base < n; base += RESID_Y * SLOTS) {
    const int ti = base + slot;
    residual_chunk<HBD>(tiles, tasks, isl.task_start + ti, ti < n, coeffs, rd, f);
  }
}

template <typename Pix>
__device__ __forceinline__ void intra_chunk_res(int (*edge)[ESIZE], const vp9hip_intra_task *__restrict__ tasks, int index,
                                                bool active, const ResidDev &rd, const FrameDev &f) {
  const int slot = threadIdx.x / SLOT, t = threadIdx.x % SLOT;
  vp9hip_intra_task tk;
  memset(&tk, 0, sizeof(tk));
  if (active) tk = tasks[index];
  const int bs = 4 << tk.tx_size;
  const int pl = tk.plane;
  const bool coded = active && rd.p != nullptr && tk.eob > 0;
  int *E = edge[slot] + EOFF;
  // residual column of this lane: independent of the neighbours, so for the small blocks (the bulk of a
  // deep chain) the loads go out before the edge loads; 16x16 / 32x32 columns are loaded where they
  // are used (32 more live registers across the edge assembly would be parked in AGPRs)
  int v4[4], v8[8];
  const bool has_col = coded && t < bs && (int)tk.x + t < f.awidth[pl];
  const int *rp = rd.p + rd.off[pl] + (size_t)tk.y * rd.stride[pl] + tk.x + t;
  const int rs = rd.stride[pl];
  const int rrows = min(bs, f.aheight[pl] - (int)tk.y);
  if (has_col) {
    if (tk.tx_size == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v4[k] = k < rrows ? rp[(size_t)k * rs] : 0;
    } else if (tk.tx_size == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v8[k] = k < rrows ? rp[(size_t)k * rs] : 0;
    }
  }
  if (active) {
    const Pix *plane = (const Pix *)f.plane[pl];
    const int stride = f.stride[pl];
    const int fw = f.awidth[pl], fh = f.aheight[pl];
    const int base = 128 << (f.bit_depth - 8);
    const bool have_top = tk.flags & 1, have_left = (tk.flags >> 1) & 1, have_right = (tk.flags >> 2) & 1;
    const int x = tk.x, y = tk.y;
    {  // above row (same rules as intra_chunk)
      int take;
      const bool ext = (bs == 4 && have_right) || (tk.flags & 8);
      if (x + 2 * bs <= fw)
        take = ext ? 2 * bs : bs;
      else if (x + bs <= fw)
        take = ext ? fw - x : bs;
      else
        take = fw - x;
      for (int i = t; i < 2 * bs; i += SLOT) {
        int e;
        if (have_top)
          e = plane[(size_t)(y - 1) * stride + x + (i < take ? i : take - 1)];
        else
          e = base - 1;
        E[1 + i] = e;
      }
      if (t == 0) E[0] = have_top ? (have_left ? (int)plane[(size_t)(y - 1) * stride + x - 1] : base + 1) : base - 1;
    }
    {
      const int valid = (y + bs <= fh) ? bs : fh - y;
      for (int i = t; i < bs; i += SLOT) {
        int e;
        if (have_left)
          e = plane[(size_t)(y + (i < valid ? i : valid - 1)) * stride + x - 1];
        else
          e = base + 1;
        E[-1 - i] = e;
      }
    }
  }
  slot_sync();
  if (active) {
    switch (tk.tx_size) {
      case 0: finish_block_res<4, Pix>(tk, t, E, v4, coded, f); break;
      case 1: finish_block_res<8, Pix>(tk, t, E, v8, coded, f); break;
      case 2: {
        int v16[16];
        if (has_col) {
#pragma unroll
          for (int k = 0; k < 16; ++k) v16[k] = k < rrows ? rp[(size_t)k * rs] : 0;
        }
        finish_block_res<16, Pix>(tk, t, E, v16, coded, f);
        break;
      }
      default: {
        int v32[32];
        if (has_col) {
#pragma unroll
          for (int k = 0; k < 32; ++k) v32[k] = k < rrows ? rp[(size_t)k * rs] : 0;
        }
        finish_block_res<32, Pix>(tk, t, E, v32, coded, f);
        break;
      }
    }
  }
  slot_sync();  // E is rewritten by the next chunk of this slot
}

// One launch per dependency wave of the whole frame (deep structures: key frames).
template <typename Pix, bool HBD>
__global__ __launch_bounds__(256) void intra_wave_kernel(const vp9hip_intra_task *__restrict__ tasks, int first,
                                                         int count, const int32_t *__restrict__ coeffs, FrameDev f) {
  __shared__ int edge[SLOTS][ESIZE];
  __shared__ int tiles[SLOTS][32 * TPITCH];
  const int ti = blockIdx.x * SLOTS + threadIdx.x / SLOT;
  intra_chunk<Pix, HBD>(edge, tiles, tasks, first + ti, ti < count, coeffs, f);
}

// One workgroup per ISLAND (connected component of the intra dependency graph, e.g. an intra
// superblock inside an inter frame): it walks the island's waves in order, 8 blocks at a time,
// with a workgroup barrier between waves — no kernel boundary, no inter-workgroup traffic.
// __syncthreads() orders the global stores of one wave before the edge loads of the next for
// the threads of this workgroup (same CU, same L1).
template <typename Pix, bool HBD, bool RES>
__device__ __forceinline__ void intra_island_body(const vp9hip_intra_task *__restrict__ tasks,
                                                  const vp9hip_intra_island *__restrict__ islands,
                                                  const int32_t *__restrict__ wave_off, const int32_t *__restrict__ coeffs,
                                                  const ResidDev &rd, const FrameDev &f, int *__restrict__ sb_done, int sb_cols,
                                                  int island) {
  __shared__ int edge[SLOTS][ESIZE];
  __shared__ int tiles[RES ? 1 : SLOTS][RES ? 1 : 32 * TPITCH];
  const vp9hip_intra_island isl = islands[island];
  // The two slots of a wavefront run their blocks one after the other wherever the blocks differ (size,
  // mode), and most waves of a deep chain have four tasks or fewer: task j of a chunk goes to wavefront
  // j % 4 (slot 2 * (j % 4) + j / 4), so that up to four tasks get a wavefront each.  With slot = task the
  // first wavefront carried two blocks in every wave (4800 cycles per wave against 3100 / 1400 / 700 for
  // the other three, in-kernel stamps on the deepest island of the bench frame).
  const int slot = ((threadIdx.x / SLOT) & 1) * (SLOTS / 2) + (threadIdx.x / SLOT) / 2;
  // the offsets of a wave are fetched a wave ahead (a scalar load, waited for only when they are used)
  const int32_t *wo = wave_off + isl.wave_off_start;
  const int nw = (int)isl.n_waves;
  int begin = wo[0], end = wo[nw > 0 ? 1 : 0];
  for (int w = 0; w < nw; ++w) {
    const int nend = wo[w + 2 <= nw ? w + 2 : nw];
    for (int base = begin; base < end; base += SLOTS) {
      const int ti = base + slot;
      if constexpr (RES)
        intra_chunk_res<Pix>(edge, tasks, isl.task_start + ti, ti < end, rd, f);
      else
        intra_chunk<Pix, HBD>(edge, (int (*)[32 * TPITCH])tiles, tasks, isl.task_start + ti, ti < end, coeffs, f);
    }
    __syncthreads();
    // Overlap with the loop filter (vp9hip_intra_islands_lf): a task with bit 0 of `reserved` set is the
    // LAST task of this island inside its luma superblock (the packer marks it); once its wave is done,
    // everything this island does inside that superblock is done, and the superblock's counter goes up —
    // the loop filter follows the walk superblock by superblock instead of waiting for whole islands
    // (the deepest one runs for ~60 waves).  Every wave's stores are complete (the __syncthreads above
    // drains vmcnt and joins the waves); an agent-scope release writes this XCD's dirty L2 lines back
    // before the counter moves: the producer half of the hand-off recipe of MI355X_MICROARCH.md.
    // (Write-through stores instead of the fence were measured: they put the memory round trip into
    // every wave of the chain, 240 -> 384 us for the walk.)
    if (sb_done != nullptr) {
      for (int ti = begin + (int)threadIdx.x; ti < end; ti += (int)blockDim.x) {
        const vp9hip_intra_task tk = tasks[isl.task_start + ti];
        if (tk.reserved & 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          // chroma subsampling from the plane sizes (the overlapped call takes 4:2:0 or single-plane frames)
          const int sx = tk.plane && f.awidth[tk.plane] < f.awidth[0], sy = tk.plane && f.aheight[tk.plane] < f.aheight[0];
          const int sb = (((int)tk.y << sy) >> 6) * sb_cols + (((int)tk.x << sx) >> 6);
          __hip_atomic_fetch_add(&sb_done[sb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    begin = end;
    end = nend;
  }
}

template <typename Pix, bool HBD, bool RES>
__global__ __launch_bounds__(256) void intra_island_kernel(const vp9hip_intra_task *__restrict__ tasks,
                                                           const vp9hip_intra_island *__restrict__ islands,
                                                           const int32_t *__restrict__ wave_off,
                                                           const int32_t *__restrict__ coeffs, ResidDev rd, FrameDev f,
                                                           int *__restrict__ sb_done, int sb_cols) {
  intra_island_body<Pix, HBD, RES>(tasks, islands, wave_off, coeffs, rd, f, sb_done, sb_cols, (int)blockIdx.x);
}

}  // namespace

#ifndef VP9HIP_INTRA_DEVICE_ONLY  // lf_kernels.hip includes this file for the device code above (fused walk + filter)
// Residual scratch of the context: one int32 per sample of the frame, planes back to back.  Growing it
// synchronises, so callers that fork streams make sure of it first.
int vp9hip_ensure_resid(vp9hip_ctx *ctx, const vp9hip_frame *frame) {
  size_t need = 0;
  for (int pl = 0; pl < 3; ++pl)
    if (frame->plane[pl]) need += (size_t)frame->awidth[pl] * frame->aheight[pl] * sizeof(int);
  if (need <= ctx->resid_bytes) return VP9HIP_OK;
  if (ctx->resid) VP9HIP_CHECK(ctx, hipFree(ctx->resid));
  ctx->resid = nullptr;
  ctx->resid_bytes = 0;
  VP9HIP_CHECK(ctx, hipMalloc(&ctx->resid, need));
  ctx->resid_bytes = need;
  return VP9HIP_OK;
}

#endif  // !VP9HIP_INTRA_DEVICE_ONLY

static ResidDev resid_dev(vp9hip_ctx *ctx, const vp9hip_frame *frame) {
  ResidDev rd;
  memset(&rd, 0, sizeof(rd));
  rd.p = (int *)ctx->resid;
  int acc = 0;
  for (int pl = 0; pl < 3; ++pl) {
    rd.off[pl] = acc;
    rd.stride[pl] = frame->awidth[pl];
    if (frame->plane[pl]) acc += frame->awidth[pl] * frame->aheight[pl];
  }
  return rd;
}

#ifndef VP9HIP_INTRA_DEVICE_ONLY

static int residual_launch(vp9hip_ctx *ctx, hipStream_t st, const vp9hip_intra_task *d_tasks,
                           const vp9hip_intra_island *d_islands, int n_islands, const int32_t *d_wave_off,
                           const int32_t *d_coeffs, const vp9hip_frame *frame) {
  const FrameDev f = to_dev(frame);
  const ResidDev rd = resid_dev(ctx, frame);
  if (frame->hbd)
    hipLaunchKernelGGL((intra_residual_kernel<true>), dim3(n_islands, RESID_Y), dim3(256), 0, st, d_tasks, d_islands,
                       d_wave_off, d_coeffs, rd, f);
  else
    hipLaunchKernelGGL((intra_residual_kernel<false>), dim3(n_islands, RESID_Y), dim3(256), 0, st, d_tasks, d_islands,
                       d_wave_off, d_coeffs, rd, f);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}

// Optional head start (include/vp9hip.h): the pre-pass reads the lists and the coefficients only, so it
// can run beside the frame's convolve and transforms.  It goes to the context's second stream, ordered
// after everything enqueued so far (the uploads of its inputs); the island launch of the same lists
// waits for it instead of running it.
extern "C" int vp9hip_intra_residual_begin(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                                           const vp9hip_intra_island *d_islands, int n_islands,
                                           const int32_t *d_wave_off, const int32_t *d_coeffs,
                                           const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!d_tasks || !d_islands || n_islands < 0 || !d_wave_off || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_residual_begin: bad argument");
  ctx->resid_tasks = nullptr;
  if (n_islands == 0 || !d_coeffs) return VP9HIP_OK;
  int rc = vp9hip_ensure_resid(ctx, frame);
  if (rc) return rc;
  if (!ctx->stream2) VP9HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  if (!ctx->ev_resid_start) {
    VP9HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_resid_start, hipEventDisableTiming));
    VP9HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_resid_done, hipEventDisableTiming));
  }
  VP9HIP_CHECK(ctx, hipEventRecord(ctx->ev_resid_start, ctx->stream));
  VP9HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_resid_start, 0));
  // the zero-fill of the filter / island counters of vp9hip_intra_islands_lf rides along (4:2:0 or luma only)
  ctx->lf_zeroed_rows = ctx->lf_zeroed_cols = 0;
  if (frame->aheight[0] <= 128 * 64 && frame->awidth[0] <= 128 * 64) {
    rc = vp9hip_lf_zero_counters(ctx, frame, ctx->stream2);
    if (rc) return rc;
  }
  rc = residual_launch(ctx, ctx->stream2, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, frame);
  if (rc) return rc;
  VP9HIP_CHECK(ctx, hipEventRecord(ctx->ev_resid_done, ctx->stream2));
  ctx->resid_tasks = d_tasks;
  ctx->resid_coeffs = d_coeffs;
  return VP9HIP_OK;
}

// The residual of the island tasks on stream `st`: the pre-pass vp9hip_intra_residual_begin started for these
// lists is waited for, or it is run now.  After this the island walk of the lists may be enqueued on `st`.
int vp9hip_islands_prepare(vp9hip_ctx *ctx, hipStream_t st, const vp9hip_intra_task *d_tasks,
                           const vp9hip_intra_island *d_islands, int n_islands, const int32_t *d_wave_off,
                           const int32_t *d_coeffs, const vp9hip_frame *frame) {
  if (d_coeffs) {
    // the inverse transforms of every coded task, in parallel, ahead of the dependent walk
    int rc = vp9hip_ensure_resid(ctx, frame);
    if (rc) return rc;
    if (ctx->resid_tasks == d_tasks && ctx->resid_coeffs == d_coeffs) {
      VP9HIP_CHECK(ctx, hipStreamWaitEvent(st, ctx->ev_resid_done, 0));  // vp9hip_intra_residual_begin did it
    } else {
      rc = residual_launch(ctx, st, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, frame);
      if (rc) return rc;
    }
  }
  ctx->resid_tasks = nullptr;
  return VP9HIP_OK;
}

int vp9hip_islands_launch(vp9hip_ctx *ctx, hipStream_t st, const vp9hip_intra_task *d_tasks,
                          const vp9hip_intra_island *d_islands, int n_islands, const int32_t *d_wave_off,
                          const int32_t *d_coeffs, const vp9hip_frame *frame, int *d_sb_done, int sb_cols) {
  const FrameDev f = to_dev(frame);
  ResidDev rd;
  memset(&rd, 0, sizeof(rd));
  int rc = vp9hip_islands_prepare(ctx, st, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, frame);
  if (rc) return rc;
  if (d_coeffs) rd = resid_dev(ctx, frame);
  if (frame->hbd)
    hipLaunchKernelGGL((intra_island_kernel<uint16_t, true, true>), dim3(n_islands), dim3(256), 0, st, d_tasks, d_islands,
                       d_wave_off, d_coeffs, rd, f, d_sb_done, sb_cols);
  else
    hipLaunchKernelGGL((intra_island_kernel<uint8_t, false, true>), dim3(n_islands), dim3(256), 0, st, d_tasks, d_islands,
                       d_wave_off, d_coeffs, rd, f, d_sb_done, sb_cols);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}

extern "C" int vp9hip_intra_pred_islands(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                                         const vp9hip_intra_island *d_islands, int n_islands,
                                         const int32_t *d_wave_off, const int32_t *d_coeffs,
                                         const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_tasks || !d_islands || n_islands < 0 || !d_wave_off || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_islands: bad argument");
  if (n_islands == 0) return VP9HIP_OK;
  return vp9hip_islands_launch(ctx, ctx->stream, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, frame, nullptr, 0);
}

extern "C" int vp9hip_intra_pred_waves(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks, const int32_t *wave_start,
                                       int n_waves, const int32_t *d_coeffs, const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_tasks || !wave_start || n_waves < 0 || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_waves: bad argument");
  const FrameDev f = to_dev(frame);
  for (int w = 0; w < n_waves; ++w) {
    const int first = wave_start[w], count = wave_start[w + 1] - first;
    if (count < 0) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_waves: wave_start not monotonic");
    if (count == 0) continue;
    const int grid = (count + SLOTS - 1) / SLOTS;
    if (frame->hbd)
      hipLaunchKernelGGL((intra_wave_kernel<uint16_t, true>), dim3(grid), dim3(256), 0, ctx->stream, d_tasks, first,
                         count, d_coeffs, f);
    else
      hipLaunchKernelGGL((intra_wave_kernel<uint8_t, false>), dim3(grid), dim3(256), 0, ctx->stream, d_tasks, first,
                         count, d_coeffs, f);
  }
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
#endif  // !VP9HIP_INTRA_DEVICE_ONLY
