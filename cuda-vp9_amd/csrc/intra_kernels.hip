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

template <int N, bool HBD, int PITCH = TPITCH>
__device__ __forceinline__ void row_pass(int *tile, int t, int tx_type, bool lossless) {
  int v[N];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = tile[t * PITCH + k];
  if (lossless) {
    if constexpr (N == 4) txfm::iwht4(v, true);
  } else if (N < 32 && (tx_type & 2)) {
    if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
  } else {
    txfm::idct1d<N, HBD>(v);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) tile[t * PITCH + k] = v[k];
}

// The row pass with the row taken straight from the coefficient buffer (N consecutive coefficients, 16 bytes per load;
// rows >= rd are zero and, in a compact slot, absent) instead of through the LDS tile: txfm_kernels.hip, round 3.
template <int N, bool HBD, int PITCH = TPITCH>
__device__ __forceinline__ void row_pass_from(int *tile, int t, int tx_type, bool lossless, const txfm::Coefs &coeffs, unsigned off, int rd) {
  int v[N];
  if (t < rd) {
    if (coeffs.c16) {
      short c[N];
      __builtin_memcpy(c, __builtin_assume_aligned((const short *)coeffs.p + off + (unsigned)(t * N), 4), N * 2);
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = c[k];
    } else {
      __builtin_memcpy(v, __builtin_assume_aligned((const int *)coeffs.p + off + (unsigned)(t * N), 4), N * 4);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = 0;
  }
  if (lossless) {
    if constexpr (N == 4) txfm::iwht4(v, true);
  } else if (N < 32 && (tx_type & 2)) {
    if constexpr (N < 32) txfm::iadst1d<N, HBD>(v);
  } else {
    txfm::idct1d<N, HBD>(v);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) tile[t * PITCH + k] = v[k];
}

template <int N, bool HBD, int PITCH = TPITCH>
__device__ __forceinline__ void col_pass(const int *tile, int t, int tx_type, bool lossless, int *v) {
  constexpr int shift = N == 4 ? 4 : (N == 8 ? 5 : 6);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = tile[k * PITCH + t];
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

__device__ __forceinline__ void slot_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local"); }

template <int N, typename Pix, bool HBD>
__device__ __forceinline__ void finish_block(const vp9hip_intra_task &tk, int t, const int *E, int *tile,
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
  // The block goes to the frame by ROWS: lane t holds column t, the columns meet in the slot's LDS tile (free once the
  // residual was taken from it) and lane t stores row t in one piece — N samples per store instead of N stores of one
  // sample per lane (txfm_kernels.hip, round 3; a 32x32 block: 32 stores per lane became two).
#pragma unroll
  for (int k = 0; k < N; ++k) tile[k * TPITCH + t] = p[k];
  slot_sync();
  const int pl = tk.plane;
  const int rows = min(N, f.aheight[pl] - (int)tk.y);
  const int cols = min(N, f.awidth[pl] - (int)tk.x);  // a multiple of 4
  if (t >= rows || cols <= 0) return;
  Pix *dst = (Pix *)f.plane[pl] + (size_t)(tk.y + t) * f.stride[pl] + tk.x;
  constexpr int SPD = 4 / (int)sizeof(Pix);  // samples per dword
  constexpr int ND = N / SPD;
  unsigned dw[ND];
#pragma unroll
  for (int q = 0; q < ND; ++q) {
    unsigned o = 0;
#pragma unroll
    for (int e = 0; e < SPD; ++e) o |= (unsigned)tile[t * TPITCH + q * SPD + e] << (e * 8 * (int)sizeof(Pix));
    dw[q] = o;
  }
  const int vd = cols / SPD;
  if (vd == ND) {
    __builtin_memcpy(__builtin_assume_aligned(dst, 4), dw, ND * 4);
  } else {
#pragma unroll
    for (int q = 0; q < ND; ++q)
      if (q < vd) ((unsigned *)dst)[q] = dw[q];
  }
}

// Up to SLOTS (8) independent transform blocks, one per 32-lane slot.  `active` slots predict (and
// add the residual of) tasks[index]; every thread of the workgroup calls this.
// A block's slot (32 lanes) lies inside one wavefront and edge[] / tiles[] of a slot are touched by that
// slot only; a wave's LDS operations execute in issue order.  The stages of a chunk therefore need the
// compiler pinned and lgkmcnt drained, not a workgroup barrier (which made the four waves of an island
// wait for the slowest one, twice per chunk, and drained its global stores as well).

template <typename Pix, bool HBD>
__device__ __forceinline__ void intra_chunk(int (*edge)[ESIZE], int (*tiles)[32 * TPITCH],
                                            const vp9hip_intra_task *__restrict__ tasks, int index, bool active,
                                            const txfm::Coefs &coeffs, const FrameDev &f) {
  const int slot = threadIdx.x / SLOT, t = threadIdx.x % SLOT;
  vp9hip_intra_task tk;
  memset(&tk, 0, sizeof(tk));
  if (active) tk = tasks[index];
  const int bs = 4 << tk.tx_size;
  const int pl = tk.plane;
  const bool lossless = tk.tx_type & 0x80;
  const bool identity = tk.tx_type & 0x40;  // the "coefficients" are already the residual
  // coded: residual present; eob<=1 blocks take the DC-only forms (vp9_idct.c:119-204)
  const bool coded = active && coeffs.p != nullptr && tk.eob > 0;
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
      const txfm::CoefAt src = txfm::at(coeffs, tk.coeff_off);
      if (!identity) {
        if (!lossless && ((tk.tx_type & 3) == 0 || bs == 32) && (bs == 4 ? tk.eob <= 1 : tk.eob == 1)) dc_kind = 1;
        if (lossless && tk.eob <= 1) dc_kind = 2;
      }
      if (dc_kind)
        dc_coeff = src[0];  // only the DC term is defined (and read) at eob <= 1
      else if (identity && t < bs) {
        for (int i = 0; i < bs; ++i) tile[i * TPITCH + t] = src[i * bs + t];
      }
    }
  }
  slot_sync();
  if (coded && !dc_kind && !identity && t < bs) {  // (the rows of coefficients straight from memory into the row pass)
    const int tt = tk.tx_type & 3;
    const int rd = txfm::coeff_rows(tk.eob, lossless ? 0 : tt, bs);
    switch (tk.tx_size) {
      case 0: row_pass_from<4, HBD>(tile, t, tt, lossless, coeffs, tk.coeff_off, rd); break;
      case 1: row_pass_from<8, HBD>(tile, t, tt, false, coeffs, tk.coeff_off, rd); break;
      case 2: row_pass_from<16, HBD>(tile, t, tt, false, coeffs, tk.coeff_off, rd); break;
      default: row_pass_from<32, HBD>(tile, t, 0, false, coeffs, tk.coeff_off, rd); break;
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

// =============================================================================================
// Island walk in LDS.  An island (vp9hip.h) whose window fits is walked without touching the frame
// between its first and its last wave:
//   1. the task records go to LDS; their bounding box per plane gives the window (one row above, one
//      column to the left, four columns to the right: the above-right samples of a 4x4 block);
//   2. the window is read from the frame once (everything around the island's blocks is final: inter
//      prediction + residual ran before, and blocks of OTHER islands are never read — an edge between
//      two intra blocks would have put them into one island);
//   3. the residual of every coded block (inverse transform, all blocks side by side, no waves) is
//      stored INTO the window at the block's own position as int16, saturated: a block's samples are
//      dead until the block is predicted, and clip(pred + res) = clip(pred + sat16(res)) because
//      0 <= pred < 4096 (|res| < 2^27 for every transform, so pred + res never wraps either);
//   4. the waves: edge assembly, prediction, + residual, clip — LDS to LDS, one sample per lane;
//   5. the island's blocks are written to the frame, row pieces of four samples; then the
//      per-superblock completion marks for the loop filter (vp9hip_intra_islands_lf).
// The dependent chain of a wave is a handful of LDS round trips instead of three memory round trips
// (task record, edge samples written by the wave before, store drain), and the frame is read and
// written once per island instead of once per wave.
constexpr int ISL_TILE = VP9HIP_ISLAND_TILE_ELEMS;
constexpr int ISL_TASKS = VP9HIP_ISLAND_MAX_TASKS;
constexpr int ISL_TX32 = VP9HIP_ISLAND_MAX_TX32;
// transform scratch: per pass 64 4x4 / 32 8x8 / 16 16x16 / 4 32x32 blocks, N rows of N + 1 dwords each
constexpr int ISL_POOL_INTS = 16 * 16 * 17;              // 4352 (64 * 20 = 1280, 32 * 72 = 2304, 4 * 1056 = 4224)
constexpr int ISL_MARKS = 96;

struct IslandLds {
  short tile[ISL_TILE];
  int pool[ISL_POOL_INTS];
  int edge[SLOTS][ESIZE];
  vp9hip_intra_task tasks[ISL_TASKS];
  int box[3][4];        // x0, y0, x1, y1 of the blocks of a plane
  int idx0[3], pitch[3];  // sample (x, y) of plane p is tile[idx0[p] + y * pitch[p] + x]
  int cnt[4], fill[4];  // coded blocks per transform size; their indices, size by size, in order[]
  int nmarks, bad;
  short order[ISL_TASKS];
  int marks[ISL_MARKS];
  short woff[ISL_TASKS + 2];  // wave offsets (relative to the island's first task)
};

// Probe builds (-DVP9HIP_STAMPS, tools/ only): wall-clock stamps (100 MHz) of a workgroup's stages
#ifdef VP9HIP_STAMPS
__device__ long long g_stamps[8 * 4096];
__shared__ int s_stamp_slot;  // the workgroup's place in the launch's order (its ticket)
#define VP9HIP_STAMP_SLOT(t) (s_stamp_slot = (t))
#define VP9HIP_STAMP(k)                                                                     \
  do {                                                                                      \
    if (threadIdx.x == 0 && s_stamp_slot < 4096) g_stamps[s_stamp_slot * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define VP9HIP_STAMP_SLOT(t) do { } while (0)
#define VP9HIP_STAMP(k) do { } while (0)
#endif

// LDS of the walk through memory (islands that do not fit)
struct IslandMemLds {
  int edge[SLOTS][ESIZE];
  int tiles[SLOTS][32 * TPITCH];
};

// sum over the 32 lanes of a slot (all of them active)
__device__ __forceinline__ int slot_sum32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);  // row_mirror
  v += __builtin_amdgcn_ds_swizzle(v, 0x401f);                    // lane ^ 16
  return v;
}

// One predicted sample (r, c) of a bs x bs block: predict_column's forms with the block size at run time
// (vpx_dsp/intrapred.c; 4x4 specials :308-373).
__device__ __forceinline__ int predict_px(int mode, int bs, int r, int c, const int *E, int dc, int maxv) {
  const int *A = E + 1;
  switch (mode) {
    case 0: return dc;
    case 1: return A[c];
    case 2: return E[-1 - r];
    case 9: return clip_to(E[-1 - r] + A[c] - A[-1], maxv);
    case 3: {
      const int i = r + c;
      if (bs == 4) return (i == 6) ? A[7] : AVG3(A[i], A[i + 1], A[i + 2]);
      return (i < bs - 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : A[bs - 1];
    }
    case 8: {
      const int k = r >> 1, i = c + k;
      if (bs != 4 && r >= 2 && c >= bs - 1 - k) return A[bs - 1];
      return (r & 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : AVG2(A[i], A[i + 1]);
    }
    case 7: {
      const int i = r + (c >> 1);
      if (i >= bs - 1) return E[-bs];
      if (!(c & 1)) return AVG2(E[-1 - i], E[-2 - i]);
      return AVG3(E[-1 - i], E[-2 - i], E[-1 - (i + 2 < bs ? i + 2 : bs - 1)]);
    }
    case 4: {
      const int d = c - r;
      return AVG3(E[d - 1], E[d], E[d + 1]);
    }
    case 5: {
      const int k = (r >> 1) < c ? (r >> 1) : c;
      const int rr = r - 2 * k, cc = c - k;
      if (rr == 0) return AVG2(E[cc], E[cc + 1]);
      if (rr == 1) return AVG3(E[cc - 1], E[cc], E[cc + 1]);
      return AVG3(E[2 - rr], E[1 - rr], E[-rr]);
    }
    case 6: {
      const int k = r < (c >> 1) ? r : (c >> 1);
      const int rr = r - k, cc = c - 2 * k;
      if (cc == 0) return AVG2(E[-rr], E[-rr - 1]);
      if (cc == 1) return AVG3(E[1 - rr], E[-rr], E[-rr - 1]);
      return AVG3(E[cc - 2], E[cc - 1], E[cc]);
    }
    default: return 0;
  }
}

__device__ __forceinline__ short sat16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// forms a coded block's residual takes (vp9_idct.c:119-204): 0 full transform, 1 DC only, 2 lossless DC only,
// 3 the slot holds the residual itself (residual-plane mode)
__device__ __forceinline__ int resid_kind(const vp9hip_intra_task &tk) {
  const int bs = 4 << tk.tx_size;
  const bool lossless = tk.tx_type & 0x80;
  if (tk.tx_type & 0x40) return 3;
  if (!lossless && ((tk.tx_type & 3) == 0 || bs == 32) && (bs == 4 ? tk.eob <= 1 : tk.eob == 1)) return 1;
  if (lossless && tk.eob <= 1) return 2;
  return 0;
}

// column t of a block's residual -> the window
template <int N>
__device__ __forceinline__ void store_resid_col(short *dst, int pitch, const int *v) {
#pragma unroll
  for (int k = 0; k < N; ++k) dst[k * pitch] = sat16(v[k]);
}

// The residual of every coded N x N block of the island: N lanes per block (lane = row in the row pass, column in
// the column pass), 256 / N blocks per pass (16x16: 16, 32x32: 4 — one per wavefront).  A block's lanes lie inside one
// wavefront, whose LDS operations execute in order: a fence per stage is enough (slot_sync).
template <int N, bool HBD>
__device__ __forceinline__ void island_residual_pass(IslandLds &S, int first, int count, const txfm::Coefs &coeffs) {
  constexpr int PITCH = N + 1;
  constexpr int BPP = N == 32 ? 4 : 256 / N;      // blocks per pass
  constexpr int LPB = N == 32 ? 64 : N;           // lanes a block owns (32x32: a wavefront, half of it idle)
  const int lb = (int)threadIdx.x / LPB, t = (int)threadIdx.x % LPB;
  int *pt = S.pool + lb * N * PITCH;
  for (int base = 0; base < count; base += BPP) {
    const bool active = base + lb < count && t < N;
    vp9hip_intra_task tk;
    memset(&tk, 0, sizeof(tk));
    if (active) tk = S.tasks[S.order[first + base + lb]];
    const int kind = resid_kind(tk);
    const bool lossless = tk.tx_type & 0x80;
    const int tt = N == 32 ? 0 : (tk.tx_type & 3);
    const txfm::CoefAt src = txfm::at(coeffs, tk.coeff_off);
    const int pl = tk.plane, pitch = S.pitch[pl];
    short *dst = &S.tile[S.idx0[pl] + (int)tk.y * pitch + (int)tk.x + t];
    const bool full = active && kind == 0;
    if (full) {
      row_pass_from<N, HBD, PITCH>(pt, t, tt, N == 4 && lossless, coeffs, tk.coeff_off,
                                   txfm::coeff_rows(tk.eob, lossless ? 0 : (tk.tx_type & 3), N));
    } else if (active) {
      if (kind == 3) {  // the slot holds the residual itself
        for (int k = 0; k < N; ++k) dst[k * pitch] = sat16(src[k * N + t]);
      } else if (kind == 1) {
        const short a1 = sat16(txfm::dc_only<N, HBD>(src[0]));
        for (int k = 0; k < N; ++k) dst[k * pitch] = a1;
      } else {  // vpx_iwht4x4_1_add_c (inv_txfm.c:71-94)
        txfm::i64 a1 = src[0] >> 2, e1 = a1 >> 1;
        a1 -= e1;
        const int ip = t == 0 ? (int)a1 : (int)e1;
        const int e = ip >> 1;
        dst[0] = sat16(ip - e);
        dst[pitch] = dst[2 * pitch] = dst[3 * pitch] = sat16(e);
      }
    }
    slot_sync();
    if (full) {
      int v[N];
      col_pass<N, HBD, PITCH>(pt, t, tt, N == 4 && lossless, v);
      store_resid_col<N>(dst, pitch, v);
    }
    slot_sync();  // the pool tile is reused by the next pass
  }
}

// Returns false (workgroup-uniform) when the island does not fit: nothing was written anywhere.
// sb_done != nullptr: completion marks for the loop filter.
template <typename Pix, bool HBD>
__device__ __forceinline__ bool island_lds_body(IslandLds &S, const vp9hip_intra_task *__restrict__ tasks,
                                                const vp9hip_intra_island &isl, const int32_t *__restrict__ wave_off,
                                                const txfm::Coefs &coeffs, const FrameDev &f,
                                                int *__restrict__ sb_done, int sb_cols) {
  const int tid = threadIdx.x;
  const int32_t *wo = wave_off + isl.wave_off_start;
  const int nw = (int)isl.n_waves;
  const int n = wo[nw];
  if (n > ISL_TASKS) return false;
  VP9HIP_STAMP(0);
  if (tid < 12) S.box[tid >> 2][tid & 3] = (tid & 2) ? 0 : 0x7fffffff;
  if (tid >= 16 && tid < 20) S.cnt[tid - 16] = 0;
  if (tid == 12) S.nmarks = S.bad = 0;
  __syncthreads();
  // ---- 1. task records, wave offsets, bounding boxes, coded blocks per transform size, completion marks
  for (int i = tid; i <= nw; i += 256) S.woff[i] = (short)wo[i];  // nw <= n <= ISL_TASKS
  for (int i = tid; i < n; i += 256) {
    const vp9hip_intra_task tk = tasks[isl.task_start + i];
    S.tasks[i] = tk;
    const int bs = 4 << tk.tx_size, pl = tk.plane < 3 ? tk.plane : 0;
    atomicMin(&S.box[pl][0], (int)tk.x);
    atomicMin(&S.box[pl][1], (int)tk.y);
    atomicMax(&S.box[pl][2], (int)tk.x + bs);
    atomicMax(&S.box[pl][3], (int)tk.y + bs);
    if (tk.plane > 2 || tk.tx_size > 3 || (tk.flags & 8)) S.bad = 1;  // (raw edges: the rtcd twins' wave launches only)
    if (coeffs.p != nullptr && tk.eob > 0) atomicAdd(&S.cnt[tk.tx_size & 3], 1);
    if (sb_done != nullptr && (tk.reserved & 1)) {
      // chroma subsampling from the plane sizes (4:2:0, 4:4:4 or single-plane frames)
      const int sx = tk.plane && f.awidth[pl] < f.awidth[0], sy = tk.plane && f.aheight[pl] < f.aheight[0];
      const int k = atomicAdd(&S.nmarks, 1);
      if (k < ISL_MARKS) S.marks[k] = (((int)tk.y << sy) >> 6) * sb_cols + (((int)tk.x << sx) >> 6);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int p = 0; p < 3; ++p) {
      const int *b = S.box[p];
      if (b[2] > b[0]) {
        const int pitch = VP9HIP_ISLAND_PITCH(b[2] - b[0]);
        S.pitch[p] = pitch;
        S.idx0[p] = acc - (b[1] - 1) * pitch - (b[0] - 1);
        acc += pitch * (b[3] - b[1] + 1);
      } else {
        S.pitch[p] = S.idx0[p] = 0;
      }
    }
    if (acc > ISL_TILE || S.cnt[3] > ISL_TX32 || S.nmarks > ISL_MARKS) S.bad = 1;
    int o = 0;
    for (int k = 0; k < 4; ++k) {
      S.fill[k] = o;
      o += S.cnt[k];
    }
  }
  __syncthreads();
  if (S.bad) return false;
  VP9HIP_STAMP(1);
  const int wave = tid >> 6, lane = tid & 63;
  // ---- 2. the window, four samples per lane and load (aligned: block positions are multiples of 4, so is the plane's
  // start in memory); the indices of the coded blocks, size by size, ride along
  if (coeffs.p != nullptr)
    for (int i = tid; i < n; i += 256)
      if (S.tasks[i].eob > 0) S.order[atomicAdd(&S.fill[S.tasks[i].tx_size & 3], 1)] = (short)i;
#pragma unroll 1
  for (int p = 0; p < 3; ++p) {
    const int *b = S.box[p];
    if (b[2] <= b[0]) continue;
    const int ox = b[0] - 1, oy = b[1] - 1, W = b[2] - b[0] + 5, H = b[3] - b[1] + 1;
    const int qx0 = (ox < 0 ? ox - 3 : ox) / 4 * 4;            // first quad's x (floor to a multiple of 4)
    const int nq = (ox + W - qx0 + 3) >> 2;                     // quads per row
    const unsigned inv = ((1u << 20) + nq - 1) / nq;            // r = i / nq for i < 2^20 / nq (here i < ~4000)
    const int pitch = S.pitch[p], idx0 = S.idx0[p];
    const Pix *plane = (const Pix *)f.plane[p];
    const int stride = f.stride[p], fw = f.awidth[p], fh = f.aheight[p];
    for (int i = tid; i < nq * H; i += 256) {
      const int r = (int)(((unsigned)i * inv) >> 20), q = i - r * nq;
      const int gy = oy + r, gx = qx0 + 4 * q;
      if (gy < 0 || gy >= fh || gx < 0 || gx >= fw) continue;  // (the plane's width is a multiple of 8)
      int v0, v1, v2, v3;
      if constexpr (sizeof(Pix) == 1) {
        const unsigned w = *(const unsigned *)(plane + (size_t)gy * stride + gx);
        v0 = w & 255, v1 = (w >> 8) & 255, v2 = (w >> 16) & 255, v3 = w >> 24;
      } else {
        const uint2 w = *(const uint2 *)(plane + (size_t)gy * stride + gx);
        v0 = w.x & 0xffff, v1 = w.x >> 16, v2 = w.y & 0xffff, v3 = w.y >> 16;
      }
      short *d = &S.tile[idx0 + gy * pitch + gx];
      const int lo = ox - gx, hi = ox + W - gx;  // window columns [lo, hi) of this quad
      if (0 >= lo && 0 < hi) d[0] = (short)v0;
      if (1 >= lo && 1 < hi) d[1] = (short)v1;
      if (2 >= lo && 2 < hi) d[2] = (short)v2;
      if (3 >= lo && 3 < hi) d[3] = (short)v3;
    }
  }
  __syncthreads();
  VP9HIP_STAMP(2);
  // ---- 3. residuals, size by size
  if (coeffs.p != nullptr) {
    int first = 0;
    if (S.cnt[0]) island_residual_pass<4, HBD>(S, first, S.cnt[0], coeffs);
    first += S.cnt[0];
    if (S.cnt[1]) {
      __syncthreads();  // (the passes lay their tiles over the same scratch)
      island_residual_pass<8, HBD>(S, first, S.cnt[1], coeffs);
    }
    first += S.cnt[1];
    if (S.cnt[2]) {
      __syncthreads();
      island_residual_pass<16, HBD>(S, first, S.cnt[2], coeffs);
    }
    first += S.cnt[2];
    if (S.cnt[3]) {
      __syncthreads();
      island_residual_pass<32, HBD>(S, first, S.cnt[3], coeffs);
    }
  }
  __syncthreads();
  VP9HIP_STAMP(3);
  // ---- 4. the waves.  Task j of a chunk goes to wavefront j % 4 (slot 2 * (j % 4) + j / 4): most waves of a deep
  // chain have four tasks or fewer, and the two slots of a wavefront run one after the other where they differ
  const int wslot = ((tid / SLOT) & 1) * (SLOTS / 2) + (tid / SLOT) / 2;
  const int maxv = (1 << f.bit_depth) - 1;
  const int basev = 128 << (f.bit_depth - 8);
  int *E = S.edge[wslot] + EOFF;
  const int t = tid % SLOT;
  for (int w = 0; w < nw; ++w) {
    const int begin = S.woff[w], end = S.woff[w + 1];
    for (int base = begin; base < end; base += SLOTS) {
      const int ti = base + wslot;
      const bool active = ti < end;
      vp9hip_intra_task tk;
      memset(&tk, 0, sizeof(tk));
      if (active) tk = S.tasks[ti];
      const int bs = 4 << tk.tx_size, lg = 2 + tk.tx_size, pl = tk.plane;
      const int pitch = S.pitch[pl], idx0 = S.idx0[pl];
      const int x = tk.x, y = tk.y;
      const bool have_top = tk.flags & 1, have_left = (tk.flags >> 1) & 1, have_right = (tk.flags >> 2) & 1;
      int dc = 0;
      if (active) {
        const int fw = f.awidth[pl], fh = f.aheight[pl];
        // above: entries 0..2bs-1 (+ above-left); the rules of intra_chunk (vp9_reconintra.c:322-393)
        int take;
        const bool ext = bs == 4 && have_right;
        if (x + 2 * bs <= fw)
          take = ext ? 2 * bs : bs;
        else if (x + bs <= fw)
          take = ext ? fw - x : bs;
        else
          take = fw - x;
        int part = 0;
        const short *arow = &S.tile[idx0 + (y - 1) * pitch + x];
        for (int i = t; i < 2 * bs; i += SLOT) {
          const int v = have_top ? (int)arow[i < take ? i : take - 1] : basev - 1;
          E[1 + i] = v;
          if (have_top && i < bs) part += v;
        }
        if (t == 0) E[0] = have_top ? (have_left ? (int)arow[-1] : basev + 1) : basev - 1;
        const int valid = (y + bs <= fh) ? bs : fh - y;
        if (t < bs) {
          const int v = have_left ? (int)S.tile[idx0 + (y + (t < valid ? t : valid - 1)) * pitch + x - 1] : basev + 1;
          E[-1 - t] = v;
          if (have_left) part += v;
        }
        if (tk.mode == 0) {  // dc_pred[left][up] (vp9_reconintra.c:86-89): rounded mean of what is there
          const int sum = slot_sum32(part);
          const int lcnt = lg + (have_top && have_left ? 1 : 0);
          dc = (have_top || have_left) ? (sum + ((1 << lcnt) >> 1)) >> lcnt : basev;
        }
      }
      slot_sync();
      if (active) {
        const bool coded = coeffs.p != nullptr && tk.eob > 0;
        short *blk = &S.tile[idx0 + y * pitch + x];
        // one loop per mode (the mode is the same for the 32 lanes of a slot): the compiler overlaps the LDS reads of
        // consecutive samples, which it cannot do across a switch inside the loop
#define ISL_PREDICT(MODE)                                                  \
  for (int i = t; i < bs * bs; i += SLOT) {                                \
    const int r = i >> lg, c = i & (bs - 1);                               \
    int px = predict_px(MODE, bs, r, c, E, dc, maxv);                      \
    short *d = blk + r * pitch + c;                                        \
    if (coded) px = clip_to(px + (int)*d, maxv);                           \
    *d = (short)px;                                                        \
  }
        switch (tk.mode) {
          case 0: ISL_PREDICT(0) break;
          case 1: ISL_PREDICT(1) break;
          case 2: ISL_PREDICT(2) break;
          case 3: ISL_PREDICT(3) break;
          case 4: ISL_PREDICT(4) break;
          case 5: ISL_PREDICT(5) break;
          case 6: ISL_PREDICT(6) break;
          case 7: ISL_PREDICT(7) break;
          case 8: ISL_PREDICT(8) break;
          case 9: ISL_PREDICT(9) break;
          default: ISL_PREDICT(10) break;
        }
#undef ISL_PREDICT
      }
      slot_sync();  // E is rewritten by the slot's next block
    }
    __syncthreads();
  }
  VP9HIP_STAMP(4);
  // ---- 5. the island's blocks -> the frame: four samples per lane and store
  const int slot = tid / SLOT;
  for (int base = 0; base < n; base += SLOTS) {
    const int ti = base + slot;
    if (ti >= n) continue;
    const vp9hip_intra_task tk = S.tasks[ti];
    const int bs = 4 << tk.tx_size, lgq = tk.tx_size, pl = tk.plane;  // bs / 4 pieces per row
    const int pitch = S.pitch[pl];
    const short *blk = &S.tile[S.idx0[pl] + (int)tk.y * pitch + (int)tk.x];
    Pix *dst = (Pix *)f.plane[pl] + (size_t)tk.y * f.stride[pl] + tk.x;
    const int rows = min(bs, f.aheight[pl] - (int)tk.y), cols = min(bs, f.awidth[pl] - (int)tk.x);  // multiples of 4
    for (int i = t; i < (bs * bs) >> 2; i += SLOT) {
      const int r = i >> lgq, c = (i & ((1 << lgq) - 1)) * 4;
      if (r >= rows || c >= cols) continue;
      const short *sp = blk + r * pitch + c;
      if constexpr (sizeof(Pix) == 1) {
        const unsigned v = (unsigned)sp[0] | ((unsigned)sp[1] << 8) | ((unsigned)sp[2] << 16) | ((unsigned)sp[3] << 24);
        *(unsigned *)(dst + (size_t)r * f.stride[pl] + c) = v;
      } else {
        uint2 v;
        v.x = (unsigned)(unsigned short)sp[0] | ((unsigned)(unsigned short)sp[1] << 16);
        v.y = (unsigned)(unsigned short)sp[2] | ((unsigned)(unsigned short)sp[3] << 16);
        *(uint2 *)(dst + (size_t)r * f.stride[pl] + c) = v;
      }
    }
  }
  VP9HIP_STAMP(5);
  if (sb_done != nullptr) {
    // Completion marks.  Producer half of the hand-off recipe (MI355X_MICROARCH.md, "Valid forms"): plain stores,
    // every storing wave's vmcnt(0) + the barrier (__syncthreads), ONE agent-scope release (writes this XCD's dirty
    // L2 lines back), waited for, then the counters; the filter polls them and acquires (lf_kernels.hip, gate_col).
    __syncthreads();
    if (wave == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int k = lane; k < S.nmarks; k += 64)
        __hip_atomic_fetch_add(&sb_done[S.marks[k]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  VP9HIP_STAMP(6);
  return true;
}

// One launch per dependency wave of the whole frame (deep structures: key frames).
template <typename Pix, bool HBD>
__global__ __launch_bounds__(256) void intra_wave_kernel(const vp9hip_intra_task *__restrict__ tasks, int first,
                                                         int count, txfm::Coefs coeffs, FrameDev f) {
  __shared__ int edge[SLOTS][ESIZE];
  __shared__ int tiles[SLOTS][32 * TPITCH];
  const int ti = blockIdx.x * SLOTS + threadIdx.x / SLOT;
  intra_chunk<Pix, HBD>(edge, tiles, tasks, first + ti, ti < count, coeffs, f);
}

// One workgroup per island, walked through the frame in memory: the island's waves in order, 8 blocks at a time,
// with a workgroup barrier between waves.  __syncthreads() orders the global stores of one wave before the edge
// loads of the next for the threads of this workgroup (same CU, same L1).  For islands that do not fit the LDS
// window.
template <typename Pix, bool HBD>
__device__ __forceinline__ void island_mem_body(IslandMemLds &M, const vp9hip_intra_task *__restrict__ tasks,
                                                const vp9hip_intra_island &isl, const int32_t *__restrict__ wave_off,
                                                const txfm::Coefs &coeffs, const FrameDev &f) {
  // task j of a chunk goes to wavefront j % 4 (see island_lds_body)
  const int slot = ((threadIdx.x / SLOT) & 1) * (SLOTS / 2) + (threadIdx.x / SLOT) / 2;
  const int32_t *wo = wave_off + isl.wave_off_start;
  const int nw = (int)isl.n_waves;
  for (int w = 0; w < nw; ++w) {
    const int begin = wo[w], end = wo[w + 1];
    for (int base = begin; base < end; base += SLOTS) {
      const int ti = base + slot;
      intra_chunk<Pix, HBD>(M.edge, M.tiles, tasks, isl.task_start + ti, ti < end, coeffs, f);
    }
    __syncthreads();
  }
}

union IslandAnyLds {
  IslandLds lds;
  IslandMemLds mem;
};

template <typename Pix, bool HBD>
__global__ __launch_bounds__(256) void intra_island_kernel(const vp9hip_intra_task *__restrict__ tasks,
                                                           const vp9hip_intra_island *__restrict__ islands,
                                                           const int32_t *__restrict__ wave_off,
                                                           txfm::Coefs coeffs, FrameDev f) {
  __shared__ IslandAnyLds S;
  if (threadIdx.x == 0) VP9HIP_STAMP_SLOT((int)blockIdx.x);
  const vp9hip_intra_island isl = islands[blockIdx.x];
  if (island_lds_body<Pix, HBD>(S.lds, tasks, isl, wave_off, coeffs, f, nullptr, 0)) return;
  __syncthreads();
  island_mem_body<Pix, HBD>(S.mem, tasks, isl, wave_off, coeffs, f);
}

}  // namespace

#ifndef VP9HIP_INTRA_DEVICE_ONLY  // lf_kernels.hip includes this file for the device code above (fused walk + filter)
extern "C" int vp9hip_intra_pred_islands(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                                         const vp9hip_intra_island *d_islands, int n_islands,
                                         const int32_t *d_wave_off, const int32_t *d_coeffs,
                                         const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_tasks || !d_islands || n_islands < 0 || !d_wave_off || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_islands: bad argument");
  if (n_islands == 0) return VP9HIP_OK;
  const FrameDev f = to_dev(frame);
  const txfm::Coefs cf = { d_coeffs, ctx->coeff16 };
  if (frame->hbd)
    hipLaunchKernelGGL((intra_island_kernel<uint16_t, true>), dim3(n_islands), dim3(256), 0, ctx->stream, d_tasks, d_islands,
                       d_wave_off, cf, f);
  else
    hipLaunchKernelGGL((intra_island_kernel<uint8_t, false>), dim3(n_islands), dim3(256), 0, ctx->stream, d_tasks, d_islands,
                       d_wave_off, cf, f);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}

extern "C" int vp9hip_intra_pred_waves(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks, const int32_t *wave_start,
                                       int n_waves, const int32_t *d_coeffs, const vp9hip_frame *frame) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  if (!d_tasks || !wave_start || n_waves < 0 || !frame_ok(frame))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_waves: bad argument");
  const FrameDev f = to_dev(frame);
  const txfm::Coefs cf = { d_coeffs, ctx->coeff16 };
  for (int w = 0; w < n_waves; ++w) {
    const int first = wave_start[w], count = wave_start[w + 1] - first;
    if (count < 0) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_pred_waves: wave_start not monotonic");
    if (count == 0) continue;
    const int grid = (count + SLOTS - 1) / SLOTS;
    if (frame->hbd)
      hipLaunchKernelGGL((intra_wave_kernel<uint16_t, true>), dim3(grid), dim3(256), 0, ctx->stream, d_tasks, first,
                         count, cf, f);
    else
      hipLaunchKernelGGL((intra_wave_kernel<uint8_t, false>), dim3(grid), dim3(256), 0, ctx->stream, d_tasks, first,
                         count, cf, f);
  }
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
#endif  // !VP9HIP_INTRA_DEVICE_ONLY
