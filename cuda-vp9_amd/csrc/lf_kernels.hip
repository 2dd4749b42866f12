// lf_kernels.hip — loop filter of a whole frame (SURVEY §8 a11–a12).
//
// Ordering.  libvpx filters superblocks in raster order, per superblock all vertical edges
// then all horizontal ones (vp9_loopfilter.c:1424-1469, 1241-1422).  Superblock (r,c) touches
// columns [64c-8, 64c+63] (vertical pass, rows of SB row r) and rows [64r-8, 64r+63]
// (horizontal pass, columns of SB column c), so it must run after (r,c-1) and (r-1,c+1) —
// libvpx's own row-MT sync rule (vp9_thread_common.c:38-55).  All superblocks with the same
// t = c + 2r are independent; here ONE launch walks the rows (lf_rows2_kernel): a workgroup per
// (superblock row, plane), rows pipelined behind each other.
//
// Inside a superblock plane the 64 (32) pixel rows are independent in the vertical pass and
// the 64 (32) columns in the horizontal pass: lane = row, then lane = column.  Each lane walks
// its line through the 8-pixel positions in order: edge filter (16/8/4 wide, chosen by the
// LOOP_FILTER_MASK bits) at position c, then the interior 4x4 edge at c+4 — the per-line order
// of filter_selectively_vert_row2 / filter_selectively_horiz (vp9_loopfilter.c:297-375,
// 453-544), including their quirk that a 16-wide "dual" call applies the first segment's
// thresholds to both segments.  The tile (72x72 samples incl. the 8 columns/rows of the
// neighbours it modifies) is staged in LDS.
//
// Algorithmic bytes per frame: 2 * P * bps (every sample read and written once) + 160 * #SB.
#include <stdlib.h>

#include "vp9hip_internal.h"

// the island walk's device code, for the fused walk + filter kernel below
#define VP9HIP_INTRA_DEVICE_ONLY
#include "intra_kernels.hip"
#undef VP9HIP_INTRA_DEVICE_ONLY

namespace {

struct LfThreshDev {
  uint8_t mblim[64], lim[64], hev_thr[64];
};

template <typename Pix>
struct TileCfg {
  // LDS row pitch in samples: odd number of dwords per row for both sample widths, so that the
  // vertical pass (lane = row) walks its row without bank conflicts
  static constexpr int TP = sizeof(Pix) == 1 ? 76 : 74;
};

__device__ __forceinline__ int iabsd(int a, int b) {
  // samples are < 2^16: one v_sad_u16 gives |a - b|
  return (int)__builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, 0u);
}
__device__ __forceinline__ int sclamp(int t, int lo, int hi) { return t < lo ? lo : (t > hi ? hi : t); }

// Filter one line across one edge held in registers: w[0..7] = p7..p0, w[8..15] = q0..q7
// (only w[4..11] are touched unless kind == 16).  kind 0 = no filter here.
// vpx_dsp/loopfilter.c: filter_mask :33, flat_mask4 :49, flat_mask5 :62, hev_mask :72,
// filter4 :76, filter8 :162, filter16 :235; highbd forms :359-447 (thresholds << (bd-8)).
// WIDE = false: the interior 4x4 edge, whose kind is 0 or 4 — no flat tests, no wide forms in its code.
template <bool WIDE>
__device__ __forceinline__ void filter_window(int *w, int q, int kind, unsigned thr3, int sh) {
  // w + q points at q0 (q = 8 for the block edge, 12 for the interior 4x4 edge)
  // Interior edges are often absent on every line: their evaluation has an early out.  For block edges "no
  // edge on this line" is one more term of the mask — an early out of its own (execution-mask region, or
  // a scalar branch on a ballot) measured slower on partitions with small and with large blocks.
  if (!WIDE && kind == 0) return;
  const int blim = (int)(thr3 & 0xff) << sh, lim = (int)((thr3 >> 8) & 0xff) << sh;
  const int thr = (int)((thr3 >> 16) & 0xff) << sh, one = 1 << sh;
  const int p3 = w[q - 4], p2 = w[q - 3], p1 = w[q - 2], p0 = w[q - 1];
  const int q0 = w[q], q1 = w[q + 1], q2 = w[q + 2], q3 = w[q + 3];
  const int d10 = iabsd(p1, p0), e10 = iabsd(q1, q0);
  const int m = max(max(max(iabsd(p3, p2), iabsd(p2, p1)), max(d10, e10)), max(iabsd(q2, q1), iabsd(q3, q2)));
  // '&' not '&&': one execution-mask region instead of two nested ones (568 -> 550 us)
  const bool mask = (kind != 0) & (m <= lim) & (iabsd(p0, q0) * 2 + (iabsd(p1, q1) >> 1) <= blim);
  if (!mask) return;  // every filter form leaves the samples unchanged when the mask is off
  bool flat = false;
  if (WIDE)  // evaluated for every lane of the mask region: a region of its own for kind >= 8 costs more than it saves
    flat = (kind >= 8) &
           (max(max(max(d10, e10), max(iabsd(p2, p0), iabsd(q2, q0))), max(iabsd(p3, p0), iabsd(q3, q0))) <= one);
  if (WIDE && flat) {
    {
      const int p4 = w[q - 5], p5 = w[q - 6], p6 = w[q - 7], p7 = w[q - 8];
      const int q4 = w[q + 4], q5 = w[q + 5], q6 = w[q + 6], q7 = w[q + 7];
      const bool flat2 = (kind == 16) &
                         (max(max(max(iabsd(p4, p0), iabsd(q4, q0)), max(iabsd(p5, p0), iabsd(q5, q0))),
                              max(max(iabsd(p6, p0), iabsd(q6, q0)), max(iabsd(p7, p0), iabsd(q7, q0)))) <= one);
      if (flat2) {
        // 15-tap [1 1 1 1 1 1 1 2 1 1 1 1 1 1 1], replication at p7 / q7: sliding sum
        int s = p7 * 7 + p6 * 2 + p5 + p4 + p3 + p2 + p1 + p0 + q0 + 8;
        w[q - 7] = s >> 4;
        s += q1 - p7 + p5 - p6; w[q - 6] = s >> 4;
        s += q2 - p7 + p4 - p5; w[q - 5] = s >> 4;
        s += q3 - p7 + p3 - p4; w[q - 4] = s >> 4;
        s += q4 - p7 + p2 - p3; w[q - 3] = s >> 4;
        s += q5 - p7 + p1 - p2; w[q - 2] = s >> 4;
        s += q6 - p7 + p0 - p1; w[q - 1] = s >> 4;
        s += q7 - p7 + q0 - p0; w[q] = s >> 4;
        s += q7 - p6 + q1 - q0; w[q + 1] = s >> 4;
        s += q7 - p5 + q2 - q1; w[q + 2] = s >> 4;
        s += q7 - p4 + q3 - q2; w[q + 3] = s >> 4;
        s += q7 - p3 + q4 - q3; w[q + 4] = s >> 4;
        s += q7 - p2 + q5 - q4; w[q + 5] = s >> 4;
        s += q7 - p1 + q6 - q5; w[q + 6] = s >> 4;
        return;
      }
    }
    // 7-tap [1 1 1 2 1 1 1], replication at p3 / q3
    int s = p3 * 3 + p2 * 2 + p1 + p0 + q0 + 4;
    w[q - 3] = s >> 3;
    s += q1 - p3 + p1 - p2; w[q - 2] = s >> 3;
    s += q2 - p3 + p0 - p1; w[q - 1] = s >> 3;
    s += q3 - p3 + q0 - p0; w[q] = s >> 3;
    s += q3 - p2 + q1 - q0; w[q + 1] = s >> 3;
    s += q3 - p1 + q2 - q1; w[q + 2] = s >> 3;
    return;
  }
  // narrow filter (filter4)
  // libvpx works on samples offset by -0x80 << sh (signed chars): the offsets cancel in the differences,
  // and clamp(x - off, -off, off - 1) + off == clamp(x, 0, 2 * off - 1) for the four results
  const int off = 0x80 << sh, lo = -off, hi = off - 1, maxv = 2 * off - 1;
  const int hev = ((d10 > thr) | (e10 > thr)) ? -1 : 0;
  int f = sclamp(p1 - q1, lo, hi) & hev;
  f = sclamp(f + __mul24(q0 - p0, 3), lo, hi);  // v_mad_i32_i24 (3 * x + f selected the half-rate v_mad_u64_u32)
  const int f1 = sclamp(f + 4, lo, hi) >> 3, f2 = sclamp(f + 3, lo, hi) >> 3;
  w[q] = sclamp(q0 - f1, 0, maxv);
  w[q - 1] = sclamp(p0 + f2, 0, maxv);
  f = ((f1 + 1) >> 1) & ~hev;
  w[q + 1] = sclamp(q1 - f, 0, maxv);
  w[q - 2] = sclamp(p1 + f, 0, maxv);
}


// ---- per-mask-bit filter controls (lane = mask bit) --------------------------------------------
template <int N>
__device__ __forceinline__ void lf_controls(unsigned *ctl, const vp9hip_lfm &m, int pl, int mi_row, int rows_mi,
                                            int mi_rows, const unsigned *thr_tab /* LDS: per level, mblim | lim << 8 | hev_thr << 16 */) {
  const int lane = threadIdx.x & 63;
  constexpr int ncol = N / 8;
  constexpr int nbits = ncol * ncol;
  unsigned *vE = ctl, *vI = ctl + 64, *hE = ctl + 128, *hI = ctl + 192;
  // vp9_loopfilter.c:297-375, 453-544
  if (lane < nbits) {
    uint64_t l16, l8, l4, a16, a8, a4, mint;
    int level, level_left = 0, level_up = 0;
    const int mr = lane / ncol, c = lane - mr * ncol;
    // N == 64: the luma masks — for the luma plane, and for the chroma planes of a 4:4:4 frame, which libvpx
    // filters with vp9_filter_block_plane_ss00 and the same LOOP_FILTER_MASK (LF_PATH_444,
    // libvpx/vp9/common/vp9_loopfilter.c:1433-1434, 1456-1458); N == 32: the uv masks of 4:2:0
    if (N == 64) {
      l16 = m.left_y[2]; l8 = m.left_y[1]; l4 = m.left_y[0];
      a16 = m.above_y[2]; a8 = m.above_y[1]; a4 = m.above_y[0];
      mint = m.int_4x4_y;
      level = m.lfl_y[lane];
      if (c > 0) level_left = m.lfl_y[lane - 1];
      if (mr > 0) level_up = m.lfl_y[lane - ncol];
    } else {
      l16 = m.left_uv[2]; l8 = m.left_uv[1]; l4 = m.left_uv[0];
      a16 = m.above_uv[2]; a8 = m.above_uv[1]; a4 = m.above_uv[0];
      mint = m.int_4x4_uv;
      // lfl_uv[(r>>1)*4 + c] = lfl_y[r*8 + 2c] for even mi rows r (vp9_loopfilter.c:1344-1348)
      level = m.lfl_y[mr * 16 + c * 2];
      if (c > 0) level_left = m.lfl_y[mr * 16 + (c - 1) * 2];
      if (mr > 0) level_up = m.lfl_y[(mr - 1) * 16 + c * 2];
    }
    // (a table in LDS: indexed by a per-lane level, the thresholds in the kernel's arguments were three single-byte
    // loads from memory behind the load of the level — two dependent round trips in wave 1's work beside the
    // horizontal pass, which frames of real streams wait for)
    auto thr3 = [&](int lv) { return thr_tab[lv & 63]; };
    const int bit = lane;
    // vertical pass
    {
      const unsigned kind = ((l16 >> bit) & 1) ? 16u : ((l8 >> bit) & 1) ? 8u : ((l4 >> bit) & 1) ? 4u : 0u;
      int lv = level;
      // vpx_lpf_vertical_16_dual applies the even mask row's thresholds to both rows of a pair
      if (kind == 16 && (mr & 1) && ((l16 >> (bit - ncol)) & 1)) lv = level_up;
      vE[bit] = (kind << 24) | thr3(lv);
      vI[bit] = (((mint >> bit) & 1) ? (4u << 24) : 0u) | thr3(level);
    }
    // horizontal pass
    {
      const bool edge_ok = !(mi_row == 0 && mr == 0);
      const unsigned rowmask16 = edge_ok ? (unsigned)((a16 >> (mr * ncol)) & ((1u << ncol) - 1)) : 0u;
      unsigned kind = 0;
      int lv = level;
      if ((rowmask16 >> c) & 1) {
        kind = 16;
        // the second segment of a 16-wide "dual" pair reuses the first's thresholds
        // (vp9_loopfilter.c:466-469); pairs form from the start of a run of 16-wide segments
        int run = 0;
        for (int k = c - 1; k >= 0 && ((rowmask16 >> k) & 1); --k) ++run;
        if (run & 1) lv = level_left;
      } else if (edge_ok && ((a8 >> bit) & 1)) {
        kind = 8;
      } else if (edge_ok && ((a4 >> bit) & 1)) {
        kind = 4;
      }
      int skip_int = -1;  // skip_border_4x4_r (vp9_loopfilter.c:1385-1387)
      if (N == 32)
        for (int r = 0; r < rows_mi; r += 2)
          if (mi_row + r == mi_rows - 1) skip_int = r >> 1;
      // the 16-wide branch of filter_selectively_horiz never filters the interior edge (:465-538)
      const bool bi = kind != 16 && (mr != skip_int) && ((mint >> bit) & 1);
      hE[bit] = (kind << 24) | thr3(lv);
      hI[bit] = (bi ? (4u << 24) : 0u) | thr3(level);
    }
  }
}

// ---- vertical edges: lane = sample row; a 16-sample window slides along the row
template <typename Pix, int N>
__device__ __forceinline__ void lf_pass_v(Pix *tile, const unsigned *ctl, int y0, int ph, int mrows, int sh,
                                          volatile unsigned *strip_flag = nullptr, unsigned strip_val = 0) {
  constexpr int TP = TileCfg<Pix>::TP;
  constexpr int n = N;
  constexpr int ncol = N / 8;
  const int lane = threadIdx.x & 63;
  const unsigned *vE = ctl, *vI = ctl + 64;
  if (lane < n && y0 + lane < ph && (lane >> 3) < mrows) {
    const int mr = lane >> 3;
    // (the compiler reads and writes the row's samples a dword at a time and unpacks / packs them; one LDS operation
    // per sample instead — no packing in the chain — measured 1669 against 2133 frames/s)
    Pix *row = tile + (8 + lane) * TP;
    int w[16], nxt[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = row[k];
    unsigned cE = vE[mr * ncol], cI = vI[mr * ncol];
#ifndef LF_V_UNROLL
#define LF_V_UNROLL 8  // whole superblock: the compiler renames the register window instead of moving it (V pass trip by trip 2042, two per trip 2100, whole 2132 frames/s)
#endif
#pragma unroll LF_V_UNROLL
    for (int c = 0; c < ncol; ++c) {
      // prefetch the next position's samples and controls while this one is filtered
      const int cn = c + 1 < ncol ? c + 1 : c;
#pragma unroll
      for (int k = 0; k < 8; ++k) nxt[k] = row[8 + cn * 8 + k];
      const unsigned nE = vE[mr * ncol + cn], nI = vI[mr * ncol + cn];
      filter_window<true>(w, 8, cE >> 24, cE, sh);
      filter_window<false>(w, 12, cI >> 24, cI, sh);
#pragma unroll
      for (int k = 0; k < 8; ++k) row[c * 8 + k] = (Pix)w[k];
      if (c == 0 && strip_flag != nullptr) {
        // the left strip (the last 8 columns of the superblock before) is final after the first position:
        // tell the publisher wave now, so that the row below gets the corner ~7/8 of a pass earlier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *strip_flag = strip_val;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w[k] = w[8 + k];
        w[8 + k] = nxt[k];
      }
      cE = nE;
      cI = nI;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) row[ncol * 8 + k] = (Pix)w[k];
  }
}

// ---- horizontal edges: lane = sample column; the window slides down the column
template <typename Pix, int N>
__device__ __forceinline__ void lf_pass_h(Pix *tile, const unsigned *ctl, int x0, int pw, int mrows, int sh) {
  constexpr int TP = TileCfg<Pix>::TP;
  constexpr int n = N;
  constexpr int ncol = N / 8;
  const int lane = threadIdx.x & 63;
  const unsigned *hE = ctl + 128, *hI = ctl + 192;
  if (lane < n && x0 + lane < pw) {
    const int c = lane >> 3;  // mask column
    Pix *col = tile + 8 + lane;
    int w[16], nxt[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = col[k * TP];
    unsigned cE = hE[c], cI = hI[c];
    // two positions per trip: the compiler renames the window instead of moving it (587 -> 568 us; four
    // per trip measures the same)
#ifndef LF_H_UNROLL
#define LF_H_UNROLL 2
#endif
#pragma unroll LF_H_UNROLL
    for (int mr = 0; mr < mrows; ++mr) {
      const int mn = mr + 1 < mrows ? mr + 1 : mr;
#pragma unroll
      for (int k = 0; k < 8; ++k) nxt[k] = col[(8 + mn * 8 + k) * TP];
      const unsigned nE = hE[mn * ncol + c], nI = hI[mn * ncol + c];
      filter_window<true>(w, 8, cE >> 24, cE, sh);
      filter_window<false>(w, 12, cI >> 24, cI, sh);
#pragma unroll
      for (int k = 0; k < 8; ++k) col[(mr * 8 + k) * TP] = (Pix)w[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w[k] = w[8 + k];
        w[8 + k] = nxt[k];
      }
      cE = nE;
      cI = nI;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) col[(mrows * 8 + k) * TP] = (Pix)w[k];
  }
}

// =============================================================================================
// Row-walking form: ONE launch.  Workgroup (r, plane) owns superblock row r and walks it left to
// right, which is the order the vertical-edge chain imposes anyway; it may start superblock c
// once row r-1 has finished superblock c+1 (the (r-1,c+1) dependency above).  Rows publish their
// progress through a counter per (plane, row); the only pixels that cross workgroups are the
// bottom 8 rows of a superblock row, which row r+1 reads as its "rows above".
//
// Hand-off protocol (MI355X_MICROARCH.md "Valid forms", cdna_hip_programming.md Guideline 16, R1
// with a counter): the handed-off rows are stored write-through (agent-scope relaxed atomic
// stores = global_store sc1), the single storing wave drains (s_waitcnt vmcnt(0)), then one lane
// stores the counter with an agent-scope atomic; the consumer polls the counter with agent-scope
// relaxed loads (s_sleep back-off, BOUNDED spin) and reads the rows with agent-scope relaxed
// loads (sc1) only after the poll matched.  Everything else a workgroup touches is private to it
// for the duration of the launch (plain loads/stores).  All sb_rows x planes workgroups are
// co-resident (69 at 1440p), and row r only ever waits on row r-1, so there is no cycle.
// The left 8 columns of a tile never leave LDS between two superblocks of a row.
// Every wait is bounded all the same: a row that gives up raises the context's error flag (vp9hip_sync reports it).
// The bound is TIME (the 100 MHz constant clock, looked at every 1024 polls), and a wait for another ROW gets twice
// the time of a wait for islands or for the workgroup's own waves: when something upstream stalls, the wait at the
// root of the chain gives up first and is the one the error record names.
constexpr long long LF_WAIT_TICKS = 30 * 1000 * 1000;  // 0.3 s
struct LfWaitClock {
  int polls = 0;
  long long t0 = 0;
  __device__ __forceinline__ bool expired(long long limit) {
    if ((++polls & 1023) != 0) return false;
    const long long now = (long long)wall_clock64();
    if (t0 == 0) t0 = now;
    return now - t0 > limit;
  }
};

// err[0]: bit 0 a filter row gave up waiting, bit 1 an island did not fit; err[4], err[5]: the launch's ticket counter
// at that moment and its grid size; err[1..3]: what the FIRST wait that gave
// up was waiting for — kind (1 island counter, 2 rows of the row above, 3 the workgroup's own filtering wave) |
// plane << 8 | row << 16, superblock column, and the value it saw.  vp9hip_sync puts them into the error text.
__device__ __forceinline__ void lf_give_up(int *err, int kind, int pl, int sr, int col, int seen, const int *ticket) {
  if (atomicOr(err, 1) == 0) {
    err[1] = kind | (pl << 8) | (sr << 16);
    err[2] = col;
    err[3] = seen;
    err[4] = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // workgroups that have started
    err[5] = (int)gridDim.x;
  }
}

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned *p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Hand-off granule (MI355X_MICROARCH.md, persistent kernels: "handoff-1to1", data-tagged granules): one naturally
// aligned 8-byte {sample dword, tag} written by ONE sc1 store and polled with sc1 loads; the tag is the launch's
// generation number, so a granule of an earlier launch never passes for this one.  One memory round trip per
// hand-off instead of three (write-through stores drained, counter, poll, sc1 loads).
typedef unsigned long long lf_granule;
__device__ __forceinline__ lf_granule ld_granule(const lf_granule *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(lf_granule *p, unsigned data, unsigned tag) {
  __hip_atomic_store(p, ((lf_granule)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Four-wave form of the row walk: wave 0 filters, wave 1 moves data in, wave 2 hands rows to the
// row below, wave 3 writes the finished superblock back.  Per superblock c:
//   phase A   wave 0: vertical pass of c, LDS flag.
//             wave 1: waits for the row above and brings its bottom 8 rows into tile rows 0..7
//             (which the vertical pass does not touch).  wave 3: bulk write-back of c-1.
//             wave 2: on the flag, the 8x8 corner that pass completed (bottom rows of the PREVIOUS
//             superblock's last 8 columns) goes out write-through; drained; v-progress = c+1.
//   phase B   wave 0: horizontal pass of c, LDS flag, right strip -> left strip of the other buffer.
//             wave 1: interior + controls of c+1 into the other buffer.
//             wave 2: on the flag, the hand-off rows out write-through; drained; h-progress = c+1.
// Wave 0 never waits for a store to drain (two drains per superblock were ~15 % of its time).
// The rows above superblock c are final once the row above has done the horizontal pass of ITS
// superblock c (columns 0..55 of c) and the vertical pass of its superblock c+1 (the last 8
// columns: that pass's first edge reaches 8 samples back) — (r-1,c+1)'s horizontal pass never
// touches columns left of 64(c+1).  Waiting only for that, and only before the horizontal pass,
// halves the lag between superblock rows from two superblock steps to one
// (critical path cols + rows instead of cols + 2*rows).
template <typename Pix, int N, int SH>
__device__ __forceinline__ void lf_row2_body(Pix *tiles, unsigned *ctls, const vp9hip_lfm *__restrict__ lfms,
                                             int sb_cols, int sr, int pl, const LfThreshDev &th, const FrameDev &f,
                                             int mi_rows, int *err, volatile unsigned *flags, const unsigned *thr_tab,
                                             const int *gate_done, const int *gate_expected, int sb_rows,
                                             lf_granule *hand_base, unsigned gen, const int *ticket) {
  constexpr int TP = TileCfg<Pix>::TP;
  constexpr int PPD = 4 / sizeof(Pix);
  constexpr int n = N;
  constexpr int TPD = TP / PPD;
  constexpr int DPR = n / PPD;
  constexpr int KI = (n * DPR + 63) / 64;
  constexpr int KA = (8 * DPR + 63) / 64;
  constexpr int TILE = 72 * TP;  // samples per tile buffer
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // sample shift of the thresholds: a compile-time constant per bit depth (SH >= 0), so that the
  // threshold unpacking and the clamps of the narrow filter fold (v_med3, byte-select operands)
  const int sh = SH >= 0 ? SH : f.bit_depth - 8;
  Pix *plane = (Pix *)f.plane[pl];
  const int stride = f.stride[pl];
  const int pw = f.awidth[pl], ph = f.aheight[pl];
  const int y0 = sr * n;
  const int mi_row = sr * 8;
  const int rows_mi = min(8, mi_rows - mi_row);
  const int mrows = N == 32 ? ((rows_mi + 1) >> 1) : rows_mi;
  const int ncols = min(sb_cols, (pw + n - 1) / n);  // superblocks of this plane row
  bool dead = false;
  // this plane's hand-off granules: 8 rows per superblock row, a granule per sample dword
  const int hpitch = stride / PPD;
  lf_granule *hand = hand_base;
  for (int q = 0; q < pl; ++q) hand += (size_t)sb_rows * 8 * (f.stride[q] / PPD);
  const bool bottom_row = sr == sb_rows - 1 || y0 + n >= ph;  // nobody below: the bottom rows go to the frame

  unsigned reg[KI];  // wave 1: interior of the next superblock between phase B and phase C

  auto load_interior = [&](int sc) {  // wave 1: global -> registers (private rows: plain loads)
    const int x0 = sc * n;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int i = lane + 64 * k;
      const int r = i / DPR, d = i - r * DPR;
      const int gx = x0 + d * PPD, gy = y0 + r;
      reg[k] = 0;
      if (i < n * DPR && gx < pw && gy < ph) reg[k] = *(const unsigned *)(plane + (size_t)gy * stride + gx);
    }
  };
  auto store_interior = [&](int sc) {  // wave 1: registers -> LDS tile + controls of superblock sc
    unsigned *t32 = (unsigned *)(tiles + (sc & 1) * TILE);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int i = lane + 64 * k;
      const int r = i / DPR, d = i - r * DPR;
      if (i < n * DPR) t32[(8 + r) * TPD + 8 / PPD + d] = reg[k];
    }
    lf_controls<N>(ctls + (sc & 1) * 256, lfms[sr * sb_cols + sc], pl, mi_row, rows_mi, mi_rows, thr_tab);
  };
  // Running beside the intra island walk (vp9hip_intra_islands_lf): superblock (sr, c) may be loaded
  // and filtered once every island touching superblocks (sr..sr+1, c-1..c+1) is done — an unfinished
  // island there still reads samples this superblock's passes change (its left / above / above-left /
  // above-right neighbours), or has not written the samples yet.  Columns are checked as the walk
  // reaches them.  Consumer half of the hand-off recipe: poll at agent scope (bounded), then an
  // agent-scope acquire before the plain loads (another workgroup of this XCD may have pulled a line
  // into L2 while an island elsewhere was still writing into it).
  auto gate_col = [&](int col) {  // wave 1
    if (gate_done == nullptr || col >= sb_cols) return;
    for (int r = sr; r <= sr + 1 && r < sb_rows; ++r) {
      const int need = gate_expected[r * sb_cols + col];
      LfWaitClock clk;
      while (!dead && __hip_atomic_load(&gate_done[r * sb_cols + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(4);
        if (clk.expired(LF_WAIT_TICKS)) {
          if (lane == 0)
            lf_give_up(err, 1, pl, r, col, __hip_atomic_load(&gate_done[r * sb_cols + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | (need << 16), ticket);
          dead = true;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  auto fetch_above = [&](int sc) {  // wave 1: bottom 8 rows of the row above -> tile rows 0..7
    unsigned *t32 = (unsigned *)(tiles + (sc & 1) * TILE);
    const int x0 = sc * n;
    unsigned above[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) above[k] = 0;
    if (sr > 0) {
      // the row above published them as tagged granules — columns 0..55 of this superblock after its horizontal
      // pass of superblock sc, the last 8 after the first position of its vertical pass of sc+1: poll until every
      // granule this wave needs carries this launch's tag
      LfWaitClock clk;
      for (;;) {
        bool ok = true;
        int have = 0;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
          const int i = lane + 64 * k;
          const int r = i / DPR, d = i - r * DPR;
          const int gx = x0 + d * PPD;
          if (i < 8 * DPR && gx < pw) {
            const lf_granule g = ld_granule(&hand[(size_t)((sr - 1) * 8 + r) * hpitch + gx / PPD]);
            above[k] = (unsigned)g;
            ok = ok && (unsigned)(g >> 32) == gen;
            have += (unsigned)(g >> 32) == gen;
          }
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0 || dead) break;
        __builtin_amdgcn_s_sleep(1);
        if (clk.expired(2 * LF_WAIT_TICKS)) {  // (lane 0 holds the leftmost granules: columns 0.. of the first row)
          if (lane == 0) lf_give_up(err, 2, pl, sr, sc, have | (KA << 16), ticket);
          dead = true;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      const int i = lane + 64 * k;
      const int r = i / DPR, d = i - r * DPR;
      if (i < 8 * DPR) t32[r * TPD + 8 / PPD + d] = above[k];
    }
  };
  // everything of superblock sc except the hand-off rows; `part` of 2: the upper / lower half of the tile rows
  // (two waves share it: with one, the filtering wave waited ~2 k cycles per step at the barrier behind its
  // vertical pass with 8-bit samples and ~9 k with 16-bit samples — twice the bytes)
  auto bulk_writeback = [&](int sc, int part) {
    const int x0 = sc * n;
    const unsigned *t32 = (const unsigned *)(tiles + (sc & 1) * TILE);
    const bool last = sc == ncols - 1;
    // tile rows 8 .. n-1 (this superblock row's own rows but the bottom 8), tile columns 0 .. n-1: 16-byte
    // pieces, one per lane — a dword per lane with a division per dword took the wave longer than the
    // filtering wave's vertical pass
    constexpr int PXC = 16 / (int)sizeof(Pix);      // samples per piece
    constexpr int cpr = n / PXC;                     // pieces per row (a power of two)
    constexpr int tot = (n - 8) * cpr, half = tot / 2;
    for (int ci = part * half + lane; ci < (part + 1) * half; ci += 64) {
      const int r = 8 + ci / cpr, c = ci % cpr;
      const int gx = x0 - 8 + c * PXC, gy = y0 - 8 + r;
      if (gy >= ph) continue;
      const unsigned *tp = t32 + r * TPD + c * 4;
      Pix *gp = plane + (size_t)gy * stride + gx;
      if (gx >= 0 && gx + PXC <= pw) {
        uint4 v;
        v.x = tp[0]; v.y = tp[1]; v.z = tp[2]; v.w = tp[3];
        __builtin_memcpy(gp, &v, 16);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (gx + k * PPD >= 0 && gx + k * PPD < pw) ((unsigned *)gp)[k] = tp[k];
      }
    }
    if (part == 1 && last) {  // the frame's last superblock of the row: its right strip goes out as well
      constexpr int ed = 8 / PPD;  // dwords per row
      for (int i = lane; i < (n - 8) * ed; i += 64) {
        const int r = 8 + i / ed, d = i % ed;
        const int gx = x0 - 8 + n + d * PPD, gy = y0 - 8 + r;
        if (gx < pw && gy < ph) *(unsigned *)(plane + (size_t)gy * stride + gx) = t32[r * TPD + n / PPD + d];
      }
    }
    if (part == 0 && sr > 0) {
      // tile rows 0..7 (the 8 rows above, final after this row's horizontal pass), tile columns 8 .. n+7; the row
      // above handed them over as granules and never wrote them to the frame; their left strip went out a step ago
      constexpr int ed = n / PPD;  // dwords per row
      for (int i = lane; i < 8 * ed; i += 64) {
        const int r = i / ed, d = 8 / PPD + i % ed;
        const int gx = x0 - 8 + d * PPD, gy = y0 - 8 + r;
        if (gx < pw) *(unsigned *)(plane + (size_t)gy * stride + gx) = t32[r * TPD + d];
      }
    }
  };
  // wave 0: tile rows n..n+7 (the bottom 8 rows of this superblock row), tile columns [c0, c1), out
  // write-through and drained
  auto handoff = [&](const unsigned *t32, int x0, int c0, int c1) {
    constexpr int wd = (n + 8) / PPD;
    for (int i = lane; i < 8 * wd; i += 64) {
      const int r = n + i / wd, d = i % wd;
      const int gx = x0 - 8 + d * PPD, gy = y0 - 8 + r;
      if (d * PPD < c0 || d * PPD >= c1 || gx < 0 || gx >= pw || gy >= ph) continue;
      if (bottom_row)
        *(unsigned *)(plane + (size_t)gy * stride + gx) = t32[r * TPD + d];
      else  // for the row below only, which writes the rows' final values to the frame after its horizontal pass
        st_granule(&hand[(size_t)(sr * 8 + (r - n)) * hpitch + gx / PPD], t32[r * TPD + d], gen);
    }
  };

  // wave 2 (publisher) sends the hand-off rows out and publishes the progress once they have drained —
  // wave 0 never waits for a store; the barriers order its LDS reads behind wave 0's passes.
  // the filtering wave is the critical path of the frame: when it shares a SIMD with waves of the
  // island walk running beside it, it issues first
  if (wave == 0) __builtin_amdgcn_s_setprio(3);
  // prologue: interior of superblock 0 into buffer 0
  if (wave == 1) {
    gate_col(0);
    gate_col(1);
    load_interior(0);
    store_interior(0);
  }
  if (threadIdx.x == 0) flags[0] = 0;
  __syncthreads();
  VP9HIP_STAMP(1);
#ifdef VP9HIP_STAMPS  // probe builds: where the filtering wave's steps go (sums over the row, 100 MHz ticks)
  long long acc_[4] = { 0, 0, 0, 0 }, tm_ = wall_clock64();
#define LF_ACC(k)                          \
  do {                                     \
    const long long n_ = wall_clock64();   \
    acc_[k] += n_ - tm_;                   \
    tm_ = n_;                              \
  } while (0)
#else
#define LF_ACC(k) do { } while (0)
#endif
  for (int sc = 0; sc < ncols; ++sc) {
    const int x0 = sc * n;
    const bool last = sc == ncols - 1;
    Pix *tile = tiles + (sc & 1) * TILE;
    unsigned *t32 = (unsigned *)tile;
    const unsigned *ctl = ctls + (sc & 1) * 256;
    // ---- phase A
    if (wave == 0) {
      lf_pass_v<Pix, N>(tile, ctl, y0, ph, mrows, sh, &flags[0], (unsigned)(sc + 1));
      if (lane == 0) flags[0] = sc + 1;  // (whatever the lanes of the pass did)
    } else if (wave == 1) {
      fetch_above(sc);
    } else if (wave == 3) {
      if (sc > 0) bulk_writeback(sc - 1, 0);
    } else {
      // publisher: neither barrier waits for its store drains (they cost the filtering wave ~1.5 k + ~1.9 k
      // cycles per step when it sent a pass's rows right behind that pass).  While superblock sc gets its
      // vertical pass: the bottom rows the horizontal pass of sc-1 completed (other tile buffer; that pass
      // ended before barrier B); then, as soon as the first position of the vertical pass is through (LDS
      // flag), the 8x8 corner it completed — same wave, drained in between, so the corner lands after the
      // rows of sc-1 it overwrites; then its share of the write-back of sc-1.
      if (sc > 0) {
        handoff((const unsigned *)(tiles + ((sc - 1) & 1) * TILE), x0 - n, 8, n);
      }
      for (LfWaitClock clk; flags[0] < (unsigned)(sc + 1);) {  // wave 0 of this workgroup: bounded all the same
        __builtin_amdgcn_s_sleep(1);
        if (clk.expired(LF_WAIT_TICKS)) {
          if (lane == 0) lf_give_up(err, 3, pl, sr, sc, (int)flags[0], ticket);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (sc > 0) handoff(t32, x0, 0, 8);
      if (sc > 0) bulk_writeback(sc - 1, 1);
    }
    LF_ACC(0);
    __syncthreads();
    LF_ACC(1);
    // ---- phase B + C: wave 0 filters and moves the strip, wave 1 prefetches, wave 2 publishes
    if (wave == 0) {
      lf_pass_h<Pix, N>(tile, ctl, x0, pw, mrows, sh);
      // the right strip becomes the left strip of the next superblock (other buffer; its rows
      // 8.. columns 0..7 are not touched by wave 1's interior store)
      if (!last && lane < n) {
        unsigned *nt32 = (unsigned *)(tiles + ((sc + 1) & 1) * TILE);
#pragma unroll
        for (int d = 0; d < 8 / PPD; ++d) nt32[(8 + lane) * TPD + d] = t32[(8 + lane) * TPD + n / PPD + d];
      }
    } else if (wave == 1) {
      if (!last) {
        gate_col(sc + 2);  // columns sc, sc+1 were checked on the way here
        load_interior(sc + 1);
        store_interior(sc + 1);
      }
    }
    LF_ACC(2);
    __syncthreads();
    LF_ACC(3);
  }
#ifdef VP9HIP_STAMPS
  if (threadIdx.x == 0 && s_stamp_slot < 4096)
    for (int k = 0; k < 4; ++k) g_stamps[s_stamp_slot * 8 + 2 + k] = acc_[k];
  if (threadIdx.x == 64 && s_stamp_slot < 4096) g_stamps[s_stamp_slot * 8 + 6] = acc_[2];  // wave 1's own phase B work
#endif
  if (wave == 2) {  // the last superblock's bottom rows, right strip included
    handoff((const unsigned *)(tiles + ((ncols - 1) & 1) * TILE), (ncols - 1) * n, 8, n + 8);
    bulk_writeback(ncols - 1, 1);
  }
  if (wave == 3) bulk_writeback(ncols - 1, 0);
}

// LDS of a filter row workgroup
template <typename Pix>
struct LfRowLds {
  __attribute__((aligned(16))) Pix tiles[2 * 72 * TileCfg<Pix>::TP];
  unsigned ctls[2 * 256];
  unsigned flags[2];
  unsigned thr[64];  // per filter level: mblim | lim << 8 | hev_thr << 16
};

template <typename Pix, int SH>
__device__ __forceinline__ void lf_row_entry(LfRowLds<Pix> &L, const vp9hip_lfm *__restrict__ lfms, int sb_cols, int sb_rows,
                                             const LfThreshDev &th, const FrameDev &f, int mi_rows, int *err,
                                             const int *gate_done, const int *gate_expected, lf_granule *hand, unsigned gen,
                                             int sr, int pl, const int *ticket) {
  if (threadIdx.x < 64) L.thr[threadIdx.x] = (unsigned)th.mblim[threadIdx.x] | ((unsigned)th.lim[threadIdx.x] << 8) | ((unsigned)th.hev_thr[threadIdx.x] << 16);
  __syncthreads();
  if (pl == 0 || f.awidth[pl] == f.awidth[0])
    lf_row2_body<Pix, 64, SH>(L.tiles, L.ctls, lfms, sb_cols, sr, pl, th, f, mi_rows, err, L.flags, L.thr, gate_done, gate_expected,
                              sb_rows, hand, gen, ticket);
  else
    lf_row2_body<Pix, 32, SH>(L.tiles, L.ctls, lfms, sb_cols, sr, pl, th, f, mi_rows, err, L.flags, L.thr, gate_done, gate_expected,
                              sb_rows, hand, gen, ticket);
}

// Forward progress.  A workgroup's place in the launch's ORDER is not its hardware index but a ticket it draws when it
// starts running (one atomic add on a per-launch counter): whoever holds a lower ticket is running or done.  The
// launches below are built so that a workgroup only ever waits for lower tickets — then every wait ends, whatever the
// hardware's dispatch order is (the workgroups of a grid are dealt round-robin to the 8 XCDs and every XCD starts its
// share as ITS slots allow: index order holds per XCD only, and rows resident on one XCD, waiting for islands queued
// on another that is full of rows waiting the other way round, were seen to give up with six decoders in flight) and
// whatever else shares the GPU: no workgroup waits for one that is not resident.
__device__ __forceinline__ int draw_ticket(int *ticket, int *ticket_next) {
  __shared__ int s_ticket;
  if (threadIdx.x == 0) {
    const int t = atomicAdd(ticket, 1);
    // the next launch's counter (launches of a context run one after the other).  An atomic at agent scope like the
    // draws themselves, and the two counters lie in cache lines of their own: a plain store would bring its line
    // into this XCD's L2, and draws of this XCD served from that copy would hand out tickets twice.
    if (t == 0) __hip_atomic_exchange(ticket_next, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ticket = t;
    VP9HIP_STAMP_SLOT(t);
  }
  __syncthreads();
  return s_ticket;
}

// The filter alone: ticket t is (row t / planes, plane t % planes).  Row r waits for row r-1 = a lower ticket.
template <typename Pix, int SH>
__global__ __launch_bounds__(256) void lf_rows2_kernel(const vp9hip_lfm *__restrict__ lfms, int sb_cols, int sb_rows,
                                                       int planes, LfThreshDev th, FrameDev f, int mi_rows, int *err,
                                                       lf_granule *hand, unsigned gen, int *ticket, int *ticket_next) {
  __shared__ LfRowLds<Pix> L;
  const int b = draw_ticket(ticket, ticket_next);
  const int sr = b / planes, pl = b % planes;
  lf_row_entry<Pix, SH>(L, lfms, sb_cols, sb_rows, th, f, mi_rows, err, nullptr, nullptr, hand, gen, sr, pl, ticket);
}

// Where the filter's rows sit among the islands in the fused launch's grid: the `planes` workgroups of row r
// start at index pos[r] + r * planes, i.e. behind pos[r] islands.
constexpr int LF_MAX_ROWS = 128;
struct RowPos {
  int pos[LF_MAX_ROWS];
};

template <typename Pix>
union WalkLfLds {
  IslandLds isl;
  LfRowLds<Pix> row;
};

// The island walk and the loop filter of a frame as ONE launch: a workgroup per island (walked in LDS) and a
// workgroup per (superblock row, plane) of the filter, the latter placed in the launch's order (tickets, above) right
// behind the last island they can ever wait for (RowPos).  Forward progress by construction: islands wait for
// nothing; a row waits for the row above (placed before it) and for island marks of superblock rows r, r + 1 (all in
// front of it).  No second stream, no fork / join events, no residual pre-pass: every dependency packet between two
// kernels cost the command processor several microseconds (rocprofv3 trace of bench.py).
// While the frame runs, the first row's workgroup also zero-fills the island counters of the NEXT launch.
template <typename Pix, int SH>
#ifndef WALK_LF_WAVES
#define WALK_LF_WAVES 2
#endif
__global__ __launch_bounds__(256, WALK_LF_WAVES) void walk_lf_kernel(const vp9hip_lfm *__restrict__ lfms, int sb_cols, int sb_rows, int planes,
                                                      LfThreshDev th, FrameDev f, int mi_rows, int *err, int *gate_done,
                                                      int *gate_next, int n_gate, const int *gate_expected,
                                                      const vp9hip_intra_task *__restrict__ tasks,
                                                      const vp9hip_intra_island *__restrict__ islands,
                                                      const int32_t *__restrict__ wave_off, txfm::Coefs coeffs,
                                                      lf_granule *hand, unsigned gen, RowPos rp, int *ticket,
                                                      int *ticket_next) {
  __shared__ WalkLfLds<Pix> S;
  const int b = draw_ticket(ticket, ticket_next);
  int k = 0;  // row groups that start at or before b: pos[r] + r * planes grows with r
  for (int hi = sb_rows; k < hi;) {
    const int mid = (k + hi) >> 1;
    if (rp.pos[mid] + mid * planes <= b)
      k = mid + 1;
    else
      hi = mid;
  }
  const int start = k > 0 ? rp.pos[k - 1] + (k - 1) * planes : 0;
  if (k > 0 && b < start + planes) {
    const int sr = k - 1, pl = b - start;
    if (sr == 0 && pl == 0)  // (write-through stores at agent scope, like every other access to the counters)
      for (int i = (int)threadIdx.x; i < n_gate; i += 256) __hip_atomic_store(&gate_next[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    VP9HIP_STAMP(0);
    lf_row_entry<Pix, SH>(S.row, lfms, sb_cols, sb_rows, th, f, mi_rows, err, gate_done, gate_expected, hand, gen, sr, pl, ticket);
    VP9HIP_STAMP(7);
    return;
  }
  const vp9hip_intra_island isl = islands[b - k * planes];
  if (!island_lds_body<Pix, sizeof(Pix) == 2>(S.isl, tasks, isl, wave_off, coeffs, f, gate_done, sb_cols)) {
    // not an island for this launch (vp9hip.h: VP9HIP_ISLAND_FITS): its marks never come, the rows around it give
    // up after their bounded wait; say why
    if (threadIdx.x == 0) atomicOr(err, 2);
  }
}

}  // namespace

#ifdef VP9HIP_STAMPS
// probe builds only: the stamps of the last fused launch (8 per workgroup; islands 0..6, filter rows 0 and 7)
extern "C" int vp9hip_debug_stamps(long long *out, int n_workgroups) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 8 * (size_t)n_workgroups) == hipSuccess ? 0 : -1;
}
#endif

// The hand-off granules of the row-walking filter (8 rows per superblock row and plane, 8 bytes per sample
// dword) and the generation number that tags this launch's granules.
static int lf_handoff_buffer(vp9hip_ctx *ctx, const vp9hip_frame *frame, int sb_rows, unsigned *gen) {
  const int ppd = frame->hbd ? 2 : 4;
  size_t need = 0;
  for (int p = 0; p < 3; ++p)
    if (frame->plane[p]) need += (size_t)sb_rows * 8 * (size_t)(frame->stride[p] / ppd) * sizeof(lf_granule);
  if (need > ctx->lf_hand_bytes || ctx->lf_gen == 0xffffffffu) {
    VP9HIP_CHECK(ctx, hipDeviceSynchronize());  // earlier launches may still use the old buffer / old tags
    if (need > ctx->lf_hand_bytes) {
      if (ctx->lf_hand) VP9HIP_CHECK(ctx, hipFree(ctx->lf_hand));
      ctx->lf_hand = nullptr;
      ctx->lf_hand_bytes = 0;
      VP9HIP_CHECK(ctx, hipMalloc(&ctx->lf_hand, need + need / 4));
      ctx->lf_hand_bytes = need + need / 4;
    }
    VP9HIP_CHECK(ctx, hipMemsetAsync(ctx->lf_hand, 0, ctx->lf_hand_bytes, ctx->stream));  // (in the launch's own stream: the
    // context's stream does not synchronise with the null stream, and a fill that lands late wipes granules)
    ctx->lf_gen = 0;
  }
  *gen = ++ctx->lf_gen;
  return VP9HIP_OK;
}

// The ticket counters of two consecutive launches (see draw_ticket): a launch counts in one and clears the other.
static int lf_tickets(vp9hip_ctx *ctx, int **cur, int **nxt) {
  constexpr int LINE = 64;  // ints: 256 bytes apart
  if (!ctx->lf_ticket) {
    VP9HIP_CHECK(ctx, hipMalloc((void **)&ctx->lf_ticket, 2 * LINE * sizeof(int)));
    VP9HIP_CHECK(ctx, hipMemsetAsync(ctx->lf_ticket, 0, 2 * LINE * sizeof(int), ctx->stream));
    ctx->lf_ticket_parity = 0;
  }
  *cur = ctx->lf_ticket + ctx->lf_ticket_parity * LINE;
  *nxt = ctx->lf_ticket + (ctx->lf_ticket_parity ^ 1) * LINE;
  ctx->lf_ticket_parity ^= 1;
  return VP9HIP_OK;
}

// Argument checks + error flag + hand-off buffer shared by the two launches.
static int lf_prepare(vp9hip_ctx *ctx, const char *who, const vp9hip_lfm *d_lfm, int sb_rows, int sb_cols,
                      const vp9hip_lf_thresh *h_thresh, const vp9hip_frame *frame, int planes, unsigned *gen) {
  if (!d_lfm || sb_rows <= 0 || sb_cols <= 0 || !h_thresh || !frame_ok(frame) || (planes != 1 && planes != 3))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "%s: bad argument", who);
  if (sb_rows != (frame->aheight[0] + 63) / 64 || sb_cols != (frame->awidth[0] + 63) / 64)
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "%s: %dx%d superblocks do not cover a %dx%d frame", who, sb_cols, sb_rows,
                frame->awidth[0], frame->aheight[0]);
  const bool c420 = frame->awidth[1] * 2 == frame->awidth[0] && frame->aheight[1] * 2 == frame->aheight[0];
  const bool c444 = frame->awidth[1] == frame->awidth[0] && frame->aheight[1] == frame->aheight[0];
  if (planes == 3 && !c420 && !c444)
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "%s: chroma must be 4:2:0 or 4:4:4 (libvpx's LF_PATH_SLOW is not implemented)", who);
  // the error flag lives in an allocation of its own: it stays set until vp9hip_sync has reported it, however
  // many frames are enqueued behind the one that gave up
  if (!ctx->lf_err_flag) {
    VP9HIP_CHECK(ctx, hipMalloc((void **)&ctx->lf_err_flag, 8 * sizeof(int)));
    VP9HIP_CHECK(ctx, hipMemsetAsync(ctx->lf_err_flag, 0, 8 * sizeof(int), ctx->stream));
  }
  ctx->lf_err_armed = true;
  return lf_handoff_buffer(ctx, frame, sb_rows, gen);
}

extern "C" int vp9hip_loop_filter_frame(vp9hip_ctx *ctx, const vp9hip_lfm *d_lfm, int sb_rows, int sb_cols,
                                        const vp9hip_lf_thresh *h_thresh, const vp9hip_frame *frame, int planes) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));  // the caller's thread may be on another device
  unsigned gen = 0;
  int rc = lf_prepare(ctx, "vp9hip_loop_filter_frame", d_lfm, sb_rows, sb_cols, h_thresh, frame, planes, &gen);
  if (rc) return rc;
  LfThreshDev th;
  memcpy(&th, h_thresh, sizeof(th));
  const FrameDev f = to_dev(frame);
  int *ticket = nullptr, *ticket_next = nullptr;
  rc = lf_tickets(ctx, &ticket, &ticket_next);
  if (rc) return rc;
#define LF_ROWS2(PIX, SH)                                                                                               \
  hipLaunchKernelGGL((lf_rows2_kernel<PIX, SH>), dim3(sb_rows * planes), dim3(256), 0, ctx->stream, d_lfm, sb_cols, sb_rows, \
                     planes, th, f, frame->aheight[0] / 8, ctx->lf_err_flag, (lf_granule *)ctx->lf_hand, gen, ticket, ticket_next)
  if (!frame->hbd)
    LF_ROWS2(uint8_t, 0);
  else if (frame->bit_depth == 10)
    LF_ROWS2(uint16_t, 2);
  else if (frame->bit_depth == 12)
    LF_ROWS2(uint16_t, 4);
  else
    LF_ROWS2(uint16_t, 0);
#undef LF_ROWS2
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}

// Island counters of the fused launch: two sets in the context scratch, used in turn — a launch counts in one
// and zero-fills the other for the launch after it (no fill packet in the queue between two frames).
static int lf_gate_sets(vp9hip_ctx *ctx, int n_gate, int **cur, int **nxt) {
  const size_t set_ints = ((size_t)n_gate + 63) & ~(size_t)63;
  if (ctx->gate_n != n_gate) {  // another frame geometry: a launch only zero-fills the n_gate counters of its own
    int rc = vp9hip_ensure_scratch(ctx, 2 * set_ints * sizeof(int));  // (synchronises when it grows)
    if (rc) return rc;
    VP9HIP_CHECK(ctx, hipMemsetAsync(ctx->scratch, 0, 2 * set_ints * sizeof(int), ctx->stream));
    ctx->gate_n = n_gate;
    ctx->gate_parity = 0;
  }
  *cur = (int *)ctx->scratch + (size_t)ctx->gate_parity * set_ints;
  *nxt = (int *)ctx->scratch + (size_t)(ctx->gate_parity ^ 1) * set_ints;
  ctx->gate_parity ^= 1;
  return VP9HIP_OK;
}

extern "C" int vp9hip_intra_islands_lf(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                                       const vp9hip_intra_island *d_islands, int n_islands,
                                       const int32_t *d_wave_off, const int32_t *d_coeffs,
                                       const int32_t *d_sb_expected, const int32_t *h_row_pos, const vp9hip_lfm *d_lfm,
                                       int sb_rows, int sb_cols, const vp9hip_lf_thresh *h_thresh,
                                       const vp9hip_frame *frame, int planes) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n_islands == 0) return vp9hip_loop_filter_frame(ctx, d_lfm, sb_rows, sb_cols, h_thresh, frame, planes);
  if (!d_tasks || !d_islands || n_islands < 0 || !d_wave_off || !d_sb_expected)
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_islands_lf: bad argument");
  if (sb_rows > LF_MAX_ROWS || sb_cols > LF_MAX_ROWS)
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_islands_lf: frame too large (%d x %d superblocks)", sb_cols, sb_rows);
  unsigned gen = 0;
  int rc = lf_prepare(ctx, "vp9hip_intra_islands_lf", d_lfm, sb_rows, sb_cols, h_thresh, frame, planes, &gen);
  if (rc) return rc;
  RowPos rp;
  memset(&rp, 0, sizeof(rp));
  for (int r = 0; r < sb_rows; ++r) {
    rp.pos[r] = h_row_pos ? h_row_pos[r] : n_islands;
    if (rp.pos[r] < 0 || rp.pos[r] > n_islands || (r > 0 && rp.pos[r] < rp.pos[r - 1]))
      VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_intra_islands_lf: h_row_pos[%d] = %d is not a non-decreasing island count", r, rp.pos[r]);
  }
  int *gate_cur = nullptr, *gate_next = nullptr;
  const int n_gate = sb_rows * sb_cols;
  rc = lf_gate_sets(ctx, n_gate, &gate_cur, &gate_next);
  if (rc) return rc;
  LfThreshDev th;
  memcpy(&th, h_thresh, sizeof(th));
  const FrameDev f = to_dev(frame);
  const int grid = sb_rows * planes + n_islands;
  const txfm::Coefs cf = { d_coeffs, ctx->coeff16 };
  int *ticket = nullptr, *ticket_next = nullptr;
  rc = lf_tickets(ctx, &ticket, &ticket_next);
  if (rc) return rc;
#define WALK_LF(PIX, SH)                                                                                              \
  hipLaunchKernelGGL((walk_lf_kernel<PIX, SH>), dim3(grid), dim3(256), 0, ctx->stream, d_lfm, sb_cols, sb_rows, planes, th, f, \
                     frame->aheight[0] / 8, ctx->lf_err_flag, gate_cur, gate_next, n_gate, d_sb_expected, d_tasks, d_islands, \
                     d_wave_off, cf, (lf_granule *)ctx->lf_hand, gen, rp, ticket, ticket_next)
  if (!frame->hbd)
    WALK_LF(uint8_t, 0);
  else if (frame->bit_depth == 10)
    WALK_LF(uint16_t, 2);
  else if (frame->bit_depth == 12)
    WALK_LF(uint16_t, 4);
  else
    WALK_LF(uint16_t, 0);
#undef WALK_LF
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
