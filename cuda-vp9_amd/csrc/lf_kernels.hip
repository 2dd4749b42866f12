// lf_kernels.hip — loop filter of a whole frame (SURVEY §8 a11–a12).
//
// Ordering.  libvpx filters superblocks in raster order, per superblock all vertical edges
// then all horizontal ones (vp9_loopfilter.c:1424-1469, 1241-1422).  Superblock (r,c) touches
// columns [64c-8, 64c+63] (vertical pass, rows of SB row r) and rows [64r-8, 64r+63]
// (horizontal pass, columns of SB column c), so it must run after (r,c-1) and (r-1,c+1) —
// libvpx's own row-MT sync rule (vp9_thread_common.c:38-55).  All superblocks with the same
// t = c + 2r are independent: one launch per anti-diagonal t, one wavefront per
// (superblock, plane).
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
#include "vp9hip_internal.h"

namespace {

struct LfThreshDev {
  uint8_t mblim[64], lim[64], hev_thr[64];
};

constexpr int TP = 76;  // LDS tile pitch in samples (19 dwords for 8-bit: odd, conflict-free column walks)

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int sclamp(int t, int bd) {
  const int lo = -(128 << (bd - 8)), hi = (128 << (bd - 8)) - 1;
  return t < lo ? lo : (t > hi ? hi : t);
}

// Filter one line across one edge.  b points at q0; taps at b[k*step].  vpx_dsp/loopfilter.c:
// filter_mask :33, flat_mask4 :49, flat_mask5 :62, hev_mask :72, filter4 :76, filter8 :162,
// filter16 :235; highbd forms :359-447 (thresholds << (bd-8)).
template <typename Pix>
__device__ __forceinline__ void filter_edge(Pix *b, int step, int kind, int blimit, int limit, int thresh, int bd) {
  const int sh = bd - 8;
  const int lim = limit << sh, blim = blimit << sh, one = 1 << sh, thr = thresh << sh;
  const int p3 = b[-4 * step], p2 = b[-3 * step], p1 = b[-2 * step], p0 = b[-step];
  const int q0 = b[0], q1 = b[step], q2 = b[2 * step], q3 = b[3 * step];
  const bool mask = !(iabs(p3 - p2) > lim || iabs(p2 - p1) > lim || iabs(p1 - p0) > lim || iabs(q1 - q0) > lim ||
                      iabs(q2 - q1) > lim || iabs(q3 - q2) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > blim);
  bool flat = false, flat2 = false;
  if (kind >= 8)
    flat = !(iabs(p1 - p0) > one || iabs(q1 - q0) > one || iabs(p2 - p0) > one || iabs(q2 - q0) > one ||
             iabs(p3 - p0) > one || iabs(q3 - q0) > one);
  if (kind == 16 && flat && mask) {
    const int p4 = b[-5 * step], p5 = b[-6 * step], p6 = b[-7 * step], p7 = b[-8 * step];
    const int q4 = b[4 * step], q5 = b[5 * step], q6 = b[6 * step], q7 = b[7 * step];
    flat2 = !(iabs(p4 - p0) > one || iabs(q4 - q0) > one || iabs(p5 - p0) > one || iabs(q5 - q0) > one ||
              iabs(p6 - p0) > one || iabs(q6 - q0) > one || iabs(p7 - p0) > one || iabs(q7 - q0) > one);
    if (flat2) {
      // 15-tap [1 1 1 1 1 1 1 2 1 1 1 1 1 1 1] with replication at p7 / q7: sliding sum
      int s = p7 * 7 + p6 * 2 + p5 + p4 + p3 + p2 + p1 + p0 + q0;
      b[-7 * step] = (Pix)((s + 8) >> 4);
      s += q1 - p7 + p5 - p6; b[-6 * step] = (Pix)((s + 8) >> 4);
      s += q2 - p7 + p4 - p5; b[-5 * step] = (Pix)((s + 8) >> 4);
      s += q3 - p7 + p3 - p4; b[-4 * step] = (Pix)((s + 8) >> 4);
      s += q4 - p7 + p2 - p3; b[-3 * step] = (Pix)((s + 8) >> 4);
      s += q5 - p7 + p1 - p2; b[-2 * step] = (Pix)((s + 8) >> 4);
      s += q6 - p7 + p0 - p1; b[-1 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p7 + q0 - p0; b[0] = (Pix)((s + 8) >> 4);
      s += q7 - p6 + q1 - q0; b[1 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p5 + q2 - q1; b[2 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p4 + q3 - q2; b[3 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p3 + q4 - q3; b[4 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p2 + q5 - q4; b[5 * step] = (Pix)((s + 8) >> 4);
      s += q7 - p1 + q6 - q5; b[6 * step] = (Pix)((s + 8) >> 4);
      return;
    }
  }
  if (flat && mask) {
    // 7-tap [1 1 1 2 1 1 1] with replication at p3 / q3
    b[-3 * step] = (Pix)((p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3);
    b[-2 * step] = (Pix)((p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3);
    b[-1 * step] = (Pix)((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3);
    b[0] = (Pix)((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3);
    b[1 * step] = (Pix)((p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3 + 4) >> 3);
    b[2 * step] = (Pix)((p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3 + 4) >> 3);
    return;
  }
  // narrow filter
  const int off = 0x80 << sh;
  const int hev = (iabs(p1 - p0) > thr || iabs(q1 - q0) > thr) ? -1 : 0;
  const int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
  int f = sclamp(ps1 - qs1, bd) & hev;
  f = sclamp(f + 3 * (qs0 - ps0), bd) & (mask ? -1 : 0);
  const int f1 = sclamp(f + 4, bd) >> 3, f2 = sclamp(f + 3, bd) >> 3;
  b[0] = (Pix)(sclamp(qs0 - f1, bd) + off);
  b[-step] = (Pix)(sclamp(ps0 + f2, bd) + off);
  f = ((f1 + 1) >> 1) & ~hev;
  b[step] = (Pix)(sclamp(qs1 - f, bd) + off);
  b[-2 * step] = (Pix)(sclamp(ps1 + f, bd) + off);
}

template <typename Pix>
__global__ __launch_bounds__(64) void lf_diag_kernel(const vp9hip_lfm *__restrict__ lfms, int sb_cols, int t,
                                                     int r_min, LfThreshDev th, FrameDev f, int mi_rows) {
  __shared__ Pix tile[72 * TP];
  __shared__ uint8_t lvl[64];
  const int lane = threadIdx.x;
  const int sr = r_min + blockIdx.x, sc = t - 2 * sr;
  const int pl = blockIdx.y;
  const vp9hip_lfm &m = lfms[sr * sb_cols + sc];
  const int bd = f.bit_depth;
  const int ss = pl ? 1 : 0;
  const int n = 64 >> ss;           // samples per superblock side in this plane
  const int ncol = 8 >> ss;         // mask columns per mask row
  const int x0 = sc * n, y0 = sr * n;
  Pix *plane = (Pix *)f.plane[pl];
  const int stride = f.stride[pl];
  const int pw = f.awidth[pl], ph = f.aheight[pl];
  const int mi_row = sr * 8;
  const int rows_mi = min(8, mi_rows - mi_row);
  const int mrows = pl ? ((rows_mi + 1) >> 1) : rows_mi;  // mask rows of this plane

  uint64_t l16, l8, l4, a16, a8, a4, mint;
  if (pl == 0) {
    l16 = m.left_y[2]; l8 = m.left_y[1]; l4 = m.left_y[0];
    a16 = m.above_y[2]; a8 = m.above_y[1]; a4 = m.above_y[0];
    mint = m.int_4x4_y;
    lvl[lane] = m.lfl_y[lane];
  } else {
    l16 = m.left_uv[2]; l8 = m.left_uv[1]; l4 = m.left_uv[0];
    a16 = m.above_uv[2]; a8 = m.above_uv[1]; a4 = m.above_uv[0];
    mint = m.int_4x4_uv;
    // lfl_uv[(r>>1)*4 + c] = lfl_y[r*8 + 2c] for even mi rows r (vp9_loopfilter.c:1344-1348)
    if (lane < 16) lvl[lane] = m.lfl_y[(lane >> 2) * 16 + (lane & 3) * 2];
  }
  // stage tile: rows y0-8 .. y0+n-1, cols x0-8 .. x0+n-1 (clipped to the plane)
  const int tw = n + 8, thh = n + 8;
  for (int i = lane; i < tw * thh; i += 64) {
    const int r = i / tw, c = i - r * tw;
    const int gx = x0 - 8 + c, gy = y0 - 8 + r;
    Pix v = 0;
    if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) v = plane[(size_t)gy * stride + gx];
    tile[r * TP + c] = v;
  }
  __syncthreads();

  // ---- vertical edges: lane = sample row
  if (lane < n && y0 + lane < ph) {
    const int mr = lane >> 3;  // mask row
    if (mr < mrows) {
      Pix *row = tile + (8 + lane) * TP + 8;
      for (int c = 0; c < ncol; ++c) {
        const int bit = mr * ncol + c;
        int level = lvl[bit];
        Pix *b = row + c * 8;
        if ((l16 >> bit) & 1) {
          // dual-16 applies the even mask row's thresholds to both rows of the pair
          int lv = level;
          if ((mr & 1) && ((l16 >> (bit - ncol)) & 1)) lv = lvl[bit - ncol];
          filter_edge<Pix>(b, 1, 16, th.mblim[lv], th.lim[lv], th.hev_thr[lv], bd);
        }
        if ((l8 >> bit) & 1) filter_edge<Pix>(b, 1, 8, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
        if ((l4 >> bit) & 1) filter_edge<Pix>(b, 1, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
        if ((mint >> bit) & 1)
          filter_edge<Pix>(b + 4, 1, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
      }
    }
  }
  __syncthreads();

  // ---- horizontal edges: lane = sample column
  if (lane < n && x0 + lane < pw) {
    const int c = lane >> 3;  // mask column
    // position of this column's segment inside a run of 16-wide segments decides whose
    // thresholds a "dual" call used (vp9_loopfilter.c:466-469)
    int skip_int = -1;
    if (pl)
      for (int r = 0; r < rows_mi; r += 2)
        if (mi_row + r == mi_rows - 1) skip_int = r >> 1;
    for (int mr = 0; mr < mrows; ++mr) {
      const bool edge_ok = !(mi_row == 0 && mr == 0);
      const int bit = mr * ncol + c;
      const int level = lvl[bit];
      Pix *b = tile + (8 + mr * 8) * TP + 8 + lane;
      const uint64_t rowmask16 = edge_ok ? ((a16 >> (mr * ncol)) & ((1u << ncol) - 1)) : 0;
      const bool b16 = (rowmask16 >> c) & 1;
      const bool b8 = edge_ok && ((a8 >> bit) & 1), b4 = edge_ok && ((a4 >> bit) & 1);
      const bool bi = (mr != skip_int) && ((mint >> bit) & 1);
      if (b16) {
        int run = 0;  // number of consecutive 16-wide segments immediately to the left
        for (int k = c - 1; k >= 0 && ((rowmask16 >> k) & 1); --k) ++run;
        const int lv = (run & 1) ? lvl[bit - 1] : level;
        filter_edge<Pix>(b, TP, 16, th.mblim[lv], th.lim[lv], th.hev_thr[lv], bd);
      } else if (b8) {
        filter_edge<Pix>(b, TP, 8, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
        if (bi) filter_edge<Pix>(b + 4 * TP, TP, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
      } else if (b4) {
        filter_edge<Pix>(b, TP, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
        if (bi) filter_edge<Pix>(b + 4 * TP, TP, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
      } else if (bi) {
        filter_edge<Pix>(b + 4 * TP, TP, 4, th.mblim[level], th.lim[level], th.hev_thr[level], bd);
      }
    }
  }
  __syncthreads();

  // write the tile back (everything except the untouched top-left 8x8 corner)
  for (int i = lane; i < tw * thh; i += 64) {
    const int r = i / tw, c = i - r * tw;
    if (r < 8 && c < 8) continue;
    const int gx = x0 - 8 + c, gy = y0 - 8 + r;
    if (gx >= 0 && gy >= 0 && gx < pw && gy < ph) plane[(size_t)gy * stride + gx] = tile[r * TP + c];
  }
}

}  // namespace

extern "C" int vp9hip_loop_filter_frame(vp9hip_ctx *ctx, const vp9hip_lfm *d_lfm, int sb_rows, int sb_cols,
                                        const vp9hip_lf_thresh *h_thresh, const vp9hip_frame *frame, int planes) {
  if (!ctx) return VP9HIP_EINVAL;
  if (!d_lfm || sb_rows <= 0 || sb_cols <= 0 || !h_thresh || !frame_ok(frame) || (planes != 1 && planes != 3))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_loop_filter_frame: bad argument");
  if (sb_rows != (frame->aheight[0] + 63) / 64 || sb_cols != (frame->awidth[0] + 63) / 64)
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_loop_filter_frame: %dx%d superblocks do not cover a %dx%d frame",
                sb_cols, sb_rows, frame->awidth[0], frame->aheight[0]);
  if (planes == 3 && (frame->awidth[1] * 2 != frame->awidth[0] || frame->aheight[1] * 2 != frame->aheight[0]))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_loop_filter_frame: only 4:2:0 chroma is supported");
  LfThreshDev th;
  memcpy(&th, h_thresh, sizeof(th));
  const FrameDev f = to_dev(frame);
  const int mi_rows = frame->aheight[0] / 8;
  const int t_max = (sb_cols - 1) + 2 * (sb_rows - 1);
  for (int t = 0; t <= t_max; ++t) {
    // superblock rows r with 0 <= t - 2r < sb_cols
    int r_min = (t - (sb_cols - 1) + 1) / 2;
    if (t - (sb_cols - 1) <= 0) r_min = 0;
    int r_max = t / 2;
    if (r_max > sb_rows - 1) r_max = sb_rows - 1;
    const int cnt = r_max - r_min + 1;
    if (cnt <= 0) continue;
    if (frame->hbd)
      hipLaunchKernelGGL(lf_diag_kernel<uint16_t>, dim3(cnt, planes), dim3(64), 0, ctx->stream, d_lfm, sb_cols, t,
                         r_min, th, f, mi_rows);
    else
      hipLaunchKernelGGL(lf_diag_kernel<uint8_t>, dim3(cnt, planes), dim3(64), 0, ctx->stream, d_lfm, sb_cols, t,
                         r_min, th, f, mi_rows);
  }
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
