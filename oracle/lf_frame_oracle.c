/*
 * lf_frame_oracle.c — CPU restatement of the VP9 loop-filter DRIVER: which filter runs on
 * which 8-pixel edge segment, with which thresholds, in which order.
 * TEST INFRASTRUCTURE ONLY (see vp9_oracle.h).
 *
 * Follows (relative to /root/reference/libvpx/):
 *   vp9/common/vp9_loopfilter.c:1424-1469  loop_filter_rows: superblocks in raster order,
 *                                          per superblock plane 0,1,2
 *   vp9/common/vp9_loopfilter.c:1241-1324  vp9_filter_block_plane_ss00 (luma)
 *   vp9/common/vp9_loopfilter.c:1326-1422  vp9_filter_block_plane_ss11 (4:2:0 chroma)
 *   vp9/common/vp9_loopfilter.c:297-375    filter_selectively_vert_row2
 *   vp9/common/vp9_loopfilter.c:453-544    filter_selectively_horiz
 *
 * The reference walks masks with "dual" calls covering two 8-line segments.  A dual call is
 * two single calls (vpx_dsp/loopfilter.c:131-137 etc.) — except the two 16-wide duals, which
 * apply the FIRST segment's thresholds to both (vp9_loopfilter.c:318-320, 466-469).  This
 * restatement therefore decides, per 8-line segment, (kind, level) and issues single calls in
 * the same per-line order: edge filter at position c, then the interior 4x4 edge at c+4.
 */
#include <string.h>

#include "vp9_oracle.h"

typedef struct {
  uint8_t *p8;
  uint16_t *p16;
  int stride;
  int bd;
} lfplane;

static void call_lpf(const lfplane *pl, int x, int y, int vertical, int kind, int level,
                     const vp9o_lf_thresh *th) {
  if (pl->p16)
    vp9o_highbd_lpf(vertical, kind, 0, pl->p16 + (size_t)y * pl->stride + x, pl->stride,
                    &th->mblim[level], &th->lim[level], &th->hev_thr[level], NULL, NULL, NULL, pl->bd);
  else
    vp9o_lpf(vertical, kind, 0, pl->p8 + (size_t)y * pl->stride + x, pl->stride, &th->mblim[level],
             &th->lim[level], &th->hev_thr[level], NULL, NULL, NULL);
}

/* One plane of one superblock.  ncol = 8 (luma) or 4 (chroma) mask columns per mask row;
 * lfl[row*ncol + col] = filter level of that 8x8.  rows = number of mask rows to process.
 * m16/m8/m4/mint: left_* masks for the vertical pass, above_* for the horizontal pass. */
static void filter_sb_plane(const lfplane *pl, int x0, int y0, int ncol, int rows, uint64_t l16,
                            uint64_t l8, uint64_t l4, uint64_t a16, uint64_t a8, uint64_t a4,
                            uint64_t mint, const uint8_t *lfl, int top_row_is_frame_top,
                            int skip_int_row /* mask row whose interior 4x4 edge is skipped, or -1 */,
                            const vp9o_lf_thresh *th) {
  /* vertical edges: mask rows are taken two at a time (filter_selectively_vert_row2) */
  for (int r = 0; r < rows; r += 2) {
    for (int c = 0; c < ncol; ++c) {
      for (int i = 0; i < 2; ++i) { /* the two 8-line halves of the dual structure */
        const int rr = r + i;
        if (rr >= 8) continue;
        const int bit = rr * ncol + c;
        const int y = y0 + rr * 8, x = x0 + c * 8;
        int level = lfl[bit];
        if ((l16 >> bit) & 1) {
          /* vpx_lpf_vertical_16_dual takes lfis[0] only (vp9_loopfilter.c:318-320) */
          if (i == 1 && ((l16 >> (r * ncol + c)) & 1)) level = lfl[r * ncol + c];
          call_lpf(pl, x, y, 1, 16, level, th);
        }
        level = lfl[bit];
        if ((l8 >> bit) & 1) call_lpf(pl, x, y, 1, 8, level, th);
        if ((l4 >> bit) & 1) call_lpf(pl, x, y, 1, 4, level, th);
        if ((mint >> bit) & 1) call_lpf(pl, x + 4, y, 1, 4, level, th);
      }
    }
  }
  /* horizontal edges, one mask row at a time (filter_selectively_horiz) */
  for (int r = 0; r < rows; ++r) {
    const int edge_ok = !(top_row_is_frame_top && r == 0);
    const int int_ok = r != skip_int_row;
    int run16 = 0; /* position inside a run of consecutive 16-wide segments */
    for (int c = 0; c < ncol; ++c) {
      const int bit = r * ncol + c;
      const int y = y0 + r * 8, x = x0 + c * 8;
      const int level = lfl[bit];
      const int b16 = edge_ok && ((a16 >> bit) & 1), b8 = edge_ok && ((a8 >> bit) & 1);
      const int b4 = edge_ok && ((a4 >> bit) & 1), bi = int_ok && ((mint >> bit) & 1);
      if (b16) {
        /* vpx_lpf_horizontal_16_dual: the second segment of a pair reuses the first's
         * thresholds (vp9_loopfilter.c:466-469); pairs form from the start of a run */
        const int lv = (run16 & 1) ? lfl[bit - 1] : level;
        call_lpf(pl, x, y, 0, 16, lv, th);
        ++run16;
        /* the 16-wide branch never filters the interior edge (else-if chain, :465-538) */
        continue;
      }
      run16 = 0;
      if (b8) {
        call_lpf(pl, x, y, 0, 8, level, th);
        if (bi) call_lpf(pl, x, y + 4, 0, 4, level, th);
      } else if (b4) {
        call_lpf(pl, x, y, 0, 4, level, th);
        if (bi) call_lpf(pl, x, y + 4, 0, 4, level, th);
      } else if (bi) {
        call_lpf(pl, x, y + 4, 0, 4, level, th);
      }
    }
  }
}

void vp9o_loop_filter_frame(const vp9o_lfm *lfm, int sb_rows, int sb_cols, const vp9o_lf_thresh *th,
                            void *const planes[3], const int strides[3], int mi_rows, int bd, int hbd,
                            int nplanes) {
  for (int sr = 0; sr < sb_rows; ++sr) {
    for (int sc = 0; sc < sb_cols; ++sc) {
      const vp9o_lfm *m = &lfm[sr * sb_cols + sc];
      const int mi_row = sr * 8;
      const int rows = mi_rows - mi_row < 8 ? mi_rows - mi_row : 8;
      for (int p = 0; p < nplanes; ++p) {
        lfplane pl = { hbd ? NULL : (uint8_t *)planes[p], hbd ? (uint16_t *)planes[p] : NULL, strides[p], bd };
        if (p == 0) {
          filter_sb_plane(&pl, sc * 64, sr * 64, 8, rows, m->left_y[2], m->left_y[1], m->left_y[0],
                          m->above_y[2], m->above_y[1], m->above_y[0], m->int_4x4_y, m->lfl_y, mi_row == 0,
                          -1, th);
        } else {
          /* chroma levels are sampled from the luma level map (vp9_loopfilter.c:1344-1348) */
          uint8_t lfl_uv[16];
          memset(lfl_uv, 0, sizeof(lfl_uv));
          for (int r = 0; r < rows; r += 2)
            for (int c = 0; c < 4; ++c) lfl_uv[(r >> 1) * 4 + c] = m->lfl_y[r * 8 + 2 * c];
          const int crow = (rows + 1) >> 1;
          /* interior 4x4 edge of the chroma row that maps to the last (odd) mi row is skipped
           * (skip_border_4x4_r, :1385-1387) */
          int skip = -1;
          for (int r = 0; r < rows; r += 2)
            if (mi_row + r == mi_rows - 1) skip = r >> 1;
          filter_sb_plane(&pl, sc * 32, sr * 32, 4, crow, m->left_uv[2], m->left_uv[1], m->left_uv[0],
                          m->above_uv[2], m->above_uv[1], m->above_uv[0], m->int_4x4_uv, lfl_uv, mi_row == 0,
                          skip, th);
        }
      }
    }
  }
}
