/*
 * ref_lf_driver.c — drives the REFERENCE's loop-filter driver code (test infrastructure;
 * compiled only into oracle/_ref/libvpxref.so, against the reference's own headers).
 *
 * Everything that decides and filters is the reference's object code:
 *   vp9_loop_filter_init, vp9_build_mask, vp9_adjust_mask, vp9_filter_block_plane_ss00/ss11
 *   (libvpx/vp9/common/vp9_loopfilter.c:238, 1528, 766, 1241, 1326) and the vpx_lpf_*_c kernels.
 * This file only (a) fills a MODE_INFO grid from a flat block list, (b) walks superblocks in the
 * raster order of loop_filter_rows (vp9_loopfilter.c:1440-1468) and (c) defines the two rtcd
 * POINTER variables vp9_loopfilter.c dispatches through, initialised to the reference's own C
 * functions exactly as setup_rtcd_internal does before any SIMD override
 * (vpx-master/vpx_dsp_rtcd.h:2074-2085).  The SSE2-suffixed names the Win64 rtcd header
 * hard-wires are mapped to the same-stem _c functions with -D flags in oracle/Makefile.
 */
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "./vpx_dsp_rtcd.h"
#include "vp9/common/vp9_loopfilter.h"
#include "vp9/common/vp9_onyxc_int.h"

void (*vpx_lpf_horizontal_16)(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit,
                              const uint8_t *thresh) = vpx_lpf_horizontal_16_c;
void (*vpx_lpf_horizontal_16_dual)(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit,
                                   const uint8_t *thresh) = vpx_lpf_horizontal_16_dual_c;

/* blocks[i] = { x, y, size (8..64, square), tx_log2 (2..5), level, skip, inter }.
 * planes: 3 pointers (uint8 or uint16 samples), strides in samples.
 * lfm_out: sb_rows*sb_cols records of sizeof(LOOP_FILTER_MASK) (after vp9_adjust_mask). */
int ref_lf_frame(const int32_t *blocks, int n_blocks, int aw, int ah, void *const planes[3],
                 const int strides[3], int bd, int hbd, int sharpness, void *lfm_out, int do_filter) {
  VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof(*cm));
  const int mi_rows = ah / 8, mi_cols = aw / 8;
  const int sb_rows = (mi_rows + 7) / 8, sb_cols = (mi_cols + 7) / 8;
  MODE_INFO *mis = (MODE_INFO *)calloc(n_blocks, sizeof(*mis));
  int levels[MAX_SEGMENTS], n_levels = 0;
  if (!cm || !mis) return -1;
  cm->mi_rows = mi_rows;
  cm->mi_cols = mi_cols;
  cm->mi_stride = mi_cols + 8;
  cm->use_highbitdepth = hbd;
  cm->bit_depth = (vpx_bit_depth_t)bd;
  cm->lf.sharpness_level = sharpness;
  cm->lf.filter_level = 32;
  cm->lf.lfm_stride = sb_cols;
  cm->lf.lfm = (LOOP_FILTER_MASK *)calloc((size_t)sb_rows * sb_cols, sizeof(LOOP_FILTER_MASK));
  vp9_loop_filter_init(cm);
  for (int i = 0; i < n_blocks; ++i) {
    const int32_t *b = blocks + 7 * i;
    MODE_INFO *mi = &mis[i];
    const int size = b[2], level = b[4];
    int seg = -1;
    for (int k = 0; k < n_levels; ++k)
      if (levels[k] == level) seg = k;
    if (seg < 0) {
      if (n_levels == MAX_SEGMENTS) return -2;
      seg = n_levels;
      levels[n_levels++] = level;
      memset(cm->lf_info.lvl[seg], level, sizeof(cm->lf_info.lvl[seg]));
    }
    mi->sb_type = size == 8 ? BLOCK_8X8 : size == 16 ? BLOCK_16X16 : size == 32 ? BLOCK_32X32 : BLOCK_64X64;
    mi->tx_size = (TX_SIZE)(b[3] - 2);
    mi->skip = (uint8_t)b[5];
    mi->segment_id = (int8_t)seg;
    mi->ref_frame[0] = b[6] ? LAST_FRAME : INTRA_FRAME;
    mi->ref_frame[1] = NONE;
    mi->mode = b[6] ? NEARESTMV : DC_PRED;
    vp9_build_mask(cm, mi, b[1] >> 3, b[0] >> 3, size >> 3, size >> 3);
  }
  for (int mi_row = 0; mi_row < mi_rows; mi_row += 8) {
    for (int mi_col = 0; mi_col < mi_cols; mi_col += 8) {
      LOOP_FILTER_MASK *lfm = get_lfm(&cm->lf, mi_row, mi_col);
      struct macroblockd_plane pl[3];
      memset(pl, 0, sizeof(pl));
      vp9_adjust_mask(cm, mi_row, mi_col, lfm);
      if (!do_filter) continue;
      for (int p = 0; p < 3; ++p) {
        const int ss = p ? 1 : 0;
        const size_t off = (size_t)((mi_row * 8) >> ss) * strides[p] + ((mi_col * 8) >> ss);
        pl[p].subsampling_x = pl[p].subsampling_y = ss;
        pl[p].dst.stride = strides[p];
        pl[p].dst.buf = hbd ? CONVERT_TO_BYTEPTR((uint16_t *)planes[p] + off) : (uint8_t *)planes[p] + off;
      }
      vp9_filter_block_plane_ss00(cm, &pl[0], mi_row, lfm);
      vp9_filter_block_plane_ss11(cm, &pl[1], mi_row, lfm);
      vp9_filter_block_plane_ss11(cm, &pl[2], mi_row, lfm);
    }
  }
  memcpy(lfm_out, cm->lf.lfm, (size_t)sb_rows * sb_cols * sizeof(LOOP_FILTER_MASK));
  free(cm->lf.lfm);
  free(mis);
  free(cm);
  return (int)sizeof(LOOP_FILTER_MASK);
}
