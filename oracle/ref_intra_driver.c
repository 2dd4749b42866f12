/*
 * ref_intra_driver.c — drives the REFERENCE's intra edge builder + predictor dispatch
 * (vp9_predict_intra_block / build_intra_predictors{,_high}, libvpx/vp9/common/vp9_reconintra.c:
 * 113-424) for one transform block.  Test infrastructure; compiled only into oracle/_ref.
 *
 * All arithmetic is the reference's object code.  This file fills the few MACROBLOCKD fields the
 * function reads (the way set_mi_row_col / set_plane_n4 do, vp9_onyxc_int.h, vp9_decodeframe.c:
 * 692-702) and defines the rtcd POINTER variables vp9_reconintra.c's tables are initialised from,
 * pointing them at the reference's own same-stem C functions (what setup_rtcd_internal does first,
 * vpx-master/vpx_dsp_rtcd.h:2074).  SSE2-suffixed names are mapped to _c with -D flags (Makefile).
 */
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "./vpx_dsp_rtcd.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/common/vp9_reconintra.h"

#define P8(stem) \
  void (*stem)(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left) = stem##_c;
#define P16(stem)                                                                                  \
  void (*stem)(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd) = \
      stem##_c;
P8(vpx_d153_predictor_4x4) P8(vpx_d153_predictor_8x8) P8(vpx_d153_predictor_16x16) P8(vpx_d153_predictor_32x32)
P8(vpx_d207_predictor_8x8) P8(vpx_d207_predictor_16x16) P8(vpx_d207_predictor_32x32)
P8(vpx_d45_predictor_16x16) P8(vpx_d45_predictor_32x32)
P8(vpx_d63_predictor_4x4) P8(vpx_d63_predictor_8x8) P8(vpx_d63_predictor_16x16) P8(vpx_d63_predictor_32x32)
P16(vpx_highbd_d117_predictor_8x8) P16(vpx_highbd_d117_predictor_16x16) P16(vpx_highbd_d117_predictor_32x32)
P16(vpx_highbd_d135_predictor_8x8) P16(vpx_highbd_d135_predictor_16x16) P16(vpx_highbd_d135_predictor_32x32)
P16(vpx_highbd_d153_predictor_8x8) P16(vpx_highbd_d153_predictor_16x16) P16(vpx_highbd_d153_predictor_32x32)
P16(vpx_highbd_d207_predictor_8x8) P16(vpx_highbd_d207_predictor_16x16) P16(vpx_highbd_d207_predictor_32x32)
P16(vpx_highbd_d45_predictor_4x4) P16(vpx_highbd_d45_predictor_8x8) P16(vpx_highbd_d45_predictor_16x16)
P16(vpx_highbd_d45_predictor_32x32)
P16(vpx_highbd_d63_predictor_8x8) P16(vpx_highbd_d63_predictor_16x16) P16(vpx_highbd_d63_predictor_32x32)

/* Predicts one transform block in place in `plane` (sample (0,0) of the plane, stride in samples).
 * mi_row/mi_col/bw8/bh8: the prediction block in 8-pixel units; aoff/loff: transform block offset
 * inside it in 4-sample units of the plane; awidth/aheight: aligned LUMA size. */
void ref_predict_intra(void *plane_base, int stride, int hbd, int bd, int awidth, int aheight, int plane,
                       int mi_row, int mi_col, int bw8, int bh8, int tx_size, int mode, int aoff, int loff) {
  static int inited = 0;
  MACROBLOCKD xd;
  YV12_BUFFER_CONFIG buf;
  MODE_INFO dummy;
  const int mi_rows = aheight / 8, mi_cols = awidth / 8;
  const int ss = plane ? 1 : 0;
  if (!inited) {
    vp9_init_intra_predictors();
    inited = 1;
  }
  memset(&xd, 0, sizeof(xd));
  memset(&buf, 0, sizeof(buf));
  memset(&dummy, 0, sizeof(dummy));
  buf.y_width = awidth;
  buf.y_height = aheight;
  buf.uv_width = awidth >> 1;
  buf.uv_height = aheight >> 1;
  buf.flags = hbd ? YV12_FLAG_HIGHBITDEPTH : 0;
  xd.cur_buf = &buf;
  xd.bd = bd;
  xd.plane[1].subsampling_x = xd.plane[1].subsampling_y = 1;
  xd.plane[2].subsampling_x = xd.plane[2].subsampling_y = 1;
  /* set_mi_row_col (vp9_onyxc_int.h) */
  xd.mb_to_top_edge = -((mi_row * MI_SIZE) * 8);
  xd.mb_to_bottom_edge = ((mi_rows - bh8 - mi_row) * MI_SIZE) * 8;
  xd.mb_to_left_edge = -((mi_col * MI_SIZE) * 8);
  xd.mb_to_right_edge = ((mi_cols - bw8 - mi_col) * MI_SIZE) * 8;
  xd.above_mi = mi_row != 0 ? &dummy : NULL;
  xd.left_mi = mi_col != 0 ? &dummy : NULL;
  {
    /* pd->n4_wl = b_width_log2 of max(bsize, 8x8) minus subsampling (set_plane_n4) */
    int bwl = 0;
    while ((1 << bwl) < bw8 * 2) ++bwl;
    const int bwl_in = bwl - ss;
    const int x = ((mi_col * 8) >> ss) + 4 * aoff, y = ((mi_row * 8) >> ss) + 4 * loff;
    if (hbd) {
      uint16_t *p = (uint16_t *)plane_base + (size_t)y * stride + x;
      vp9_predict_intra_block(&xd, bwl_in, (TX_SIZE)tx_size, (PREDICTION_MODE)mode, CONVERT_TO_BYTEPTR(p), stride,
                              CONVERT_TO_BYTEPTR(p), stride, aoff, loff, plane);
    } else {
      uint8_t *p = (uint8_t *)plane_base + (size_t)y * stride + x;
      vp9_predict_intra_block(&xd, bwl_in, (TX_SIZE)tx_size, (PREDICTION_MODE)mode, p, stride, p, stride, aoff, loff,
                              plane);
    }
  }
}
