/*
 * ref_frame_driver.c — drives the REFERENCE's own block-walking code over a list of decoded
 * blocks (test infrastructure; compiled only into oracle/_ref/libvpxref.so, against the
 * reference's headers).  It pins the product's host packers (cuda-vp9_amd/csrc/vp9hip_pack.c):
 *
 *   ref_inter_frame   vp9_build_inter_predictors_sb + vp9_setup_scale_factors_for_frame +
 *                     vp9_setup_pre_planes / vp9_setup_dst_planes (libvpx/vp9/common/
 *                     vp9_reconinter.c:126-298, vp9_scale.c:46-170): MV averaging of sub-8x8
 *                     blocks, MV clamping, reference scaling, per-plane block sizes
 *   ref_intra_frame   vp9_foreach_transformed_block_in_plane (vp9_blockd.c:37-75) visiting
 *                     vp9_predict_intra_block (vp9_reconintra.c:404-424) the way the decoder's
 *                     intra loop does (vp9/decoder/vp9_decodeframe.c:1073-1115): visit order,
 *                     frame-edge clipping, uv transform size, availability, tile columns
 *                     (vp9_tile_set_col, vp9_tile_common.c:28-31)
 *   ref_lf_frame2     as ref_lf_frame (ref_lf_driver.c) for any BLOCK_SIZE
 *   ref_lf_levels     vp9_loop_filter_frame_init (vp9_loopfilter.c:252-295)
 *
 * Every decision and all arithmetic run in the reference's object code; this file only fills the
 * structures those functions read, the way set_offsets / set_mi_row_col / set_plane_n4 do
 * (vp9_decodeframe.c:692-702, 868-900; vp9_onyxc_int.h:422-433).
 *
 * Block record: 35 int32 —
 *   [0] mi_row [1] mi_col [2] sb_type [3] tx_size [4] skip [5] interp_filter [6] ref_frame0
 *   [7] ref_frame1 [8] mode [9] uv_mode [10..13] sub_mode [14] filter_level
 *   [15..18] mv[ref][row,col]  [19..34] sub_mv[blk][ref][row,col]
 */
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "./vpx_dsp_rtcd.h"
#include "vp9/common/vp9_blockd.h"
#include "vp9/common/vp9_loopfilter.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/common/vp9_reconinter.h"
#include "vp9/common/vp9_reconintra.h"
#include "vp9/common/vp9_scale.h"
#include "vp9/common/vp9_tile_common.h"

#define REC 35

static void fill_mi(MODE_INFO *mi, const int32_t *b) {
  memset(mi, 0, sizeof(*mi));
  mi->sb_type = (BLOCK_SIZE)b[2];
  mi->tx_size = (TX_SIZE)b[3];
  mi->skip = (int8_t)b[4];
  mi->interp_filter = (INTERP_FILTER)b[5];
  mi->ref_frame[0] = (MV_REFERENCE_FRAME)b[6];
  mi->ref_frame[1] = (MV_REFERENCE_FRAME)(b[7] > 0 ? b[7] : NONE);
  mi->mode = (PREDICTION_MODE)(b[6] > 0 ? NEARESTMV : b[8]);
  mi->uv_mode = (PREDICTION_MODE)b[9];
  for (int r = 0; r < 2; ++r) {
    mi->mv[r].as_mv.row = (int16_t)b[15 + 2 * r];
    mi->mv[r].as_mv.col = (int16_t)b[16 + 2 * r];
  }
  for (int k = 0; k < 4; ++k) {
    if (b[6] > 0) {
      for (int r = 0; r < 2; ++r) {
        mi->bmi[k].as_mv[r].as_mv.row = (int16_t)b[19 + 4 * k + 2 * r];
        mi->bmi[k].as_mv[r].as_mv.col = (int16_t)b[20 + 4 * k + 2 * r];
      }
    } else {
      mi->bmi[k].as_mode = (PREDICTION_MODE)b[10 + k];
    }
  }
}

static void fill_yv12(YV12_BUFFER_CONFIG *buf, void *const planes[3], const int strides[3], int w, int h, int ss,
                      int hbd, int bd) {
  memset(buf, 0, sizeof(*buf));
  buf->y_crop_width = w;
  buf->y_crop_height = h;
  buf->y_width = (w + 7) & ~7;
  buf->y_height = (h + 7) & ~7;
  buf->uv_crop_width = (w + ss) >> ss;
  buf->uv_crop_height = (h + ss) >> ss;
  buf->uv_width = buf->y_width >> ss;
  buf->uv_height = buf->y_height >> ss;
  buf->y_stride = strides[0];
  buf->uv_stride = strides[1];
  buf->subsampling_x = buf->subsampling_y = ss;
  buf->bit_depth = (unsigned)bd;
  buf->flags = hbd ? YV12_FLAG_HIGHBITDEPTH : 0;
  buf->y_buffer = hbd ? CONVERT_TO_BYTEPTR(planes[0]) : (uint8_t *)planes[0];
  buf->u_buffer = hbd ? CONVERT_TO_BYTEPTR(planes[1]) : (uint8_t *)planes[1];
  buf->v_buffer = hbd ? CONVERT_TO_BYTEPTR(planes[2]) : (uint8_t *)planes[2];
}

/* the assignments of set_mi_row_col (vp9_onyxc_int.h:422-429) and set_plane_n4
 * (vp9_decodeframe.c:692-702) */
static void set_block_geometry(MACROBLOCKD *xd, int mi_row, int mi_col, int bw8, int bh8, int mi_rows, int mi_cols,
                               int ss) {
  int bwl = 0, bhl = 0;
  while ((1 << bwl) < bw8 * 2) ++bwl;
  while ((1 << bhl) < bh8 * 2) ++bhl;
  xd->mb_to_top_edge = -((mi_row * MI_SIZE) * 8);
  xd->mb_to_bottom_edge = ((mi_rows - bh8 - mi_row) * MI_SIZE) * 8;
  xd->mb_to_left_edge = -((mi_col * MI_SIZE) * 8);
  xd->mb_to_right_edge = ((mi_cols - bw8 - mi_col) * MI_SIZE) * 8;
  for (int i = 0; i < MAX_MB_PLANE; ++i) {
    const int s = i ? ss : 0;
    xd->plane[i].subsampling_x = xd->plane[i].subsampling_y = s;
    xd->plane[i].n4_w = (uint16_t)((bw8 << 1) >> s);
    xd->plane[i].n4_h = (uint16_t)((bh8 << 1) >> s);
    xd->plane[i].n4_wl = (uint8_t)(bwl - s);
    xd->plane[i].n4_hl = (uint8_t)(bhl - s);
  }
}

/* cur_planes / ref_planes: pointers to sample (0,0) of planes that carry a border of at least
 * 160 samples replicated from the crop edge (the caller extends them; the reference's
 * encoder-side predictor reads the border instead of emulating it). */
int ref_inter_frame(const int32_t *blocks, int n_blocks, int w, int h, int ss, int bd, int hbd,
                    void *const cur_planes[3], const int cur_strides[3], void *const ref_planes[9],
                    const int ref_strides[9], const int ref_w[3], const int ref_h[3]) {
  const int mi_rows = ((h + 7) & ~7) / 8, mi_cols = ((w + 7) & ~7) / 8;
  YV12_BUFFER_CONFIG cur, refbuf[3];
  RefBuffer rb[3];
  MACROBLOCKD *xd = (MACROBLOCKD *)calloc(1, sizeof(*xd));
  if (!xd) return -1;
  fill_yv12(&cur, cur_planes, cur_strides, w, h, ss, hbd, bd);
  memset(rb, 0, sizeof(rb));
  for (int k = 0; k < 3; ++k) {
    if (ref_w[k] <= 0) continue;
    fill_yv12(&refbuf[k], &ref_planes[3 * k], &ref_strides[3 * k], ref_w[k], ref_h[k], ss, hbd, bd);
    rb[k].buf = &refbuf[k];
    rb[k].idx = k;
    vp9_setup_scale_factors_for_frame(&rb[k].sf, ref_w[k], ref_h[k], w, h, hbd);
    if (!vp9_is_valid_scale(&rb[k].sf)) {
      free(xd);
      return -2;
    }
  }
  xd->cur_buf = &cur;
  xd->bd = bd;
  for (int i = 0; i < n_blocks; ++i) {
    const int32_t *b = blocks + REC * i;
    MODE_INFO mi, *mip = &mi;
    if (b[6] <= 0) continue;
    fill_mi(&mi, b);
    const BLOCK_SIZE bsize = mi.sb_type < BLOCK_8X8 ? BLOCK_8X8 : mi.sb_type;
    const int bw8 = num_8x8_blocks_wide_lookup[bsize], bh8 = num_8x8_blocks_high_lookup[bsize];
    xd->mi = &mip;
    set_block_geometry(xd, b[0], b[1], bw8, bh8, mi_rows, mi_cols, ss);
    vp9_setup_dst_planes(xd->plane, &cur, b[0], b[1]);
    for (int r = 0; r < 1 + has_second_ref(&mi); ++r) {
      RefBuffer *ref_buf = &rb[mi.ref_frame[r] - LAST_FRAME];
      xd->block_refs[r] = ref_buf;
      vp9_setup_pre_planes(xd, r, ref_buf->buf, b[0], b[1], &ref_buf->sf);
    }
    vp9_build_inter_predictors_sb(xd, b[0], b[1], bsize);
  }
  free(xd);
  return 0;
}

typedef struct {
  MACROBLOCKD *xd;
  int32_t *log;
  int n, cap;
} intra_arg;

/* the body of the decoder's intra loop (vp9_decodeframe.c:1099-1106), prediction only */
static void intra_visit(int plane, int block, int row, int col, BLOCK_SIZE plane_bsize, TX_SIZE tx_size, void *argp) {
  intra_arg *a = (intra_arg *)argp;
  MACROBLOCKD *xd = a->xd;
  struct macroblockd_plane *pd = &xd->plane[plane];
  const MODE_INFO *mi = xd->mi[0];
  PREDICTION_MODE mode = plane == 0 ? mi->mode : mi->uv_mode;
  const int stride = pd->dst.stride;
  uint8_t *dst;
  (void)block;
  (void)plane_bsize;
  if (xd->cur_buf->flags & YV12_FLAG_HIGHBITDEPTH)
    dst = CONVERT_TO_BYTEPTR(CONVERT_TO_SHORTPTR(pd->dst.buf) + 4 * row * stride + 4 * col);
  else
    dst = &pd->dst.buf[4 * row * stride + 4 * col];
  if (mi->sb_type < BLOCK_8X8 && plane == 0) mode = mi->bmi[(row << 1) + col].as_mode;
  vp9_predict_intra_block(xd, pd->n4_wl, tx_size, mode, dst, stride, dst, stride, col, row, plane);
  if (a->n < a->cap) {
    int32_t *l = a->log + 5 * a->n;
    l[0] = plane;
    l[1] = row;
    l[2] = col;
    l[3] = tx_size;
    l[4] = mode;
  }
  ++a->n;
}

/* planes: sample (0,0) of buffers with a border (libvpx reads up to 2*bs beyond the block).
 * Returns the number of transform blocks visited; log receives 5 int32 per visit
 * (plane, row, col, tx_size, mode), rows/cols in 4-sample units inside the block. */
int ref_intra_frame(const int32_t *blocks, int n_blocks, int w, int h, int ss, int bd, int hbd, void *const planes[3],
                    const int strides[3], int log2_tile_cols, int32_t *log, int log_cap) {
  static int inited = 0;
  const int mi_rows = ((h + 7) & ~7) / 8, mi_cols = ((w + 7) & ~7) / 8;
  YV12_BUFFER_CONFIG cur;
  MODE_INFO dummy;
  VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof(*cm));
  MACROBLOCKD *xd = (MACROBLOCKD *)calloc(1, sizeof(*xd));
  intra_arg arg = { xd, log, 0, log_cap };
  if (!cm || !xd) return -1;
  if (!inited) {
    vp9_init_intra_predictors();
    inited = 1;
  }
  memset(&dummy, 0, sizeof(dummy));
  cm->mi_rows = mi_rows;
  cm->mi_cols = mi_cols;
  cm->log2_tile_cols = log2_tile_cols;
  fill_yv12(&cur, planes, strides, w, h, ss, hbd, bd);
  xd->cur_buf = &cur;
  xd->bd = bd;
  for (int i = 0; i < n_blocks; ++i) {
    const int32_t *b = blocks + REC * i;
    MODE_INFO mi, *mip = &mi;
    TileInfo tile;
    if (b[6] > 0) continue;
    fill_mi(&mi, b);
    const BLOCK_SIZE bsize = mi.sb_type < BLOCK_8X8 ? BLOCK_8X8 : mi.sb_type;
    const int bw8 = num_8x8_blocks_wide_lookup[bsize], bh8 = num_8x8_blocks_high_lookup[bsize];
    xd->mi = &mip;
    set_block_geometry(xd, b[0], b[1], bw8, bh8, mi_rows, mi_cols, ss);
    /* which tile column holds the block: the reference's own tile arithmetic */
    for (int t = 0; t < (1 << log2_tile_cols); ++t) {
      vp9_tile_set_col(&tile, cm, t);
      if (b[1] >= tile.mi_col_start && b[1] < tile.mi_col_end) break;
    }
    /* set_mi_row_col (vp9_onyxc_int.h:430-432) */
    xd->above_mi = (b[0] != 0) ? &dummy : NULL;
    xd->left_mi = (b[1] > tile.mi_col_start) ? &dummy : NULL;
    vp9_setup_dst_planes(xd->plane, &cur, b[0], b[1]);
    for (int plane = 0; plane < MAX_MB_PLANE; ++plane)
      vp9_foreach_transformed_block_in_plane(xd, bsize, plane, intra_visit, &arg);
  }
  free(xd);
  free(cm);
  return arg.n;
}

/* Loop-filter masks + filtering through the reference's driver, any BLOCK_SIZE.
 * lvl table: every block carries its own level; blocks with equal level share a segment id. */
int ref_lf_frame3(const int32_t *blocks, int n_blocks, int aw, int ah, void *const planes[3], const int strides[3],
                  int bd, int hbd, int sharpness, void *lfm_out, int do_filter, int chroma_ss);
int ref_lf_frame2(const int32_t *blocks, int n_blocks, int aw, int ah, void *const planes[3], const int strides[3],
                  int bd, int hbd, int sharpness, void *lfm_out, int do_filter) {
  return ref_lf_frame3(blocks, n_blocks, aw, ah, planes, strides, bd, hbd, sharpness, lfm_out, do_filter, 1);
}
/* chroma_ss 1: 4:2:0 (LF_PATH_420); 0: 4:4:4 — every plane through vp9_filter_block_plane_ss00 with the same
 * mask record (LF_PATH_444, libvpx/vp9/common/vp9_loopfilter.c:1433-1458) */
int ref_lf_frame3(const int32_t *blocks, int n_blocks, int aw, int ah, void *const planes[3], const int strides[3],
                  int bd, int hbd, int sharpness, void *lfm_out, int do_filter, int chroma_ss) {
  VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof(*cm));
  const int mi_rows = ah / 8, mi_cols = aw / 8;
  const int sb_rows = (mi_rows + 7) / 8, sb_cols = (mi_cols + 7) / 8;
  int levels[MAX_SEGMENTS], n_levels = 0;
  if (!cm) return -1;
  cm->mi_rows = mi_rows;
  cm->mi_cols = mi_cols;
  cm->use_highbitdepth = hbd;
  cm->bit_depth = (vpx_bit_depth_t)bd;
  cm->lf.sharpness_level = sharpness;
  cm->lf.filter_level = 32;
  cm->lf.lfm_stride = sb_cols;
  cm->lf.lfm = (LOOP_FILTER_MASK *)calloc((size_t)sb_rows * sb_cols, sizeof(LOOP_FILTER_MASK));
  vp9_loop_filter_init(cm);
  for (int i = 0; i < n_blocks; ++i) {
    const int32_t *b = blocks + REC * i;
    MODE_INFO mi;
    int seg = -1;
    fill_mi(&mi, b);
    for (int k = 0; k < n_levels; ++k)
      if (levels[k] == b[14]) seg = k;
    if (seg < 0) {
      if (n_levels == MAX_SEGMENTS) return -2;
      seg = n_levels;
      levels[n_levels++] = b[14];
      memset(cm->lf_info.lvl[seg], b[14], sizeof(cm->lf_info.lvl[seg]));
    }
    mi.segment_id = (int8_t)seg;
    {
      const BLOCK_SIZE bsize = mi.sb_type < BLOCK_8X8 ? BLOCK_8X8 : mi.sb_type;
      /* decode_block passes the block size in mi units, unclipped (vp9_decodeframe.c:1203-1204, 1240) */
      vp9_build_mask(cm, &mi, b[0], b[1], num_8x8_blocks_wide_lookup[bsize], num_8x8_blocks_high_lookup[bsize]);
    }
  }
  for (int mi_row = 0; mi_row < mi_rows; mi_row += 8) {
    for (int mi_col = 0; mi_col < mi_cols; mi_col += 8) {
      LOOP_FILTER_MASK *lfm = get_lfm(&cm->lf, mi_row, mi_col);
      struct macroblockd_plane pl[3];
      memset(pl, 0, sizeof(pl));
      vp9_adjust_mask(cm, mi_row, mi_col, lfm);
      if (!do_filter) continue;
      for (int p = 0; p < 3; ++p) {
        const int ss = p ? chroma_ss : 0;
        const size_t off = (size_t)((mi_row * 8) >> ss) * strides[p] + ((mi_col * 8) >> ss);
        pl[p].subsampling_x = pl[p].subsampling_y = ss;
        pl[p].dst.stride = strides[p];
        pl[p].dst.buf = hbd ? CONVERT_TO_BYTEPTR((uint16_t *)planes[p] + off) : (uint8_t *)planes[p] + off;
      }
      vp9_filter_block_plane_ss00(cm, &pl[0], mi_row, lfm);
      if (chroma_ss) {
        vp9_filter_block_plane_ss11(cm, &pl[1], mi_row, lfm);
        vp9_filter_block_plane_ss11(cm, &pl[2], mi_row, lfm);
      } else {
        vp9_filter_block_plane_ss00(cm, &pl[1], mi_row, lfm);
        vp9_filter_block_plane_ss00(cm, &pl[2], mi_row, lfm);
      }
    }
  }
  memcpy(lfm_out, cm->lf.lfm, (size_t)sb_rows * sb_cols * sizeof(LOOP_FILTER_MASK));
  free(cm->lf.lfm);
  free(cm);
  return (int)sizeof(LOOP_FILTER_MASK);
}

/* vp9_loop_filter_frame_init + the threshold tables of vp9_loop_filter_init. */
void ref_lf_levels(int default_lvl, int sharpness, const int32_t seg_enabled[8], const int32_t seg_data[8], int abs_delta,
                   int mode_ref_delta_enabled, const int8_t ref_deltas[4], const int8_t mode_deltas[2],
                   uint8_t out_lvl[8][4][2], uint8_t out_thresh[3][64]) {
  VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof(*cm));
  cm->lf.sharpness_level = sharpness;
  vp9_loop_filter_init(cm);
  cm->lf.mode_ref_delta_enabled = (uint8_t)mode_ref_delta_enabled;
  memcpy(cm->lf.ref_deltas, ref_deltas, 4);
  memcpy(cm->lf.mode_deltas, mode_deltas, 2);
  cm->seg.abs_delta = (uint8_t)abs_delta;
  for (int s = 0; s < 8; ++s) {
    if (seg_enabled[s]) {
      cm->seg.enabled = 1;
      cm->seg.feature_mask[s] |= 1u << SEG_LVL_ALT_LF;
      cm->seg.feature_data[s][SEG_LVL_ALT_LF] = (int16_t)seg_data[s];
    }
  }
  vp9_loop_filter_frame_init(cm, default_lvl);
  memcpy(out_lvl, cm->lf_info.lvl, sizeof(cm->lf_info.lvl));
  for (int l = 0; l < 64; ++l) {
    out_thresh[0][l] = cm->lf_info.lfthr[l].mblim[0];
    out_thresh[1][l] = cm->lf_info.lfthr[l].lim[0];
    out_thresh[2][l] = cm->lf_info.lfthr[l].hev_thr[0];
  }
  free(cm);
}

/* ---- one whole frame through the reference's C functions ------------------------------------------
 * ref_recon_frame: what the reference's decoder does to a frame after parsing it, block by block in
 * decode order (stock order, libvpx/vp9/decoder/vp9_decodeframe.c:1073-1196 run per block): inter blocks
 * = vp9_build_inter_predictors_sb then the inverse transform + add of every coded transform block; intra
 * blocks = per transform block vp9_predict_intra_block then inverse transform + add; then, when
 * filter != 0, vp9_build_mask per block (with libvpx's `eobtotal == 0 -> skip`, :1195) + vp9_adjust_mask +
 * vp9_filter_block_plane_ss00 / ss11 per superblock.  Used as the CPU baseline of bench.py (kind
 * "reference") and as the expected frame of the GPU frame tests — independent of the product's packers.
 *
 * coefs[p]: concatenated N*N blocks in decode order, one slot per visited transform block of every
 * non-skip block (detoken_block's layout, :919-1024); eobs[p]: frame-strided plane, one int at each
 * transform block's origin.  Transform selection by eob as vp9_idct{4x4,8x8,16x16,32x32}_add /
 * vp9_iht*_add do it (libvpx/vp9/common/vp9_idct.c:119-204); high-bitdepth buffers take the fork's
 * residual-storing full transforms + highbd_clip_pixel_add, the composition its phase B + block_sum make
 * (:173-341). */
#include "./vp9_rtcd.h"
#include "vpx_dsp/inv_txfm.h"

typedef struct {
  MACROBLOCKD *xd;
  const int32_t *const *eobs;
  const int *eob_strides;
  const int32_t *coef[3];
  int mi_row, mi_col, ss, lossless, intra, eobtotal;
} recon_arg;

static void ref_inverse_add(MACROBLOCKD *xd, TX_SIZE tx_size, TX_TYPE tx_type, const tran_low_t *dq, int eob, uint8_t *dst,
                            int stride, int lossless) {
  if (xd->cur_buf->flags & YV12_FLAG_HIGHBITDEPTH) {
    tran_high_t res[32 * 32];
    const int n = 4 << tx_size;
    uint16_t *d = CONVERT_TO_SHORTPTR(dst);
    switch (tx_size) {
      case TX_4X4:
        if (tx_type == DCT_DCT) vpx_highbd_idct4x4_16_add_c(dq, res, n, xd->bd);
        else vp9_highbd_iht4x4_16_add_c(dq, res, n, tx_type, xd->bd);
        break;
      case TX_8X8:
        if (tx_type == DCT_DCT) vpx_highbd_idct8x8_64_add_c(dq, res, n, xd->bd);
        else vp9_highbd_iht8x8_64_add_c(dq, res, n, tx_type, xd->bd);
        break;
      case TX_16X16:
        if (tx_type == DCT_DCT) vpx_highbd_idct16x16_256_add_c(dq, res, n, xd->bd);
        else vp9_highbd_iht16x16_256_add_c(dq, res, n, tx_type, xd->bd);
        break;
      default: vpx_highbd_idct32x32_1024_add_c(dq, res, n, xd->bd); break;
    }
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) d[y * stride + x] = highbd_clip_pixel_add(d[y * stride + x], res[y * n + x], xd->bd);
    return;
  }
  if (lossless) {
    if (eob > 1) vpx_iwht4x4_16_add_c(dq, dst, stride);
    else vpx_iwht4x4_1_add_c(dq, dst, stride);
    return;
  }
  switch (tx_size) {
    case TX_4X4:
      if (tx_type != DCT_DCT) vp9_iht4x4_16_add_c(dq, dst, stride, tx_type);
      else if (eob > 1) vpx_idct4x4_16_add_c(dq, dst, stride);
      else vpx_idct4x4_1_add_c(dq, dst, stride);
      break;
    case TX_8X8:
      if (tx_type != DCT_DCT) vp9_iht8x8_64_add_c(dq, dst, stride, tx_type);
      else if (eob == 1) vpx_idct8x8_1_add_c(dq, dst, stride);
      else if (eob <= 12) vpx_idct8x8_12_add_c(dq, dst, stride);
      else vpx_idct8x8_64_add_c(dq, dst, stride);
      break;
    case TX_16X16:
      if (tx_type != DCT_DCT) vp9_iht16x16_256_add_c(dq, dst, stride, tx_type);
      else if (eob == 1) vpx_idct16x16_1_add_c(dq, dst, stride);
      else if (eob <= 10) vpx_idct16x16_10_add_c(dq, dst, stride);
      else if (eob <= 38) vpx_idct16x16_38_add_c(dq, dst, stride);
      else vpx_idct16x16_256_add_c(dq, dst, stride);
      break;
    default:
      if (eob == 1) vpx_idct32x32_1_add_c(dq, dst, stride);
      else if (eob <= 34) vpx_idct32x32_34_add_c(dq, dst, stride);
      else if (eob <= 135) vpx_idct32x32_135_add_c(dq, dst, stride);
      else vpx_idct32x32_1024_add_c(dq, dst, stride);
      break;
  }
}

static void recon_visit(int plane, int block, int row, int col, BLOCK_SIZE plane_bsize, TX_SIZE tx_size, void *argp) {
  recon_arg *a = (recon_arg *)argp;
  MACROBLOCKD *xd = a->xd;
  struct macroblockd_plane *pd = &xd->plane[plane];
  const MODE_INFO *mi = xd->mi[0];
  const int stride = pd->dst.stride, s = plane ? a->ss : 0;
  PREDICTION_MODE mode = plane == 0 ? mi->mode : mi->uv_mode;
  uint8_t *dst;
  (void)block;
  (void)plane_bsize;
  if (xd->cur_buf->flags & YV12_FLAG_HIGHBITDEPTH)
    dst = CONVERT_TO_BYTEPTR(CONVERT_TO_SHORTPTR(pd->dst.buf) + 4 * row * stride + 4 * col);
  else
    dst = &pd->dst.buf[4 * row * stride + 4 * col];
  if (a->intra) {
    if (mi->sb_type < BLOCK_8X8 && plane == 0) mode = mi->bmi[(row << 1) + col].as_mode;
    vp9_predict_intra_block(xd, pd->n4_wl, tx_size, mode, dst, stride, dst, stride, col, row, plane);
  }
  if (!mi->skip) {
    const int x = ((a->mi_col * 8) >> s) + 4 * col, y = ((a->mi_row * 8) >> s) + 4 * row;
    const int eob = a->eobs[plane][(size_t)y * a->eob_strides[plane] + x];
    const TX_TYPE tx_type = (!a->intra || plane || a->lossless || tx_size == TX_32X32) ? DCT_DCT : intra_mode_to_tx_type_lookup[mode];
    if (eob > 0) ref_inverse_add(xd, tx_size, tx_type, (const tran_low_t *)a->coef[plane], eob, dst, stride, a->lossless);
    a->eobtotal += eob;
    a->coef[plane] += 16 << (tx_size << 1);
  }
}

int ref_recon_frame(const int32_t *blocks, int n_blocks, int w, int h, int ss, int bd, int hbd, void *const cur_planes[3],
                    const int cur_strides[3], void *const ref_planes[9], const int ref_strides[9], const int ref_w[3],
                    const int ref_h[3], const int32_t *const coefs[3], const int32_t *const eobs[3], const int eob_strides[3],
                    int log2_tile_cols, int lossless, int filter, int sharpness) {
  static int inited = 0;
  const int mi_rows = ((h + 7) & ~7) / 8, mi_cols = ((w + 7) & ~7) / 8;
  YV12_BUFFER_CONFIG cur, refbuf[3];
  RefBuffer rb[3];
  MODE_INFO dummy;
  VP9_COMMON *cm = (VP9_COMMON *)calloc(1, sizeof(*cm));
  MACROBLOCKD *xd = (MACROBLOCKD *)calloc(1, sizeof(*xd));
  uint8_t *lf_skip = (uint8_t *)malloc((size_t)n_blocks + 1);
  recon_arg arg;
  if (!cm || !xd || !lf_skip) return -1;
  if (!inited) {
    vp9_init_intra_predictors();
    inited = 1;
  }
  memset(&dummy, 0, sizeof(dummy));
  memset(&arg, 0, sizeof(arg));
  cm->mi_rows = mi_rows;
  cm->mi_cols = mi_cols;
  cm->log2_tile_cols = log2_tile_cols;
  fill_yv12(&cur, cur_planes, cur_strides, w, h, ss, hbd, bd);
  memset(rb, 0, sizeof(rb));
  for (int k = 0; k < 3; ++k) {
    if (ref_w[k] <= 0) continue;
    fill_yv12(&refbuf[k], &ref_planes[3 * k], &ref_strides[3 * k], ref_w[k], ref_h[k], ss, hbd, bd);
    rb[k].buf = &refbuf[k];
    rb[k].idx = k;
    vp9_setup_scale_factors_for_frame(&rb[k].sf, ref_w[k], ref_h[k], w, h, hbd);
    if (!vp9_is_valid_scale(&rb[k].sf)) return -2;
  }
  xd->cur_buf = &cur;
  xd->bd = bd;
  arg.xd = xd;
  arg.eobs = eobs;
  arg.eob_strides = eob_strides;
  arg.ss = ss;
  arg.lossless = lossless;
  for (int p = 0; p < 3; ++p) arg.coef[p] = coefs[p];
  for (int i = 0; i < n_blocks; ++i) {
    const int32_t *b = blocks + REC * i;
    MODE_INFO mi, *mip = &mi;
    TileInfo tile;
    fill_mi(&mi, b);
    const BLOCK_SIZE bsize = mi.sb_type < BLOCK_8X8 ? BLOCK_8X8 : mi.sb_type;
    const int bw8 = num_8x8_blocks_wide_lookup[bsize], bh8 = num_8x8_blocks_high_lookup[bsize];
    xd->mi = &mip;
    set_block_geometry(xd, b[0], b[1], bw8, bh8, mi_rows, mi_cols, ss);
    for (int t = 0; t < (1 << log2_tile_cols); ++t) {
      vp9_tile_set_col(&tile, cm, t);
      if (b[1] >= tile.mi_col_start && b[1] < tile.mi_col_end) break;
    }
    xd->above_mi = (b[0] != 0) ? &dummy : NULL;
    xd->left_mi = (b[1] > tile.mi_col_start) ? &dummy : NULL;
    vp9_setup_dst_planes(xd->plane, &cur, b[0], b[1]);
    arg.mi_row = b[0];
    arg.mi_col = b[1];
    arg.intra = b[6] <= 0;
    arg.eobtotal = 0;
    if (!arg.intra) {
      for (int r = 0; r < 1 + has_second_ref(&mi); ++r) {
        RefBuffer *ref_buf = &rb[mi.ref_frame[r] - LAST_FRAME];
        xd->block_refs[r] = ref_buf;
        vp9_setup_pre_planes(xd, r, ref_buf->buf, b[0], b[1], &ref_buf->sf);
      }
      vp9_build_inter_predictors_sb(xd, b[0], b[1], bsize);
    }
    if (arg.intra || !mi.skip)
      for (int plane = 0; plane < MAX_MB_PLANE; ++plane)
        vp9_foreach_transformed_block_in_plane(xd, bsize, plane, recon_visit, &arg);
    lf_skip[i] = (uint8_t)(mi.skip || (!arg.intra && mi.sb_type >= BLOCK_8X8 && arg.eobtotal == 0));
  }
  free(xd);
  free(cm);
  if (filter) {
    /* masks + filtering exactly as ref_lf_frame2, with the skip flags as libvpx's loop filter sees them */
    int32_t *b2 = (int32_t *)malloc(sizeof(int32_t) * REC * (size_t)(n_blocks + 1));
    void *lfm = malloc((size_t)((mi_rows + 7) / 8) * ((mi_cols + 7) / 8) * sizeof(LOOP_FILTER_MASK));
    if (!b2 || !lfm) return -1;
    memcpy(b2, blocks, sizeof(int32_t) * REC * (size_t)n_blocks);
    for (int i = 0; i < n_blocks; ++i) b2[REC * i + 4] = lf_skip[i];
    const int rc = ref_lf_frame3(b2, n_blocks, mi_cols * 8, mi_rows * 8, cur_planes, cur_strides, bd, hbd, sharpness, lfm, 1, ss);
    free(b2);
    free(lfm);
    if (rc < 0) return rc;
  }
  free(lf_skip);
  return 0;
}
