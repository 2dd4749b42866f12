/*
 * ref_stream_wraps.c — STREAM-LEVEL ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU bodies of the reference's two device entry points
 *     wrap_cuda_inter_prediction / wrap_cuda_intra_prediction
 * (/root/reference/vpx-master/cuda_extern_wrap.cpp:5-17; called from decode_tiles,
 * libvpx/vp9/decoder/vp9_decodeframe.c:2546, :2564), written on top of the reference's OWN C
 * functions, so that the reference's vpxdec can be linked without CUDA and decode a bitstream
 * on the CPU.  It is the checker the HIP path's per-frame MD5s are compared with; only
 * oracle/build_refvpx.sh links it (into oracle/_ref/vpx/vpxdec_c* and vpxenc_c).
 *
 * Every sample is produced by a reference function:
 *   inter   vp9_build_inter_predictors_sb           libvpx/vp9/common/vp9_reconinter.c:253
 *           (stock semantics: compound, scaled references, clamp_mv_to_umv_border_sb, sub-8x8
 *           MV averaging) reading private copies of the reference frames with libvpx's
 *           encoder-side 160-sample replicated border (vpx_extend_frame_borders_c,
 *           libvpx/vpx_scale/generic/yv12extend.c:169) — the scheme libvpx itself holds equivalent
 *           to the decoder's on-the-fly border emulation (vp9_reconinter.c:93-95); the decoder-
 *           side builder dec_build_inter_predictors is an empty shell in this fork
 *           (vp9_decodeframe.c:556-560), which is why SURVEY §8(c) prescribes this one
 *   intra   vp9_predict_intra_block                 libvpx/vp9/common/vp9_reconintra.c:404
 *           in the order of the retained CPU routine intra_predict_and_reconstruct
 *           (vp9_decodeframe.c:1073-1115)
 *   residual, as the caller left it:
 *     - unchanged caller (its CPU phase B ran, vp9_decodeframe.c:2443-2486): int64 residual planes
 *       added with highbd_clip_pixel_add, the arithmetic of block_sum (:290-341) /
 *       inter_residual_sum (:1117-1148).  High-bitdepth buffers only, like the reference.
 *     - after vp9hip_shim_attach_frame_buffer() (INTEGRATION.md mode B: phase B deleted): inverse
 *       transforms from frameBuf.dqcoeff / plane_eob in detoken_block's order (:919-1024) through
 *       vp9_idct*_add / vp9_iht*_add / vp9_iwht4x4_add (8-bit buffers,
 *       libvpx/vp9/common/vp9_idct.c:119-204) or the fork's residual-storing highbd *_c
 *       functions + highbd_clip_pixel_add (the composition its phase B + block_sum make), and
 *       libvpx's `if (!less8x8 && eobtotal == 0) mi->skip = 1` (:1195)
 *   loop filter, after vp9hip_shim_set_gpu_loop_filter(pbi, 1) (mode C: phase E deleted):
 *           vp9_build_mask_frame + vp9_loop_filter_frame (libvpx/vp9/common/vp9_loopfilter.c:1490,
 *           :1471) — what the reference's encoder runs (vp9/encoder/vp9_encoder.c:3364-3372).
 *
 * The three vp9hip_shim_* hooks have the shim's names and prototypes on purpose: ONE patched
 * vp9_decodeframe object links against either this file (CPU, oracle) or libvp9hip_shim.so (HIP).
 *
 * Pinning: oracle/_ref/vpx/vpxenc_c is the reference's encoder linked with THIS decoder;
 * `vpxenc --test-decode=fatal` compares every decoded frame with the encoder's own (independent,
 * stock) reconstruction — tests/golden/streams/make_streams.sh runs it for every committed stream.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "./vp9_rtcd.h"
#include "./vpx_dsp_rtcd.h"
#include "./vpx_scale_rtcd.h"
#include "buffers_struct.h"
#include "vp9/common/vp9_idct.h"
#include "vp9/common/vp9_loopfilter.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/common/vp9_reconinter.h"
#include "vp9/common/vp9_reconintra.h"
#include "vp9/common/vp9_scan.h"
#include "vp9/common/vp9_tile_common.h"
#include "vp9/decoder/vp9_decoder.h"
#include "vpx_dsp/inv_txfm.h"
#include "vpx_scale/yv12config.h"

#define ORACLE_MAX_DECODERS 4

typedef struct {
  VP9Decoder *pbi;
  const frameBuf *attached;
  int eob_shift;
  tran_low_t *dq_start[3];
  int filter_here;
  /* bordered copies of finished frames, one per frame-buffer index */
  YV12_BUFFER_CONFIG ext[FRAME_BUFFERS];
  struct {
    const uint8_t *alloc;
    int w, h, valid;
  } tag[FRAME_BUFFERS];
  RefBuffer ext_ref[3];
  /* mi->skip as PARSED, per block of the list: the inter pass may turn a block into skip
   * (eobtotal == 0) whose coefficient slots the intra pass still has to step over */
  uint8_t *parsed_skip;
  int parsed_cap, parsed_valid;
} oracle_state;

static oracle_state g_or[ORACLE_MAX_DECODERS];

/* VP9_ORACLE_FAULT=lf|res|intra seeds a deliberate bug (no loop filter / inter residual dropped / one
 * intra mode replaced) so that tests can show `vpxenc --test-decode=fatal` and the MD5 comparison
 * actually notice a wrong decoder. */
static int fault(const char *name) {
  const char *f = getenv("VP9_ORACLE_FAULT");
  return f && !strcmp(f, name);
}

static void die(const char *msg) {
  fprintf(stderr, "stream oracle: %s\n", msg);
  exit(3);
}

static oracle_state *state_of(VP9Decoder *pbi) {
  int free_i = -1;
  for (int i = 0; i < ORACLE_MAX_DECODERS; ++i) {
    if (g_or[i].pbi == pbi) return &g_or[i];
    if (!g_or[i].pbi && free_i < 0) free_i = i;
  }
  if (free_i < 0) die("too many decoder instances");
  memset(&g_or[free_i], 0, sizeof(g_or[free_i]));
  g_or[free_i].pbi = pbi;
  return &g_or[free_i];
}

void vp9hip_shim_attach_frame_buffer(struct VP9Decoder *pbi, const struct frame_buffer *frameBuffer) {
  oracle_state *s = state_of(pbi);
  s->attached = frameBuffer;
  if (frameBuffer)
    for (int p = 0; p < 3; ++p) s->dq_start[p] = frameBuffer->dqcoeff[p];
}

/* the shim's hook of the same name (include/vp9hip_libvpx_shim.h): here plain memory kept per decoder */
void *vp9hip_shim_frame_memory(struct VP9Common *cm, int which, size_t bytes) {
  static struct {
    VP9_COMMON *cm;
    void *p[4];
    size_t cap[4];
  } mem[ORACLE_MAX_DECODERS];
  if (which < 0 || which > 3) return NULL;
  for (int i = 0; i < ORACLE_MAX_DECODERS; ++i) {
    if (mem[i].cm != cm && mem[i].cm != NULL) continue;
    mem[i].cm = cm;
    if (bytes > mem[i].cap[which]) {
      free(mem[i].p[which]);
      mem[i].p[which] = malloc(bytes + bytes / 8);
      mem[i].cap[which] = mem[i].p[which] ? bytes + bytes / 8 : 0;
    }
    return mem[i].p[which];
  }
  return NULL;
}

/* the tile-parallel hooks exist for the shim only; the oracle is linked with the serial driver */
void vp9hip_shim_run_parallel(struct VP9Decoder *pbi, int n, void (*fn)(void *arg, int index), void *arg) {
  (void)pbi;
  for (int i = 0; i < n; ++i) fn(arg, i);
}

void vp9hip_shim_set_eob_layout(struct VP9Decoder *pbi, int log2_granularity) {
  state_of(pbi)->eob_shift = log2_granularity == 2 ? 2 : 0;
}

void vp9hip_shim_mark(struct VP9Decoder *pbi, int mark) {
  (void)pbi;
  (void)mark;
}

void vp9hip_shim_set_gpu_loop_filter(struct VP9Decoder *pbi, int enable) { state_of(pbi)->filter_here = enable != 0; }

void vp9hip_shim_release(struct VP9Decoder *pbi) {
  for (int i = 0; i < ORACLE_MAX_DECODERS; ++i)
    if (g_or[i].pbi == pbi) {
      for (int k = 0; k < FRAME_BUFFERS; ++k) vpx_free_frame_buffer(&g_or[i].ext[k]);
      free(g_or[i].parsed_skip);
      memset(&g_or[i], 0, sizeof(g_or[i]));
    }
}

/* a bordered (VP9_ENC_BORDER_IN_PIXELS) copy of a finished frame */
static const YV12_BUFFER_CONFIG *bordered_copy(oracle_state *s, VP9_COMMON *cm, int idx, const YV12_BUFFER_CONFIG *src) {
  YV12_BUFFER_CONFIG *dst = &s->ext[idx];
  if (s->tag[idx].valid && s->tag[idx].alloc == src->buffer_alloc && s->tag[idx].w == src->y_crop_width &&
      s->tag[idx].h == src->y_crop_height)
    return dst;
  const int hbd = (src->flags & YV12_FLAG_HIGHBITDEPTH) != 0;
  if (vpx_realloc_frame_buffer(dst, src->y_crop_width, src->y_crop_height, src->subsampling_x, src->subsampling_y, hbd,
                               VP9_ENC_BORDER_IN_PIXELS, cm->byte_alignment, NULL, NULL, NULL))
    die("out of memory");
  const int bps = hbd ? 2 : 1;
  const uint8_t *sp[3] = { src->y_buffer, src->u_buffer, src->v_buffer };
  uint8_t *dp[3] = { dst->y_buffer, dst->u_buffer, dst->v_buffer };
  for (int p = 0; p < 3; ++p) {
    const int w = p ? src->uv_width : src->y_width, h = p ? src->uv_height : src->y_height;
    const int ss = p ? src->uv_stride : src->y_stride, ds = p ? dst->uv_stride : dst->y_stride;
    const uint8_t *a = hbd ? (const uint8_t *)CONVERT_TO_SHORTPTR(sp[p]) : sp[p];
    uint8_t *b = hbd ? (uint8_t *)CONVERT_TO_SHORTPTR(dp[p]) : dp[p];
    for (int y = 0; y < h; ++y) memcpy(b + (size_t)y * ds * bps, a + (size_t)y * ss * bps, (size_t)w * bps);
  }
  vpx_extend_frame_borders_c(dst);
  s->tag[idx].alloc = src->buffer_alloc;
  s->tag[idx].w = src->y_crop_width;
  s->tag[idx].h = src->y_crop_height;
  s->tag[idx].valid = 1;
  return dst;
}

/* set_offsets of the caller without the parse-time writes (vp9_decodeframe.c:830-857) */
static MODE_INFO *block_offsets(VP9_COMMON *cm, MACROBLOCKD *xd, int mi_row, int mi_col, int bwl, int bhl) {
  const int bw = 1 << (bwl - 1), bh = 1 << (bhl - 1);
  xd->mi = cm->mi_grid_visible + mi_row * cm->mi_stride + mi_col;
  for (int i = 0; i < MAX_MB_PLANE; i++) {
    xd->plane[i].n4_w = (bw << 1) >> xd->plane[i].subsampling_x;
    xd->plane[i].n4_h = (bh << 1) >> xd->plane[i].subsampling_y;
    xd->plane[i].n4_wl = bwl - xd->plane[i].subsampling_x;
    xd->plane[i].n4_hl = bhl - xd->plane[i].subsampling_y;
  }
  set_mi_row_col(xd, &xd->tile, mi_row, bh, mi_col, bw, cm->mi_rows, cm->mi_cols);
  vp9_setup_dst_planes(xd->plane, get_frame_new_buffer(cm), mi_row, mi_col);
  return xd->mi[0];
}

typedef struct {
  int plane, row, col, n;
  TX_SIZE tx_size;
  uint8_t *dst;
  int stride;
} txb_pos;

/* the transform-block visit of detoken_block / inter_decode / intra_decode (vp9_decodeframe.c:919-1196) */
#define FOREACH_TXB(xd, mi, T, BODY)                                                                                   \
  for (int plane_ = 0; plane_ < MAX_MB_PLANE; ++plane_) {                                                              \
    const struct macroblockd_plane *const pd_ = &(xd)->plane[plane_];                                                  \
    const TX_SIZE txs_ = plane_ ? get_uv_tx_size(mi, pd_) : (mi)->tx_size;                                             \
    const int step_ = 1 << txs_;                                                                                       \
    const int mbw_ = pd_->n4_w + ((xd)->mb_to_right_edge >= 0 ? 0 : (xd)->mb_to_right_edge >> (5 + pd_->subsampling_x)); \
    const int mbh_ = pd_->n4_h + ((xd)->mb_to_bottom_edge >= 0 ? 0 : (xd)->mb_to_bottom_edge >> (5 + pd_->subsampling_y)); \
    (xd)->max_blocks_wide = (xd)->mb_to_right_edge >= 0 ? 0 : mbw_;                                                    \
    (xd)->max_blocks_high = (xd)->mb_to_bottom_edge >= 0 ? 0 : mbh_;                                                   \
    for (int row_ = 0; row_ < mbh_; row_ += step_)                                                                     \
      for (int col_ = 0; col_ < mbw_; col_ += step_) {                                                                 \
        txb_pos T;                                                                                                     \
        T.plane = plane_;                                                                                              \
        T.row = row_;                                                                                                  \
        T.col = col_;                                                                                                  \
        T.tx_size = txs_;                                                                                              \
        T.n = 16 << (txs_ << 1);                                                                                       \
        T.stride = pd_->dst.stride;                                                                                    \
        T.dst = &pd_->dst.buf[4 * row_ * T.stride + 4 * col_];                                                         \
        BODY                                                                                                           \
      }                                                                                                                \
  }

static void add_residual_plane(const MACROBLOCKD *xd, const txb_pos *t, const tran_high_t *res_plane, int mi_row,
                               int mi_col) {
  const struct macroblockd_plane *pd = &xd->plane[t->plane];
  const int by = (mi_row * MI_SIZE) >> pd->subsampling_y, bx = (mi_col * MI_SIZE) >> pd->subsampling_x;
  const tran_high_t *r = res_plane + (size_t)(by + 4 * t->row) * t->stride + bx + 4 * t->col;
  uint16_t *d = CONVERT_TO_SHORTPTR(t->dst);
  const int n = 4 << t->tx_size;
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) d[y * t->stride + x] = highbd_clip_pixel_add(d[y * t->stride + x], r[y * t->stride + x], xd->bd);
}

/* inverse transform + add of one block from its dequantised coefficients */
static void inverse_add(MACROBLOCKD *xd, const txb_pos *t, TX_TYPE tx_type, const tran_low_t *dq, int eob) {
  if (xd->cur_buf->flags & YV12_FLAG_HIGHBITDEPTH) {
    tran_high_t res[32 * 32];
    const int n = 4 << t->tx_size;
    uint16_t *d = CONVERT_TO_SHORTPTR(t->dst);
    if (xd->lossless) die("lossless high-bitdepth streams: the fork's vpx_highbd_iwht4x4_16_add_c is broken (inv_txfm.c:1346-1352)");
    memset(res, 0, sizeof(res[0]) * n * n);
    switch (t->tx_size) {
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
      for (int x = 0; x < n; ++x) d[y * t->stride + x] = highbd_clip_pixel_add(d[y * t->stride + x], res[y * n + x], xd->bd);
    return;
  }
  if (xd->lossless) {
    vp9_iwht4x4_add(dq, t->dst, t->stride, eob);
    return;
  }
  switch (t->tx_size) {
    case TX_4X4: vp9_iht4x4_add(tx_type, dq, t->dst, t->stride, eob); break;
    case TX_8X8: vp9_iht8x8_add(tx_type, dq, t->dst, t->stride, eob); break;
    case TX_16X16: vp9_iht16x16_add(tx_type, dq, t->dst, t->stride, eob); break;
    default: vp9_idct32x32_add(dq, t->dst, t->stride, eob); break;
  }
}

static int eob_at(const oracle_state *s, const MACROBLOCKD *xd, const txb_pos *t, int mi_row, int mi_col) {
  const struct macroblockd_plane *pd = &xd->plane[t->plane];
  const int by = (mi_row * MI_SIZE) >> pd->subsampling_y, bx = (mi_col * MI_SIZE) >> pd->subsampling_x;
  const int sh = s->eob_shift;
  return s->attached->plane_eob[t->plane][(size_t)((by + 4 * t->row) >> sh) * (t->stride >> sh) + ((bx + 4 * t->col) >> sh)];
}

typedef void (*block_fn)(oracle_state *s, VP9_COMMON *cm, MACROBLOCKD *xd, MODE_INFO *mi, int index, int mi_row, int mi_col,
                         const tran_high_t *const res[3], tran_low_t *dq[3]);

/* the caller's walk over its block list (vp9_decodeframe.c:2443-2486) */
static void walk_blocks(oracle_state *s, VP9_COMMON *cm, VP9Decoder *pbi, const int *size_for_mb, const ModeInfoBuf *MiBuf,
                        int tile_rows, int tile_cols, const tran_high_t *const res[3], block_fn fn) {
  tran_low_t *dq[3] = { s->dq_start[0], s->dq_start[1], s->dq_start[2] };
  int i = 0;
  for (int tile_row = 0; tile_row < tile_rows; ++tile_row) {
    TileInfo tile;
    vp9_tile_set_row(&tile, cm, tile_row);
    for (int mi_row = tile.mi_row_start; mi_row < tile.mi_row_end; mi_row += MI_BLOCK_SIZE)
      for (int tile_col = 0; tile_col < tile_cols; ++tile_col) {
        const int col = pbi->inv_tile_order ? tile_cols - tile_col - 1 : tile_col;
        TileWorkerData *td = pbi->tile_worker_data + tile_cols * tile_row + col;
        vp9_tile_set_col(&tile, cm, col);
        for (int mi_col = tile.mi_col_start; mi_col < tile.mi_col_end; mi_col += MI_BLOCK_SIZE) {
          for (int k = 0; k < *size_for_mb; ++k, ++i) {
            MODE_INFO *mi = block_offsets(cm, &td->xd, MiBuf->mi_row[i], MiBuf->mi_col[i], MiBuf->bwl[i], MiBuf->bhl[i]);
            if (mi != MiBuf->mi[i]) die("block list and mode-info grid disagree");
            fn(s, cm, &td->xd, mi, i, MiBuf->mi_row[i], MiBuf->mi_col[i], res, dq);
          }
          ++size_for_mb;
        }
      }
  }
}

static void skip_coeff_slots(MACROBLOCKD *xd, MODE_INFO *mi, tran_low_t *dq[3]) {
  FOREACH_TXB(xd, mi, t, { dq[t.plane] += t.n; })
}

static void inter_block(oracle_state *s, VP9_COMMON *cm, MACROBLOCKD *xd, MODE_INFO *mi, int index, int mi_row, int mi_col,
                        const tran_high_t *const res[3], tran_low_t *dq[3]) {
  s->parsed_skip[index] = (uint8_t)(mi->skip != 0);
  if (!is_inter_block(mi)) {
    if (s->attached && !mi->skip) skip_coeff_slots(xd, mi, dq);
    return;
  }
  for (int ref = 0; ref < 1 + has_second_ref(mi); ++ref) {
    const int k = mi->ref_frame[ref] - LAST_FRAME;
    RefBuffer *rb = &cm->frame_refs[k];
    if (!vp9_is_valid_scale(&rb->sf)) die("reference frame has invalid dimensions");
    s->ext_ref[k].idx = rb->idx;
    s->ext_ref[k].sf = rb->sf;
    s->ext_ref[k].buf = (YV12_BUFFER_CONFIG *)bordered_copy(s, cm, rb->idx, rb->buf);
    xd->block_refs[ref] = &s->ext_ref[k];
    vp9_setup_pre_planes(xd, ref, s->ext_ref[k].buf, mi_row, mi_col, &rb->sf);
  }
  vp9_build_inter_predictors_sb(xd, mi_row, mi_col, VPXMAX(mi->sb_type, BLOCK_8X8));
  if (mi->skip || fault("res")) return;
  if (s->attached) {
    int eobtotal = 0;
    FOREACH_TXB(xd, mi, t, {
      const int eob = eob_at(s, xd, &t, mi_row, mi_col);
      if (eob > 0) inverse_add(xd, &t, DCT_DCT, dq[t.plane], eob);
      eobtotal += eob;
      dq[t.plane] += t.n;
    })
    if (mi->sb_type >= BLOCK_8X8 && eobtotal == 0) mi->skip = 1; /* vp9_decodeframe.c:1195 */
  } else {
    FOREACH_TXB(xd, mi, t, { add_residual_plane(xd, &t, res[t.plane], mi_row, mi_col); })
  }
}

static void intra_block(oracle_state *s, VP9_COMMON *cm, MACROBLOCKD *xd, MODE_INFO *mi, int index, int mi_row, int mi_col,
                        const tran_high_t *const res[3], tran_low_t *dq[3]) {
  (void)cm;
  if (is_inter_block(mi)) {
    const int skip_as_parsed = s->parsed_valid ? s->parsed_skip[index] : (mi->skip != 0);
    if (s->attached && !skip_as_parsed) skip_coeff_slots(xd, mi, dq);
    return;
  }
  FOREACH_TXB(xd, mi, t, {
    PREDICTION_MODE mode = t.plane ? mi->uv_mode : mi->mode;
    if (mi->sb_type < BLOCK_8X8 && t.plane == 0) mode = mi->bmi[(t.row << 1) + t.col].as_mode;
    if (mode == D45_PRED && fault("intra")) mode = D63_PRED;
    vp9_predict_intra_block(xd, xd->plane[t.plane].n4_wl, t.tx_size, mode, t.dst, t.stride, t.dst, t.stride, t.col, t.row,
                            t.plane);
    if (!mi->skip) {
      if (s->attached) {
        const int eob = eob_at(s, xd, &t, mi_row, mi_col);
        const TX_TYPE tx_type = (t.plane || xd->lossless) ? DCT_DCT : intra_mode_to_tx_type_lookup[mode];
        if (eob > 0) inverse_add(xd, &t, tx_type, dq[t.plane], eob);
        dq[t.plane] += t.n;
      } else {
        add_residual_plane(xd, &t, res[t.plane], mi_row, mi_col);
      }
    }
  })
}

/* VP9_ORACLE_DUMP_BLOCKS=<file>: the parsed mode information of every frame, one 64-byte record per block in decode
 * order (the layout of vp9hip_block, include/vp9hip_pack.h) — what tests/test_fe_blocks.py compares the product's
 * own bitstream front-end with.  reserved[0] segment id, reserved[2] the skip flag as parsed, reserved2: a
 * checksum over the block's eobs and coefficients. */
static FILE *g_dump;
static void dump_block(oracle_state *s, VP9_COMMON *cm, MACROBLOCKD *xd, MODE_INFO *mi, int index, int mi_row, int mi_col,
                       const tran_high_t *const res[3], tran_low_t *dq[3]) {
  unsigned char rec[64];
  (void)res;
  memset(rec, 0, sizeof(rec));
  const int parsed_skip = s->parsed_valid ? s->parsed_skip[index] : (mi->skip != 0);
  const short pos[2] = { (short)mi_row, (short)mi_col };
  memcpy(rec, pos, 4);
  rec[4] = (unsigned char)mi->sb_type;
  rec[5] = (unsigned char)mi->tx_size;
  rec[6] = (unsigned char)(mi->skip != 0);
  rec[7] = (unsigned char)mi->interp_filter;
  rec[8] = (unsigned char)mi->ref_frame[0];
  rec[9] = (unsigned char)mi->ref_frame[1];
  rec[10] = (unsigned char)mi->mode;
  rec[11] = (unsigned char)mi->uv_mode;
  for (int i = 0; i < 4; ++i) rec[12 + i] = (unsigned char)mi->bmi[i].as_mode;
  {
    static const unsigned char lut[14] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 1 }; /* mode_lf_lut, vp9_loopfilter.c:207 */
    rec[16] = cm->lf.filter_level ? cm->lf_info.lvl[mi->segment_id][mi->ref_frame[0]][lut[mi->mode]] : 0;
  }
  rec[17] = (unsigned char)mi->segment_id;
  rec[19] = (unsigned char)parsed_skip;
  for (int r = 0; r < 2; ++r) {
    const short mv[2] = { mi->mv[r].as_mv.row, mi->mv[r].as_mv.col };
    memcpy(rec + 20 + 4 * r, mv, 4);
  }
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 2; ++r) {
      const short mv[2] = { mi->bmi[i].as_mv[r].as_mv.row, mi->bmi[i].as_mv[r].as_mv.col };
      memcpy(rec + 28 + 8 * i + 4 * r, mv, 4);
    }
  unsigned cs = 0;
  if (s->attached && !parsed_skip) {
    FOREACH_TXB(xd, mi, t, {
      const int eob = eob_at(s, xd, &t, mi_row, mi_col);
      unsigned sum = 0;
      if (eob > 0)
        for (int i = 0; i < t.n; ++i) sum += (unsigned)dq[t.plane][i] * (unsigned)(i + 1);
      cs = cs * 1000003u + sum + (unsigned)eob * 7919u;
      dq[t.plane] += t.n;
    })
  }
  memcpy(rec + 60, &cs, 4);
  fwrite(rec, 1, sizeof(rec), g_dump);
}

static void reserve_parsed(oracle_state *s, const VP9_COMMON *cm, const int *size_for_mb) {
  const int n_sb = ((cm->mi_rows + 7) >> 3) * ((cm->mi_cols + 7) >> 3);
  int n = 0;
  for (int i = 0; i < n_sb; ++i) n += size_for_mb[i];
  if (n > s->parsed_cap) {
    free(s->parsed_skip);
    s->parsed_cap = n + 1024;
    s->parsed_skip = (uint8_t *)malloc((size_t)s->parsed_cap);
    if (!s->parsed_skip) die("out of memory");
  }
}

/* initBuf's plane pointers (vp9_decodeframe.c:2242-2262) for the inter entry point, which is handed
 * the base of the residual allocation only */
static void residual_planes(const YV12_BUFFER_CONFIG *cur, int byte_alignment, const tran_high_t *residuals,
                            const tran_high_t *out[3]) {
  const int uv_border_h = cur->border >> cur->subsampling_y, uv_border_w = cur->border >> cur->subsampling_x;
  const int align = byte_alignment == 0 ? 1 : byte_alignment;
  const uint64_t yplane_size = (cur->y_height + 2 * cur->border) * (uint64_t)cur->y_stride + byte_alignment;
  const uint64_t uvplane_size = (cur->uv_height + 2 * uv_border_h) * (uint64_t)cur->uv_stride + byte_alignment;
  out[0] = (const tran_high_t *)yv12_align_addr(residuals + (cur->border * cur->y_stride) + cur->border, align);
  out[1] = (const tran_high_t *)yv12_align_addr(residuals + yplane_size + (uv_border_h * cur->uv_stride) + uv_border_w, align);
  out[2] = (const tran_high_t *)yv12_align_addr(
      residuals + yplane_size + uvplane_size + (uv_border_h * cur->uv_stride) + uv_border_w, align);
}

int wrap_cuda_inter_prediction(int n, double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf,
                               VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols, tran_high_t *residuals) {
  oracle_state *s = state_of(pbi);
  const YV12_BUFFER_CONFIG *cur = get_frame_new_buffer(cm);
  const tran_high_t *res[3] = { NULL, NULL, NULL };
  (void)n;
  s->tag[cm->new_fb_idx].valid = 0;
  if (!s->attached) {
    if (!(cur->flags & YV12_FLAG_HIGHBITDEPTH)) die("residual-plane mode needs a high-bitdepth frame buffer (as the reference does)");
    residual_planes(cur, cm->byte_alignment, residuals, res);
  }
  reserve_parsed(s, cm, size_for_mb);
  walk_blocks(s, cm, pbi, size_for_mb, MiBuf, tile_rows, tile_cols, res, inter_block);
  s->parsed_valid = 1;
  if (gpu_copy) *gpu_copy = 0.0;
  if (gpu_run) *gpu_run = 0.0;
  return 0;
}

int wrap_cuda_intra_prediction(double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf, VP9_COMMON *cm,
                               VP9Decoder *pbi, int tile_rows, int tile_cols, frameBuf *frameBuffer) {
  oracle_state *s = state_of(pbi);
  YV12_BUFFER_CONFIG *cur = get_frame_new_buffer(cm);
  const tran_high_t *res[3] = { NULL, NULL, NULL };
  s->tag[cm->new_fb_idx].valid = 0;
  if (!s->attached) {
    if (!(cur->flags & YV12_FLAG_HIGHBITDEPTH)) die("residual-plane mode needs a high-bitdepth frame buffer (as the reference does)");
    for (int p = 0; p < 3; ++p) res[p] = frameBuffer->plane_residuals[p];
  }
  if (!g_dump && getenv("VP9_ORACLE_DUMP_BLOCKS")) g_dump = fopen(getenv("VP9_ORACLE_DUMP_BLOCKS"), "wb");
  if (g_dump) {
    const int n_sb = ((cm->mi_rows + 7) >> 3) * ((cm->mi_cols + 7) >> 3);
    int hdr[4] = { 0x56503946, 0, cm->width, cm->height };
    for (int i = 0; i < n_sb; ++i) hdr[1] += size_for_mb[i];
    fwrite(hdr, sizeof(int), 4, g_dump);
    walk_blocks(s, cm, pbi, size_for_mb, MiBuf, tile_rows, tile_cols, res, dump_block);
    fflush(g_dump);
  }
  walk_blocks(s, cm, pbi, size_for_mb, MiBuf, tile_rows, tile_cols, res, intra_block);
  s->parsed_valid = 0;
  if (s->filter_here && cm->lf.filter_level && !cm->skip_loop_filter && !fault("lf")) {
    vp9_build_mask_frame(cm, cm->lf.filter_level, 0);
    vp9_loop_filter_frame(cur, cm, &pbi->mb, cm->lf.filter_level, 0, 0);
  }
  if (gpu_copy) *gpu_copy = 0.0;
  if (gpu_run) *gpu_run = 0.0;
  return 0;
}
