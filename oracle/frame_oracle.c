/*
 * frame_oracle.c — sequential CPU reconstruction of one frame from the same packed work
 * lists the HIP path consumes (include/vp9hip.h), built only from the oracle's own block
 * functions.  TEST INFRASTRUCTURE ONLY (see vp9_oracle.h): the checker for whole-frame
 * parity and the "port" CPU baseline of bench.py.
 *
 * Phase order = the reference's decode_tiles (vp9/decoder/vp9_decodeframe.c:2536-2620):
 * inter prediction (+ residual of inter blocks), then intra prediction + residual in decode
 * order, then the loop filter.  The list record layouts are duplicated here on purpose so
 * that the oracle does not include product headers.
 */
#include <string.h>

#include "vp9_oracle.h"

typedef struct {
  uint32_t coeff_off;
  uint16_t x, y;
  uint8_t plane, tx_size, tx_type, reserved;
  uint16_t eob, reserved2;
} o_txb;

typedef struct {
  int16_t dst_x, dst_y;
  uint8_t w, h, plane, flags;
  int32_t pos_x[2], pos_y[2];
  uint8_t ref[2], step_x[2], step_y[2], reserved[2];
} o_inter;

typedef struct {
  uint32_t coeff_off;
  uint16_t x, y;
  uint8_t plane, tx_size, tx_type, mode;
  uint16_t eob;
  uint8_t flags, reserved;
} o_intra;

static void txb_add(const vp9o_frame *f, int plane, int x, int y, int tx_size, int tx_type, int eob,
                    const int32_t *c) {
  const int n = 4 << tx_size;
  const int lossless = tx_type >> 7;
  /* blocks may overhang the aligned plane: work on a temporary and copy the visible part */
  const int vw = f->awidth[plane] - x < n ? f->awidth[plane] - x : n;
  const int vh = f->aheight[plane] - y < n ? f->aheight[plane] - y : n;
  if (vw <= 0 || vh <= 0) return;
  if (f->hbd) {
    uint16_t tmp[32 * 32];
    uint16_t *p = (uint16_t *)f->plane[plane] + (size_t)y * f->stride[plane] + x;
    memset(tmp, 0, sizeof(tmp));
    for (int r = 0; r < vh; ++r) memcpy(tmp + r * n, p + (size_t)r * f->stride[plane], vw * 2);
    vp9o_highbd_inv_txfm_add(n, tx_type & 3, lossless, c, tmp, n, eob, f->bit_depth);
    for (int r = 0; r < vh; ++r) memcpy(p + (size_t)r * f->stride[plane], tmp + r * n, vw * 2);
  } else {
    uint8_t tmp[32 * 32];
    uint8_t *p = (uint8_t *)f->plane[plane] + (size_t)y * f->stride[plane] + x;
    memset(tmp, 0, sizeof(tmp));
    for (int r = 0; r < vh; ++r) memcpy(tmp + r * n, p + (size_t)r * f->stride[plane], vw);
    vp9o_inv_txfm_add(n, tx_type & 3, lossless, c, tmp, n, eob);
    for (int r = 0; r < vh; ++r) memcpy(p + (size_t)r * f->stride[plane], tmp + r * n, vw);
  }
}

void vp9o_recon_inter_list(const void *tasks_, int n, const vp9o_frame *refs, const vp9o_frame *dst) {
  const o_inter *tasks = (const o_inter *)tasks_;
  for (int i = 0; i < n; ++i) {
    const o_inter *t = &tasks[i];
    const int pl = t->plane, filt = (t->flags >> 1) & 7, nref = (t->flags & 1) ? 2 : 1;
    const int vw = dst->awidth[pl] - t->dst_x < t->w ? dst->awidth[pl] - t->dst_x : t->w;
    const int vh = dst->aheight[pl] - t->dst_y < t->h ? dst->aheight[pl] - t->dst_y : t->h;
    if (vw <= 0 || vh <= 0) continue;
    uint16_t tmp16[64 * 64];
    uint8_t tmp8[64 * 64];
    for (int r = 0; r < nref; ++r) {
      const vp9o_frame *rf = &refs[t->ref[r]];
      if (dst->hbd)
        vp9o_highbd_inter_predict_block((const uint16_t *)rf->plane[pl], rf->stride[pl], rf->width[pl],
                                        rf->height[pl], t->pos_x[r], t->pos_y[r], t->step_x[r], t->step_y[r],
                                        filt, t->w, t->h, tmp16, 64, r, dst->bit_depth);
      else
        vp9o_inter_predict_block((const uint8_t *)rf->plane[pl], rf->stride[pl], rf->width[pl], rf->height[pl],
                                 t->pos_x[r], t->pos_y[r], t->step_x[r], t->step_y[r], filt, t->w, t->h, tmp8,
                                 64, r);
    }
    for (int y = 0; y < vh; ++y) {
      if (dst->hbd)
        memcpy((uint16_t *)dst->plane[pl] + (size_t)(t->dst_y + y) * dst->stride[pl] + t->dst_x, tmp16 + y * 64,
               vw * 2);
      else
        memcpy((uint8_t *)dst->plane[pl] + (size_t)(t->dst_y + y) * dst->stride[pl] + t->dst_x, tmp8 + y * 64, vw);
    }
  }
}

void vp9o_recon_txb_list(const void *blocks_, int n, const int32_t *coeffs, const vp9o_frame *f) {
  const o_txb *b = (const o_txb *)blocks_;
  for (int i = 0; i < n; ++i)
    txb_add(f, b[i].plane, b[i].x, b[i].y, b[i].tx_size, b[i].tx_type, b[i].eob, coeffs + b[i].coeff_off);
}

void vp9o_recon_intra_list(const void *tasks_, int n, const int32_t *coeffs, const vp9o_frame *f) {
  const o_intra *tasks = (const o_intra *)tasks_;
  for (int i = 0; i < n; ++i) {
    const o_intra *t = &tasks[i];
    const int pl = t->plane, bs = 4 << t->tx_size;
    vp9o_intra_args a = { t->mode, bs, t->flags & 1, (t->flags >> 1) & 1, (t->flags >> 2) & 1, t->x, t->y,
                          f->awidth[pl], f->aheight[pl] };
    const int vw = f->awidth[pl] - t->x < bs ? f->awidth[pl] - t->x : bs;
    const int vh = f->aheight[pl] - t->y < bs ? f->aheight[pl] - t->y : bs;
    if (vw <= 0 || vh <= 0) continue;
    if (f->hbd) {
      uint16_t tmp[32 * 32];
      uint16_t *p = (uint16_t *)f->plane[pl] + (size_t)t->y * f->stride[pl] + t->x;
      vp9o_highbd_predict_intra(&a, p, f->stride[pl], tmp, bs, f->bit_depth);
      for (int r = 0; r < vh; ++r) memcpy(p + (size_t)r * f->stride[pl], tmp + r * bs, vw * 2);
    } else {
      uint8_t tmp[32 * 32];
      uint8_t *p = (uint8_t *)f->plane[pl] + (size_t)t->y * f->stride[pl] + t->x;
      vp9o_predict_intra(&a, p, f->stride[pl], tmp, bs);
      for (int r = 0; r < vh; ++r) memcpy(p + (size_t)r * f->stride[pl], tmp + r * bs, vw);
    }
    if (coeffs && t->eob) txb_add(f, pl, t->x, t->y, t->tx_size, t->tx_type, t->eob, coeffs + t->coeff_off);
  }
}
