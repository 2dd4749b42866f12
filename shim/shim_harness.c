/*
 * shim_harness.c — TEST HARNESS for shim/vp9hip_libvpx_shim.c (not product).
 *
 * Plays the role of the reference's decode_tiles (libvpx/vp9/decoder/vp9_decodeframe.c:2303-2639)
 * around the two wrapper calls: it builds the very structures that function hands over — a
 * VP9Decoder / VP9_COMMON with a BufferPool of libvpx frame buffers (allocated by the reference's
 * own vpx_realloc_frame_buffer, border VP9_DEC_BORDER_IN_PIXELS), the MODE_INFO array + ModeInfoBuf
 * index arrays + size_for_mb, and a frameBuf laid out by initBuf's rules (:2242-2270) — from flat
 * test vectors, calls wrap_cuda_inter_prediction / wrap_cuda_intra_prediction exactly as :2546 /
 * :2564 do, and returns the host frame the wrappers delivered.
 *
 * Compiled in the development container against the reference's headers (it cannot travel as
 * source-plus-headers: /root/reference does not exist on the GPU box); the built .so travels.
 * Block record: the 35-int32 layout of oracle/ref_frame_driver.c.
 */
#include <stdio.h>
#include <setjmp.h>
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "buffers_struct.h"
#include "vp9/common/vp9_loopfilter.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/decoder/vp9_decoder.h"
#include "vpx_scale/yv12config.h"

#include "vp9hip_libvpx_shim.h"

extern int wrap_cuda_inter_prediction(int n, double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf,
                                      VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols,
                                      tran_high_t *residuals);
extern int wrap_cuda_intra_prediction(double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf,
                                      VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols,
                                      frameBuf *frameBuffer);

#define REC 35

typedef struct {
  VP9Decoder *pbi;
  BufferPool *pool;
  unsigned int frame_no;
  int gpu_lf;
} harness;

static void vp9hip_shim_set_gpu_loop_filter_once(harness *H, VP9Decoder *pbi, int on) {
  if (H->gpu_lf != on) vp9hip_shim_set_gpu_loop_filter(pbi, on); /* toggling forgets the resident frames */
  H->gpu_lf = on;
}

void *shimtest_create(void) {
  harness *h = (harness *)calloc(1, sizeof(*h));
  h->pbi = (VP9Decoder *)vpx_memalign(32, sizeof(VP9Decoder));
  memset(h->pbi, 0, sizeof(VP9Decoder));
  h->pool = (BufferPool *)calloc(1, sizeof(BufferPool));
  h->pbi->common.buffer_pool = h->pool;
  return h;
}

void shimtest_destroy(void *hp) {
  harness *h = (harness *)hp;
  if (!h) return;
  vp9hip_shim_release(h->pbi);
  for (int i = 0; i < FRAME_BUFFERS; ++i) vpx_free_frame_buffer(&h->pool->frame_bufs[i].buf);
  free(h->pbi->common.lf.lfm);
  free(h->pool);
  vpx_free(h->pbi);
  free(h);
}

static void copy_plane(void *dst, int dstride, const void *src, int sstride, int w, int h, int bps) {
  for (int y = 0; y < h; ++y) memcpy((char *)dst + (size_t)y * dstride * bps, (const char *)src + (size_t)y * sstride * bps, (size_t)w * bps);
}

static void *plane_ptr(const YV12_BUFFER_CONFIG *b, int p) {
  uint8_t *q = p == 0 ? b->y_buffer : p == 1 ? b->u_buffer : b->v_buffer;
  return (b->flags & YV12_FLAG_HIGHBITDEPTH) ? (void *)CONVERT_TO_SHORTPTR(q) : (void *)q;
}

/*
 * One frame through the wrappers.
 *   blocks / n_blocks        decode order
 *   inter_frame              0: key frame (only the intra wrapper is called, as the caller does)
 *   coefficient_mode         1: vp9hip_shim_attach_frame_buffer + dqcoeff/eob; 0: residual planes
 *   ref_planes[3*k+p]        aligned-size planes of reference k (stride = aligned width), or NULL
 *   dq[p], eob[p]            coefficient arrays and eob planes (aligned size, stride = aligned width)
 *   res[p]                   int64 residual planes (aligned size) for the residual mode
 *   out_planes[p]            receive the delivered frame (aligned size, stride = aligned width)
 *   times[4]                 gpu_copy / gpu_run of the inter and intra wrapper
 *   opts                     NULL, or 8 ints: [0] GPU loop filter on (vp9hip_shim_set_gpu_loop_filter; the
 *                            harness then builds cm->lf.lfm with the reference's own vp9_build_mask, one
 *                            call per block as decode_block does, and the thresholds with
 *                            vp9_loop_filter_init), [1] sharpness, [2] new_fb_idx, [3..5] frame-buffer index
 *                            of LAST / GOLDEN / ALTREF, [6] bit k: fill reference k's host buffer from
 *                            ref_planes (else it keeps what an earlier frame left there), [7] bit k:
 *                            overwrite reference k's HOST buffer with garbage first (a resident device copy
 *                            must then be what the wrappers use)
 * Returns 0 or the vpx error code; errbuf gets cm->error.detail.
 */
int shimtest_frame(void *hp, const int32_t *blocks, int n_blocks, int w, int h, int bd, int hbd, int log2_tile_cols,
                   int lossless, int inter_frame, int coefficient_mode, void *const ref_planes[9], const int ref_w[3],
                   const int ref_h[3], const int32_t *const dq[3], const int32_t *const eob[3], const int64_t *const res[3],
                   void *const out_planes[3], double times[4], char *errbuf, int errbuf_len, const int32_t *opts) {
  harness *H = (harness *)hp;
  VP9Decoder *pbi = H->pbi;
  VP9_COMMON *cm = &pbi->common;
  const int bps = hbd ? 2 : 1;
  const int aw = (w + 7) & ~7, ah = (h + 7) & ~7;
  int rc = 0;
  cm->width = w;
  cm->height = h;
  cm->mi_cols = aw >> 3;
  cm->mi_rows = ah >> 3;
  cm->subsampling_x = cm->subsampling_y = 1;
  cm->bit_depth = (vpx_bit_depth_t)bd;
  cm->use_highbitdepth = hbd;
  cm->log2_tile_cols = log2_tile_cols;
  cm->frame_type = inter_frame ? INTER_FRAME : KEY_FRAME;
  cm->intra_only = 0;
  cm->current_video_frame = ++H->frame_no;
  const int gpu_lf = opts ? opts[0] : 0;
  const int new_idx = opts ? opts[2] : 3;
  cm->new_fb_idx = new_idx;
  cm->error.error_code = VPX_CODEC_OK;
  cm->error.setjmp = 0;
  pbi->mb.lossless = lossless;
  if (vpx_realloc_frame_buffer(&H->pool->frame_bufs[new_idx].buf, w, h, 1, 1, hbd, VP9_DEC_BORDER_IN_PIXELS, 0, NULL, NULL, NULL)) return -100;
  YV12_BUFFER_CONFIG *cur = &H->pool->frame_bufs[new_idx].buf;
  memset(cur->buffer_alloc, 0x55, cur->frame_size);
  for (int k = 0; k < 3; ++k) {
    cm->frame_refs[k].buf = NULL;
    cm->frame_refs[k].idx = -1;
    if (!inter_frame || ref_w[k] <= 0) continue;
    const int ridx = opts ? opts[3 + k] : k;
    YV12_BUFFER_CONFIG *rb = &H->pool->frame_bufs[ridx].buf;
    if (!opts || ((opts[6] >> k) & 1)) {
      if (vpx_realloc_frame_buffer(rb, ref_w[k], ref_h[k], 1, 1, hbd, VP9_DEC_BORDER_IN_PIXELS, 0, NULL, NULL, NULL)) return -101;
      memset(rb->buffer_alloc, 0xaa, rb->frame_size);
      for (int p = 0; p < 3; ++p) {
        const int pw = p ? rb->uv_width : rb->y_width, ph = p ? rb->uv_height : rb->y_height;
        copy_plane(plane_ptr(rb, p), p ? rb->uv_stride : rb->y_stride, ref_planes[3 * k + p], pw, pw, ph, bps);
      }
    }
    if (opts && ((opts[7] >> k) & 1)) memset(rb->buffer_alloc, 0x3c, rb->frame_size);
    cm->frame_refs[k].buf = rb;
    cm->frame_refs[k].idx = ridx;
  }

  /* MODE_INFO + ModeInfoBuf + size_for_mb, as decode_block leaves them (:1226-1233) */
  MODE_INFO *mis = (MODE_INFO *)calloc((size_t)n_blocks + 1, sizeof(MODE_INFO));
  ModeInfoBuf MiBuf;
  MiBuf.mi = (MODE_INFO **)calloc((size_t)n_blocks + 1, sizeof(MODE_INFO *));
  MiBuf.mi_row = (int *)calloc((size_t)n_blocks + 1, sizeof(int));
  MiBuf.mi_col = (int *)calloc((size_t)n_blocks + 1, sizeof(int));
  MiBuf.bwl = (int *)calloc((size_t)n_blocks + 1, sizeof(int));
  MiBuf.bhl = (int *)calloc((size_t)n_blocks + 1, sizeof(int));
  const int sb_cols = (cm->mi_cols + 7) >> 3, sb_rows = (cm->mi_rows + 7) >> 3;
  int *size_for_mb = (int *)calloc((size_t)sb_cols * sb_rows + 1, sizeof(int));
  int levels[MAX_SEGMENTS], n_levels = 0;
  cm->lf.filter_level = gpu_lf ? 32 : 0;
  cm->lf.sharpness_level = opts ? opts[1] : 0;
  cm->skip_loop_filter = 0;
  if (gpu_lf) {
    vp9_loop_filter_init(cm);  /* lfthr[] from the sharpness */
    cm->lf.lfm_stride = sb_cols;
    free(cm->lf.lfm);
    cm->lf.lfm = (LOOP_FILTER_MASK *)calloc((size_t)sb_rows * sb_cols, sizeof(LOOP_FILTER_MASK));
  }
  for (int i = 0; i < n_blocks && !rc; ++i) {
    const int32_t *b = blocks + REC * i;
    MODE_INFO *mi = &mis[i];
    int seg = -1;
    mi->sb_type = (BLOCK_SIZE)b[2];
    mi->tx_size = (TX_SIZE)b[3];
    mi->skip = (int8_t)b[4];
    mi->interp_filter = (INTERP_FILTER)b[5];
    mi->ref_frame[0] = (MV_REFERENCE_FRAME)b[6];
    mi->ref_frame[1] = (MV_REFERENCE_FRAME)(b[7] > 0 ? b[7] : NONE);
    mi->mode = (PREDICTION_MODE)(b[6] > 0 ? NEWMV : b[8]);
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
    for (int k = 0; k < n_levels; ++k)
      if (levels[k] == b[14]) seg = k;
    if (seg < 0) {
      if (n_levels == MAX_SEGMENTS) {
        rc = -102;
        break;
      }
      seg = n_levels;
      levels[n_levels++] = b[14];
      memset(cm->lf_info.lvl[seg], b[14], sizeof(cm->lf_info.lvl[seg]));
    }
    mi->segment_id = (int8_t)seg;
    MiBuf.mi[i] = mi;
    MiBuf.mi_row[i] = b[0];
    MiBuf.mi_col[i] = b[1];
    {
      const BLOCK_SIZE bs = mi->sb_type < BLOCK_8X8 ? BLOCK_8X8 : mi->sb_type;
      MiBuf.bwl[i] = b_width_log2_lookup[bs];
      MiBuf.bhl[i] = b_height_log2_lookup[bs];
    }
    ++size_for_mb[(b[0] >> 3) * sb_cols + (b[1] >> 3)];
  }
  if (gpu_lf && !rc)
    for (int i = 0; i < n_blocks; ++i) { /* decode_block, :1238-1241 */
      const BLOCK_SIZE bs = mis[i].sb_type < BLOCK_8X8 ? BLOCK_8X8 : mis[i].sb_type;
      vp9_build_mask(cm, &mis[i], MiBuf.mi_row[i], MiBuf.mi_col[i], num_8x8_blocks_wide_lookup[bs],
                     num_8x8_blocks_high_lookup[bs]);
    }

  /* frameBuf by initBuf's rules (:2242-2270) */
  frameBuf fb;
  memset(&fb, 0, sizeof(fb));
  tran_low_t *dqp[3] = { (tran_low_t *)dq[0], (tran_low_t *)dq[1], (tran_low_t *)dq[2] };
  if (!rc) {
    const int uv_border_h = cur->border >> cur->subsampling_y, uv_border_w = cur->border >> cur->subsampling_x;
    const uint64_t yplane_size = (cur->y_height + 2 * cur->border) * (uint64_t)cur->y_stride;
    const uint64_t uvplane_size = (cur->uv_height + 2 * uv_border_h) * (uint64_t)cur->uv_stride;
    fb.residuals = (tran_high_t *)calloc(cur->frame_size, sizeof(tran_high_t));
    fb.eob = (int *)malloc(cur->frame_size * sizeof(int));
    memset(fb.eob, 0x7f, cur->frame_size * sizeof(int)); /* initBuf leaves it uninitialised */
    fb.plane_residuals[0] = fb.residuals + (cur->border * cur->y_stride) + cur->border;
    fb.plane_residuals[1] = fb.residuals + yplane_size + (uv_border_h * cur->uv_stride) + uv_border_w;
    fb.plane_residuals[2] = fb.residuals + yplane_size + uvplane_size + (uv_border_h * cur->uv_stride) + uv_border_w;
    fb.plane_eob[0] = fb.eob + (cur->border * cur->y_stride) + cur->border;
    fb.plane_eob[1] = fb.eob + yplane_size + (uv_border_h * cur->uv_stride) + uv_border_w;
    fb.plane_eob[2] = fb.eob + yplane_size + uvplane_size + (uv_border_h * cur->uv_stride) + uv_border_w;
    fb.dqcoeff = dqp;
    for (int p = 0; p < 3; ++p) {
      const int pw = p ? cur->uv_width : cur->y_width, ph = p ? cur->uv_height : cur->y_height;
      const int st = p ? cur->uv_stride : cur->y_stride;
      if (eob && eob[p]) copy_plane(fb.plane_eob[p], st, eob[p], pw, pw, ph, (int)sizeof(int));
      if (res && res[p]) copy_plane(fb.plane_residuals[p], st, res[p], pw, pw, ph, (int)sizeof(tran_high_t));
    }
    vp9hip_shim_attach_frame_buffer(pbi, coefficient_mode ? &fb : NULL);
    if (opts) vp9hip_shim_set_gpu_loop_filter_once(H, pbi, gpu_lf);
    times[0] = times[1] = times[2] = times[3] = 0.0;
    /* the trap libvpx installs around a frame (vp9_receive_compressed_data, libvpx/vp9/decoder/vp9_decoder.c:
     * 458-466): an error inside a wrapper unwinds to here with longjmp, past the wrapper's own frames */
    if (setjmp(cm->error.jmp)) {
      cm->error.setjmp = 0;
    } else {
      cm->error.setjmp = 1;
      if (inter_frame)
        wrap_cuda_inter_prediction(w * h, &times[0], &times[1], size_for_mb, &MiBuf, cm, pbi, 1, 1 << log2_tile_cols, fb.residuals);
      wrap_cuda_intra_prediction(&times[2], &times[3], size_for_mb, &MiBuf, cm, pbi, 1, 1 << log2_tile_cols, &fb);
      cm->error.setjmp = 0;
    }
    rc = (int)cm->error.error_code;
    if (rc && errbuf) snprintf(errbuf, (size_t)errbuf_len, "%s", cm->error.detail);
    for (int p = 0; p < 3 && !rc; ++p) {
      const int pw = p ? cur->uv_width : cur->y_width, ph = p ? cur->uv_height : cur->y_height;
      copy_plane(out_planes[p], pw, plane_ptr(cur, p), p ? cur->uv_stride : cur->y_stride, pw, ph, bps);
    }
    free(fb.residuals);
    free(fb.eob);
  }
  free(size_for_mb);
  free(MiBuf.mi);
  free(MiBuf.mi_row);
  free(MiBuf.mi_col);
  free(MiBuf.bwl);
  free(MiBuf.bhl);
  free(mis);
  return rc;
}
