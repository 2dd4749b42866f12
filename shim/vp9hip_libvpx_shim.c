/*
 * vp9hip_libvpx_shim.c — the reference's frame-level call surface on top of libvp9hip.so.
 *
 * Exports exactly the two symbols the reference's decoder links against
 *   int wrap_cuda_inter_prediction(int n, double *gpu_copy, double *gpu_run, int *size_for_mb,
 *         ModeInfoBuf *MiBuf, VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols,
 *         tran_high_t *residuals);
 *   int wrap_cuda_intra_prediction(double *gpu_copy, double *gpu_run, int *size_for_mb,
 *         ModeInfoBuf *MiBuf, VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols,
 *         frameBuf *frameBuffer);
 * (/root/reference/vpx-master/cuda_extern_wrap.cpp:5-17; declared by the caller at
 * libvpx/vp9/decoder/vp9_decodeframe.c:2299-2302, called at :2546 and :2564), with the
 * reference's types (vpx-master/buffers_struct.h:9-57), so it replaces cuda_extern_wrap.cpp +
 * inter_cuda_kernel.cu + intra_cuda_kernel.cu at link time and vpxdec needs no change.
 *
 * This file is C, compiled against the libvpx tree it is linked into (it needs the layouts of
 * MODE_INFO / VP9_COMMON / VP9Decoder / YV12_BUFFER_CONFIG and that tree's vpx_config.h).  All it
 * does is copy fields into the plain records of include/vp9hip_pack.h and call
 * include/vp9hip_decoder.h; packing, transfers and kernels live in libvp9hip.so.
 *
 * Two residual modes:
 *   - as called by the unchanged reference: the CPU transforms of its phase B already ran
 *     (vp9_decodeframe.c:2443-2534) and `residuals` / frameBuffer->plane_residuals hold int64
 *     residual planes -> they are added on the GPU (what the reference's kernels do,
 *     vpx-master/inter_cuda_kernel.cu:821-829).  High-bitdepth frames only, like the reference.
 *   - after vp9hip_shim_attach_frame_buffer(pbi, frameBuffer): the inverse transforms run on the
 *     GPU from frameBuffer->dqcoeff / plane_eob and phase B can be deleted; `residuals` is ignored.
 *
 * Errors: a HIP / argument failure is reported through vpx_internal_error(&cm->error, ...), which
 * longjmps to the decoder's trap when one is set (libvpx/vp9/decoder/vp9_decoder.c:458-466) —
 * there is no CPU fallback.  Out-parameters *gpu_copy / *gpu_run are seconds, as in the
 * reference (vpx-master/inter_cuda_kernel.cu:1069-1101).
 */
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "./vpx_config.h"
#include "buffers_struct.h"
#include "vp9/common/vp9_loopfilter.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/decoder/vp9_decoder.h"
#include "vpx_ports/mem.h"

#include "vp9hip_decoder.h"
#include "vp9hip_libvpx_shim.h"

#define SHIM_MAX_DECODERS 16
/* pool slots: 0..2 = LAST/GOLDEN/ALTREF of the frame being decoded, 3 = the frame itself */
#define SLOT_CUR 3

typedef struct {
  VP9Decoder *pbi;
  vp9hip_decoder *dec;
  const frameBuf *attached;     /* coefficient mode when non-NULL */
  int eob_shift;                /* 0: the reference's frame-strided eob plane; 2: one int per 4x4 position (E11) */
  const tran_low_t *dq_start[3]; /* start of the per-plane coefficient arrays (attach time) */
  vp9hip_block *blocks;
  int blocks_cap;
  int frame_open;               /* the inter wrapper began this frame and left it on the device */
  unsigned int open_frame_no;
  MODE_INFO **open_mi;
  /* GPU loop filter + resident references: pool slot i shadows cm->buffer_pool->
   * frame_bufs[i]; a slot is valid while it holds exactly what the host buffer holds */
  /* page-locked memory lent to the caller's initBuf (vp9hip_shim_frame_memory): 0..2 coefficient arrays, 3 eob plane */
  void *frame_mem[4];
  size_t frame_mem_cap[4];
  size_t frame_mem_req[4]; /* bytes the caller asked for last */
  /* tile-parallel entropy stage: per-block coefficient offsets + the filled regions of this frame */
  uint32_t *block_off;
  int block_off_cap, tile_layout_blocks, tile_layout_compact;
  vp9hip_coeff_region regions[64 * 3];
  int n_regions;
  int creating;
  int gpu_lf;
  struct {
    const uint8_t *alloc;
    int w, h, bd, hbd, valid;
  } resident[VP9HIP_POOL_SLOTS];
} shim_state;

/* One record per decoder instance (VP9Decoder*), found under a lock: decoder threads of one process — one per
 * GPU, each with its own VP9Decoder — create, look up and release their records concurrently.  Everything else
 * in a record is only touched by the thread that owns that decoder (libvpx's contract for a codec instance). */
static shim_state g_state[SHIM_MAX_DECODERS];
static pthread_mutex_t g_state_lock = PTHREAD_MUTEX_INITIALIZER;

static double now_s(void);

/* VP9HIP_SHIM_TRACE=1: where the time of the two entry points goes, summed over the stream and printed
 * at exit (seconds of host wall time; "kernels" are waits for the GPU).  A measurement aid for ONE decoder per
 * process: the counters are process-wide and not synchronised — with several decoder threads the sums mix. */
static struct {
  int on, frames;
  double gather, pack_upload, refs, inter_wait, masks, intra_wait, download;
  double gpu_inter_ms, gpu_intra_ms;
  double coeff_mb, blocks;
} g_trace;

static double g_mark_t[5], g_mark_sum[5], g_mark_outside;

void vp9hip_shim_mark(struct VP9Decoder *pbi, int mark) {
  (void)pbi;
  if (!g_trace.on || mark < 0 || mark > 4) return;
  const double t = now_s();
  if (mark == 0) {
    if (g_mark_t[4] > 0.0) g_mark_outside += t - g_mark_t[4];
  } else {
    g_mark_sum[mark] += t - g_mark_t[mark - 1];
  }
  g_mark_t[mark] = t;
}

static void trace_report(void) {
  if (!g_trace.frames) return;
  if (g_mark_sum[4] > 0.0)
    fprintf(stderr,
            "vp9hip shim: decode_tiles per frame: set-up %.3f ms, entropy decode %.3f ms, reconstruction entry points %.3f ms, "
            "tear-down %.3f ms; outside decode_tiles (headers, probability adaptation, vpxdec) %.3f ms\n",
            g_mark_sum[1] * 1e3 / g_trace.frames, g_mark_sum[2] * 1e3 / g_trace.frames, g_mark_sum[3] * 1e3 / g_trace.frames,
            g_mark_sum[4] * 1e3 / g_trace.frames, g_mark_outside * 1e3 / g_trace.frames);
  const double n = g_trace.frames, ms = 1e3 / n;
  fprintf(stderr,
          "vp9hip shim: %d frames; per frame: gather blocks %.3f ms, pack + list/coefficient upload %.3f ms, reference "
          "upload + slot %.3f ms, wait inter kernels %.3f ms (GPU %.3f), masks %.3f ms, wait intra+filter kernels %.3f ms "
          "(GPU %.3f), frame download %.3f ms; total in the entry points %.3f ms\n",
          g_trace.frames, g_trace.gather * ms, g_trace.pack_upload * ms, g_trace.refs * ms, g_trace.inter_wait * ms,
          g_trace.gpu_inter_ms / n, g_trace.masks * ms, g_trace.intra_wait * ms, g_trace.gpu_intra_ms / n, g_trace.download * ms,
          (g_trace.gather + g_trace.pack_upload + g_trace.refs + g_trace.inter_wait + g_trace.masks + g_trace.intra_wait +
           g_trace.download) * ms);
  fprintf(stderr, "vp9hip shim: per frame: %.0f blocks, %.2f MB of coefficient slots uploaded\n", g_trace.blocks / n, g_trace.coeff_mb / n);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static shim_state *state_of(VP9Decoder *pbi, VP9_COMMON *cm) {
  int free_i = -1;
  shim_state *s = NULL;
  pthread_mutex_lock(&g_state_lock);
  for (int i = 0; i < SHIM_MAX_DECODERS && !s; ++i) {
    if (g_state[i].pbi == pbi) s = &g_state[i];
    if (!g_state[i].pbi && free_i < 0) free_i = i;
  }
  if (!s && free_i >= 0) {
    s = &g_state[free_i];
    memset(s, 0, sizeof(*s));
    s->pbi = pbi; /* reserves the slot; the GPU objects follow outside the lock */
    s->creating = 1;
  }
  pthread_mutex_unlock(&g_state_lock);
  if (!s) {
    vpx_internal_error(&cm->error, VPX_CODEC_MEM_ERROR, "vp9hip shim: too many decoder instances");
    return NULL; /* reached only when no setjmp trap is installed */
  }
  if (s->creating) {
    const char *dev = getenv("VP9HIP_DEVICE");
    if (getenv("VP9HIP_SHIM_TRACE") && !g_trace.on) {
      g_trace.on = 1;
      atexit(trace_report);
    }
    int rc = vp9hip_decoder_create(dev ? atoi(dev) : 0, &s->dec);
    if (rc != VP9HIP_OK) {
      pthread_mutex_lock(&g_state_lock);
      memset(s, 0, sizeof(*s));
      pthread_mutex_unlock(&g_state_lock);
      vpx_internal_error(&cm->error, VPX_CODEC_ERROR, "vp9hip shim: %s", vp9hip_last_error(NULL));
      return NULL;
    }
    s->creating = 0;
  }
  return s;
}

void vp9hip_shim_attach_frame_buffer(struct VP9Decoder *pbi, const struct frame_buffer *frameBuffer) {
  shim_state *s = state_of(pbi, &pbi->common);
  if (!s) return;
  s->attached = frameBuffer;
  s->tile_layout_blocks = -1; /* a new frame: any tile layout belongs to the previous one */
  if (frameBuffer)
    for (int p = 0; p < 3; ++p) s->dq_start[p] = frameBuffer->dqcoeff[p];
}

void vp9hip_shim_set_eob_layout(struct VP9Decoder *pbi, int log2_granularity) {
  shim_state *s = state_of(pbi, &pbi->common);
  if (s) s->eob_shift = log2_granularity == 2 ? 2 : 0;
}

void *vp9hip_shim_frame_memory(struct VP9Common *cm, int which, size_t bytes) {
  /* the decoder's VP9_COMMON is a member of its VP9Decoder (libvpx/vp9/decoder/vp9_decoder.h) */
  VP9Decoder *pbi = (VP9Decoder *)((char *)cm - offsetof(VP9Decoder, common));
  shim_state *s = state_of(pbi, cm);
  if (!s || which < 0 || which > 3) return NULL;
  s->frame_mem_req[which] = bytes;
  if (bytes > s->frame_mem_cap[which]) {
    /* nothing of an earlier frame is in flight here: the intra wrapper synchronised before it returned */
    vp9hip_decoder_host_free(s->dec, s->frame_mem[which]);
    s->frame_mem_cap[which] = 0;
    s->frame_mem[which] = vp9hip_decoder_host_alloc(s->dec, bytes + bytes / 8);
    if (!s->frame_mem[which]) {
      vpx_internal_error(&cm->error, VPX_CODEC_MEM_ERROR, "vp9hip shim: out of page-locked memory");
      return NULL;
    }
    s->frame_mem_cap[which] = bytes + bytes / 8;
  }
  return s->frame_mem[which];
}

/* ---- tile-parallel entropy stage --------------------------------------------------------------------- */
#define SHIM_POOL_MAX 64
static struct {
  pthread_mutex_t mu;
  pthread_cond_t go, done;
  pthread_t th[SHIM_POOL_MAX];
  int n_threads, pending, next, n_items;
  unsigned gen;
  void (*fn)(void *, int);
  void *arg;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER };
/* One run at a time: the pool holds ONE job record (fn, arg, next, pending), and decoder threads of one process —
 * one per GPU — enter here concurrently; a second run started while the first is in flight would overwrite it.
 * Held for the whole call; g_pool.mu is released while an item runs. */
static pthread_mutex_t g_pool_run = PTHREAD_MUTEX_INITIALIZER;

static void *pool_worker(void *unused) {
  unsigned seen = 0;
  (void)unused;
  pthread_mutex_lock(&g_pool.mu);
  for (;;) {
    while (g_pool.gen == seen) pthread_cond_wait(&g_pool.go, &g_pool.mu);
    seen = g_pool.gen;
    while (g_pool.next < g_pool.n_items) {
      const int i = g_pool.next++;
      void (*fn)(void *, int) = g_pool.fn;
      void *arg = g_pool.arg;
      pthread_mutex_unlock(&g_pool.mu);
      fn(arg, i);
      pthread_mutex_lock(&g_pool.mu);
      if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
    }
  }
  return NULL;
}

void vp9hip_shim_run_parallel(struct VP9Decoder *pbi, int n, void (*fn)(void *arg, int index), void *arg) {
  (void)pbi;
  if (n <= 0) return;
  int want = n - 1;
  const char *e = getenv("VP9HIP_SHIM_THREADS");
  if (e && atoi(e) - 1 < want) want = atoi(e) - 1;
  if (want > SHIM_POOL_MAX) want = SHIM_POOL_MAX;
  pthread_mutex_lock(&g_pool_run);
  pthread_mutex_lock(&g_pool.mu);
  while (g_pool.n_threads < want) {
    if (pthread_create(&g_pool.th[g_pool.n_threads], NULL, pool_worker, NULL)) break;
    ++g_pool.n_threads;
  }
  if (g_pool.n_threads == 0 || want <= 0) {
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool_run);
    for (int i = 0; i < n; ++i) fn(arg, i);
    return;
  }
  g_pool.fn = fn;
  g_pool.arg = arg;
  g_pool.n_items = n;
  g_pool.next = 0;
  g_pool.pending = n;
  ++g_pool.gen;
  pthread_cond_broadcast(&g_pool.go);
  /* the caller works too */
  while (g_pool.next < g_pool.n_items) {
    const int i = g_pool.next++;
    pthread_mutex_unlock(&g_pool.mu);
    fn(arg, i);
    pthread_mutex_lock(&g_pool.mu);
    --g_pool.pending;
  }
  while (g_pool.pending) pthread_cond_wait(&g_pool.done, &g_pool.mu);
  pthread_mutex_unlock(&g_pool.mu);
  pthread_mutex_unlock(&g_pool_run);
}

uint32_t *vp9hip_shim_block_off_buffer(struct VP9Decoder *pbi, int n_blocks) {
  shim_state *s = state_of(pbi, &pbi->common);
  if (!s) return NULL;
  if (n_blocks > s->block_off_cap) {
    free(s->block_off);
    s->block_off_cap = n_blocks + n_blocks / 4 + 256;
    s->block_off = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)s->block_off_cap);
    if (!s->block_off) {
      s->block_off_cap = 0;
      vpx_internal_error(&pbi->common.error, VPX_CODEC_MEM_ERROR, "vp9hip shim: out of memory");
      return NULL;
    }
  }
  return s->block_off;
}

void vp9hip_shim_set_tile_layout(struct VP9Decoder *pbi, int n_blocks, int n_regions, const int64_t *start, const int64_t *count,
                                 int flags) {
  shim_state *s = state_of(pbi, &pbi->common);
  if (!s) return;
  s->tile_layout_blocks = -1;
  if (n_regions < 0 || n_regions > 64 || n_blocks < 0 || n_blocks > s->block_off_cap) return;
  s->n_regions = 0;
  for (int t = 0; t < n_regions; ++t)
    for (int p = 0; p < 3; ++p) {
      vp9hip_coeff_region *g = &s->regions[s->n_regions++];
      g->plane = p;
      g->reserved = 0;
      g->start = start[3 * t + p];
      g->count = count[3 * t + p];
    }
  s->tile_layout_compact = (flags & VP9HIP_SHIM_COEFF_COMPACT) != 0;
  s->tile_layout_blocks = n_blocks;
}

void vp9hip_shim_set_gpu_loop_filter(struct VP9Decoder *pbi, int enable) {
  shim_state *s = state_of(pbi, &pbi->common);
  if (!s || s->gpu_lf == (enable != 0)) return; /* callers may repeat the call for every frame */
  s->gpu_lf = enable != 0;
  for (int i = 0; i < VP9HIP_POOL_SLOTS; ++i) s->resident[i].valid = 0;
}

void vp9hip_shim_release(struct VP9Decoder *pbi) {
  shim_state old;
  int found = 0;
  pthread_mutex_lock(&g_state_lock);
  for (int i = 0; i < SHIM_MAX_DECODERS; ++i)
    if (g_state[i].pbi == pbi) {
      old = g_state[i];
      memset(&g_state[i], 0, sizeof(g_state[i]));
      found = 1;
      break;
    }
  pthread_mutex_unlock(&g_state_lock);
  if (!found) return;
  for (int k = 0; k < 4; ++k) vp9hip_decoder_host_free(old.dec, old.frame_mem[k]);
  vp9hip_decoder_destroy(old.dec);
  free(old.blocks);
  free(old.block_off);
}

#define SHIM_CHECK(s, cm, expr)                                                                        \
  do {                                                                                                 \
    if ((expr) != VP9HIP_OK) {                                                                         \
      (s)->frame_open = 0;                                                                             \
      vpx_internal_error(&(cm)->error, VPX_CODEC_ERROR, "vp9hip shim: %s", vp9hip_decoder_error((s)->dec)); \
      return -1; /* reached only when no setjmp trap is installed */                                   \
    }                                                                                                  \
  } while (0)

static void host_frame(const YV12_BUFFER_CONFIG *b, int bit_depth, vp9hip_host_frame *h) {
  const int hbd = (b->flags & YV12_FLAG_HIGHBITDEPTH) != 0;
  memset(h, 0, sizeof(*h));
  h->plane[0] = hbd ? (void *)CONVERT_TO_SHORTPTR(b->y_buffer) : (void *)b->y_buffer;
  h->plane[1] = hbd ? (void *)CONVERT_TO_SHORTPTR(b->u_buffer) : (void *)b->u_buffer;
  h->plane[2] = hbd ? (void *)CONVERT_TO_SHORTPTR(b->v_buffer) : (void *)b->v_buffer;
  h->stride[0] = b->y_stride;
  h->stride[1] = h->stride[2] = b->uv_stride;
  h->width = b->y_crop_width;
  h->height = b->y_crop_height;
  h->ss_x = b->subsampling_x;
  h->ss_y = b->subsampling_y;
  h->bit_depth = bit_depth;
  h->hbd = hbd;
}

static int slot_of_cur(const shim_state *s, const VP9_COMMON *cm) { return s->gpu_lf ? cm->new_fb_idx : SLOT_CUR; }

static int is_resident(const shim_state *s, int slot, const YV12_BUFFER_CONFIG *b, int bd) {
  return s->gpu_lf && slot >= 0 && slot < VP9HIP_POOL_SLOTS && s->resident[slot].valid &&
         s->resident[slot].alloc == b->buffer_alloc && s->resident[slot].w == b->y_crop_width &&
         s->resident[slot].h == b->y_crop_height && s->resident[slot].bd == bd &&
         s->resident[slot].hbd == ((b->flags & YV12_FLAG_HIGHBITDEPTH) != 0);
}

static void mark_resident(shim_state *s, int slot, const YV12_BUFFER_CONFIG *b, int bd) {
  s->resident[slot].alloc = b->buffer_alloc;
  s->resident[slot].w = b->y_crop_width;
  s->resident[slot].h = b->y_crop_height;
  s->resident[slot].bd = bd;
  s->resident[slot].hbd = (b->flags & YV12_FLAG_HIGHBITDEPTH) != 0;
  s->resident[slot].valid = 1;
}

/* mode_lf_lut + get_filter_level (libvpx/vp9/common/vp9_loopfilter.c:197-225) */
static int filter_level_of(const VP9_COMMON *cm, const MODE_INFO *mi) {
  static const int mode_lf_lut[MB_MODE_COUNT] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 1 };
  return cm->lf_info.lvl[mi->segment_id][mi->ref_frame[0]][mode_lf_lut[mi->mode]];
}

static int gather_blocks(shim_state *s, VP9_COMMON *cm, const int *size_for_mb, const ModeInfoBuf *MiBuf) {
  const int n_sb = ((cm->mi_rows + 7) >> 3) * ((cm->mi_cols + 7) >> 3);
  int n = 0;
  for (int i = 0; i < n_sb; ++i) n += size_for_mb[i];
  if (n > s->blocks_cap) {
    free(s->blocks);
    s->blocks_cap = n + n / 4 + 64;
    s->blocks = (vp9hip_block *)malloc(sizeof(vp9hip_block) * (size_t)s->blocks_cap);
    if (!s->blocks) {
      s->blocks_cap = 0;
      vpx_internal_error(&cm->error, VPX_CODEC_MEM_ERROR, "vp9hip shim: out of memory");
      return -1;
    }
  }
  for (int i = 0; i < n; ++i) {
    const MODE_INFO *mi = MiBuf->mi[i];
    vp9hip_block *b = &s->blocks[i];
    memset(b, 0, sizeof(*b));
    b->mi_row = (int16_t)MiBuf->mi_row[i];
    b->mi_col = (int16_t)MiBuf->mi_col[i];
    b->sb_type = (uint8_t)mi->sb_type;
    b->tx_size = (uint8_t)mi->tx_size;
    b->skip = (uint8_t)(mi->skip != 0);
    b->ref_frame[0] = mi->ref_frame[0];
    b->ref_frame[1] = mi->ref_frame[1];
    b->filter_level = (uint8_t)filter_level_of(cm, mi);
    if (is_inter_block(mi)) {
      b->interp_filter = (uint8_t)mi->interp_filter;
      for (int r = 0; r < 2; ++r) {
        b->mv[r][0] = mi->mv[r].as_mv.row;
        b->mv[r][1] = mi->mv[r].as_mv.col;
        for (int k = 0; k < 4; ++k) {
          b->sub_mv[k][r][0] = mi->bmi[k].as_mv[r].as_mv.row;
          b->sub_mv[k][r][1] = mi->bmi[k].as_mv[r].as_mv.col;
        }
      }
    } else {
      b->mode = (uint8_t)mi->mode;
      b->uv_mode = (uint8_t)mi->uv_mode;
      for (int k = 0; k < 4; ++k) b->sub_mode[k] = (uint8_t)mi->bmi[k].as_mode;
    }
  }
  return n;
}

static void frame_params(const VP9_COMMON *cm, const VP9Decoder *pbi, const YV12_BUFFER_CONFIG *cur, int coefficient_mode,
                         int gpu_lf, vp9hip_frame_params *P) {
  memset(P, 0, sizeof(*P));
  P->width = cm->width;
  P->height = cm->height;
  P->ss_x = cm->subsampling_x;
  P->ss_y = cm->subsampling_y;
  P->bit_depth = (int)cm->bit_depth;
  P->hbd = (cur->flags & YV12_FLAG_HIGHBITDEPTH) != 0;
  P->lossless = pbi->mb.lossless;
  P->log2_tile_cols = cm->log2_tile_cols;
  P->assume_coded = !coefficient_mode;
  P->build_lf_masks = gpu_lf && cm->lf.filter_level && !cm->skip_loop_filter;
  if (cm->frame_type != KEY_FRAME && !cm->intra_only)
    for (int k = 0; k < 3; ++k) {
      const YV12_BUFFER_CONFIG *rb = cm->frame_refs[k].buf;
      if (rb && rb->y_crop_width > 0) {
        P->ref_width[k] = rb->y_crop_width;
        P->ref_height[k] = rb->y_crop_height;
      }
    }
}

/* initBuf's plane pointers (libvpx/vp9/decoder/vp9_decodeframe.c:2242-2262): the residual planes
 * mirror the frame buffer's layout sample for sample. */
static void residual_planes(const YV12_BUFFER_CONFIG *cur, int byte_alignment, const tran_high_t *residuals,
                            const int64_t *out[3], int32_t stride[3]) {
  const int uv_border_h = cur->border >> cur->subsampling_y, uv_border_w = cur->border >> cur->subsampling_x;
  const int align = byte_alignment == 0 ? 1 : byte_alignment;
  const uint64_t yplane_size = (cur->y_height + 2 * cur->border) * (uint64_t)cur->y_stride + byte_alignment;
  const uint64_t uvplane_size = (cur->uv_height + 2 * uv_border_h) * (uint64_t)cur->uv_stride + byte_alignment;
  out[0] = (const int64_t *)yv12_align_addr(residuals + (cur->border * cur->y_stride) + cur->border, align);
  out[1] = (const int64_t *)yv12_align_addr(residuals + yplane_size + (uv_border_h * cur->uv_stride) + uv_border_w, align);
  out[2] = (const int64_t *)yv12_align_addr(
      residuals + yplane_size + uvplane_size + (uv_border_h * cur->uv_stride) + uv_border_w, align);
  stride[0] = cur->y_stride;
  stride[1] = stride[2] = cur->uv_stride;
}

/* Pack + upload lists/coefficients (+ residual planes) for the current frame. */
static int begin_frame(shim_state *s, VP9_COMMON *cm, VP9Decoder *pbi, int *size_for_mb, ModeInfoBuf *MiBuf,
                       const tran_high_t *residuals, const frameBuf *fb_for_residuals) {
  const YV12_BUFFER_CONFIG *cur = &cm->buffer_pool->frame_bufs[cm->new_fb_idx].buf;
  vp9hip_frame_params P;
  const double tg0 = now_s();
  const int n = gather_blocks(s, cm, size_for_mb, MiBuf);
  if (n < 0) return -1;
  g_trace.gather += now_s() - tg0;
  frame_params(cm, pbi, cur, s->attached != NULL, s->gpu_lf, &P);
  if (s->attached) {
    vp9hip_coeff_layout L;
    const int32_t *dq[3];
    for (int p = 0; p < 3; ++p) {
      L.eob[p] = s->attached->plane_eob[p];
      L.eob_stride[p] = (p ? cur->uv_stride : cur->y_stride) >> s->eob_shift;
      dq[p] = (const int32_t *)s->dq_start[p];
    }
    if (sizeof(tran_low_t) != sizeof(int32_t)) {
      vpx_internal_error(&cm->error, VPX_CODEC_ERROR, "vp9hip shim: needs a CONFIG_VP9_HIGHBITDEPTH build (32-bit tran_low_t)");
      return -1;
    }
    /* coefficient arrays the caller got from vp9hip_shim_frame_memory are page-locked and stay untouched
     * until the frame has been delivered: they travel asynchronously */
    memset(&L.eob_shift, 0, sizeof(L) - offsetof(vp9hip_coeff_layout, eob_shift));
    L.eob_shift = s->eob_shift;
    if (s->tile_layout_blocks >= 0) {
      /* the entropy stage ran one thread per tile column: slots are consecutive per tile (E10) */
      if (s->tile_layout_blocks != n) {
        const int described = s->tile_layout_blocks;
        s->tile_layout_blocks = -1;
        vpx_internal_error(&cm->error, VPX_CODEC_ERROR, "vp9hip shim: tile layout describes %d blocks, the list has %d",
                           described, n);
        return -1;
      }
      L.block_off = s->block_off;
      int64_t base = 0;
      for (int p = 0; p < 3; ++p) {
        L.plane_base[p] = base;
        base += (int64_t)(s->frame_mem_req[p] / sizeof(int32_t));
      }
      L.total = base;
      L.regions = s->regions;
      L.n_regions = s->n_regions;
      L.compact = s->tile_layout_compact;
      s->tile_layout_blocks = -1; /* consumed */
    }
    const int persistent = s->frame_mem[0] && dq[0] == (const int32_t *)s->frame_mem[0] && dq[1] == (const int32_t *)s->frame_mem[1] &&
                           dq[2] == (const int32_t *)s->frame_mem[2];
    SHIM_CHECK(s, cm, vp9hip_decoder_begin_frame_ex(s->dec, &P, s->blocks, n, &L, dq, persistent ? VP9HIP_BEGIN_HOST_PERSISTENT : 0));
    if (g_trace.on) {
      const vp9hip_packed *pk = vp9hip_decoder_packed(s->dec);
      g_trace.blocks += n;
      if (L.block_off)
        for (int64_t r = 0; r < L.n_regions; ++r) g_trace.coeff_mb += 4e-6 * (double)L.regions[r].count;
      else
        g_trace.coeff_mb += 4e-6 * (double)(pk->coeff_count[0] + pk->coeff_count[1] + pk->coeff_count[2]);
    }
  } else {
    const int64_t *res[3];
    int32_t rs[3];
    SHIM_CHECK(s, cm, vp9hip_decoder_begin_frame(s->dec, &P, s->blocks, n, NULL, NULL));
    if (fb_for_residuals) {
      for (int p = 0; p < 3; ++p) res[p] = (const int64_t *)fb_for_residuals->plane_residuals[p];
      rs[0] = cur->y_stride;
      rs[1] = rs[2] = cur->uv_stride;
    } else {
      residual_planes(cur, cm->byte_alignment, residuals, res, rs);
    }
    SHIM_CHECK(s, cm, vp9hip_decoder_set_residual_planes(s->dec, res, rs));
  }
  return 0;
}

int wrap_cuda_inter_prediction(int n, double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf,
                               VP9_COMMON *cm, VP9Decoder *pbi, int tile_rows, int tile_cols, tran_high_t *residuals) {
  shim_state *s = state_of(pbi, cm);
  const YV12_BUFFER_CONFIG *cur = &cm->buffer_pool->frame_bufs[cm->new_fb_idx].buf;
  int ref_slot[3] = { -1, -1, -1 };
  float ms = 0.f;
  (void)n;
  (void)tile_rows;
  (void)tile_cols;
  if (!s) return -1;
  const double t0 = now_s();
  s->frame_open = 0;
  if (begin_frame(s, cm, pbi, size_for_mb, MiBuf, residuals, NULL)) return -1;
  const double t_begin = now_s();
  /* the reference re-sends its three references for every frame (inter_cuda_kernel.cu:1073-1079); only
   * the ones the frame's blocks use travel here, and with the GPU loop filter enabled a reference that
   * was decoded (or uploaded) earlier is still in its pool slot */
  {
    const vp9hip_packed *pk = vp9hip_decoder_packed(s->dec);
    for (int k = 0; k < 3; ++k) {
      vp9hip_host_frame h;
      if (!((pk->refs_used >> k) & 1)) continue;
      const YV12_BUFFER_CONFIG *rb = cm->frame_refs[k].buf;
      const int slot = s->gpu_lf ? cm->frame_refs[k].idx : k;
      if (slot < 0 || slot >= VP9HIP_POOL_SLOTS || (s->gpu_lf && slot == cm->new_fb_idx)) {
        vpx_internal_error(&cm->error, VPX_CODEC_ERROR, "vp9hip shim: reference %d has no usable frame buffer index", k);
        return -1;
      }
      if (!is_resident(s, slot, rb, (int)cm->bit_depth)) {
        host_frame(rb, (int)cm->bit_depth, &h);
        SHIM_CHECK(s, cm, vp9hip_decoder_upload(s->dec, slot, &h));
        if (s->gpu_lf) mark_resident(s, slot, rb, (int)cm->bit_depth);
      }
      ref_slot[k] = slot;
    }
  }
  if (s->gpu_lf) s->resident[slot_of_cur(s, cm)].valid = 0;
  SHIM_CHECK(s, cm, vp9hip_decoder_alloc_slot(s->dec, slot_of_cur(s, cm), cm->width, cm->height, cm->subsampling_x,
                                              (int)cm->bit_depth, (cur->flags & YV12_FLAG_HIGHBITDEPTH) != 0, 1));
  const double t1 = now_s();
  SHIM_CHECK(s, cm, vp9hip_decoder_run(s->dec, VP9HIP_PHASE_INTER, ref_slot, slot_of_cur(s, cm), NULL, NULL));
  /* the kernels only read device memory and page-locked arrays nobody touches before the intra wrapper has
   * delivered the frame: no need to wait here (the reference's out-parameter *gpu_run then reads 0 for
   * this entry point; VP9HIP_SHIM_TRACE measures it) */
  if (g_trace.on) {
    SHIM_CHECK(s, cm, vp9hip_decoder_sync(s->dec));
    SHIM_CHECK(s, cm, vp9hip_decoder_last_run_ms(s->dec, &ms));
  }
  g_trace.pack_upload += t_begin - t0;
  g_trace.refs += t1 - t_begin;
  g_trace.inter_wait += now_s() - t1;
  g_trace.gpu_inter_ms += ms;
  /* the frame stays on the device: wrap_cuda_intra_prediction runs next on the same frame
   * (vp9_decodeframe.c:2546-2564) and delivers it to the host */
  s->frame_open = 1;
  s->open_frame_no = cm->current_video_frame;
  s->open_mi = MiBuf->mi;
  if (gpu_copy) *gpu_copy = t1 - t0;
  if (gpu_run) *gpu_run = (double)ms * 1e-3;
  return 0;
}

int wrap_cuda_intra_prediction(double *gpu_copy, double *gpu_run, int *size_for_mb, ModeInfoBuf *MiBuf, VP9_COMMON *cm,
                               VP9Decoder *pbi, int tile_rows, int tile_cols, frameBuf *frameBuffer) {
  shim_state *s = state_of(pbi, cm);
  const YV12_BUFFER_CONFIG *cur = &cm->buffer_pool->frame_bufs[cm->new_fb_idx].buf;
  const int ref_slot[3] = { -1, -1, -1 };
  vp9hip_host_frame h;
  float ms = 0.f;
  (void)tile_rows;
  (void)tile_cols;
  if (!s) return -1;
  double t_copy = 0.0;
  double t0 = now_s();
  if (!(s->frame_open && s->open_frame_no == cm->current_video_frame && s->open_mi == MiBuf->mi)) {
    /* key / intra-only frame path of the caller: the inter wrapper was not called */
    if (begin_frame(s, cm, pbi, size_for_mb, MiBuf, NULL, frameBuffer)) return -1;
    if (s->gpu_lf) s->resident[slot_of_cur(s, cm)].valid = 0;
    SHIM_CHECK(s, cm, vp9hip_decoder_alloc_slot(s->dec, slot_of_cur(s, cm), cm->width, cm->height, cm->subsampling_x,
                                                (int)cm->bit_depth, (cur->flags & YV12_FLAG_HIGHBITDEPTH) != 0, 1));
  }
  s->frame_open = 0;
  /* phase E on the GPU (vp9hip_shim_set_gpu_loop_filter): the masks were built by the packer from the
   * blocks (vp9_build_mask + vp9_adjust_mask semantics, with the skip flag libvpx's loop filter sees: an
   * inter block of 8x8 or more without a coded coefficient counts as skipped, vp9_decodeframe.c:1195 — the
   * masks cm->lf.lfm holds were accumulated at parse time, before that update); libvpx's threshold
   * table; the island walk and the filter then run side by side */
  int phases = VP9HIP_PHASE_INTRA;
  vp9hip_lf_thresh th;
  const int filter = s->gpu_lf && cm->lf.filter_level && !cm->skip_loop_filter;
  if (filter) {
    for (int l = 0; l < 64; ++l) {
      th.mblim[l] = cm->lf_info.lfthr[l].mblim[0];
      th.lim[l] = cm->lf_info.lfthr[l].lim[0];
      th.hev_thr[l] = cm->lf_info.lfthr[l].hev_thr[0];
    }
    phases |= VP9HIP_PHASE_LF;
  }
  t_copy += now_s() - t0;
  g_trace.masks += now_s() - t0;
  t0 = now_s();
  SHIM_CHECK(s, cm, vp9hip_decoder_run(s->dec, phases, ref_slot, slot_of_cur(s, cm), NULL, filter ? &th : NULL));
  SHIM_CHECK(s, cm, vp9hip_decoder_sync(s->dec));
  SHIM_CHECK(s, cm, vp9hip_decoder_last_run_ms(s->dec, &ms));
  g_trace.intra_wait += now_s() - t0;
  g_trace.gpu_intra_ms += ms;
  /* the reference's contract: the reconstructed frame is in the host buffer on return, because the
   * CPU loop filter runs next (vp9_decodeframe.c:2585; intra_cuda_kernel.cu:1368).  With the GPU loop
   * filter the delivered frame is already filtered (the caller drops phase E) and the device copy
   * stays valid as a reference for later frames. */
  t0 = now_s();
  host_frame(cur, (int)cm->bit_depth, &h);
  SHIM_CHECK(s, cm, vp9hip_decoder_download(s->dec, slot_of_cur(s, cm), &h));
  if (s->gpu_lf) mark_resident(s, slot_of_cur(s, cm), cur, (int)cm->bit_depth);
  t_copy += now_s() - t0;
  g_trace.download += now_s() - t0;
  ++g_trace.frames;
  if (gpu_copy) *gpu_copy = t_copy;
  if (gpu_run) *gpu_run = (double)ms * 1e-3;
  return 0;
}
