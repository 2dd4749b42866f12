/*
 * vp9fe.c — VP9 bitstream front-end (include/vp9hip_fe.h): uncompressed + compressed header, tile
 * partitioning, boolean decoder, partition / mode / motion vector / coefficient token parse with their
 * contexts, backward adaptation of the probability contexts, reference map.  Output: the decode-order block
 * list, compact coefficient slots + eob plane and frame parameters the frame driver takes.
 *
 * Restated from the format's definition as the reference implements it (file:line cited at each part); own
 * structures: a block IS a vp9hip_block record (the output list doubles as the mode-info store, a grid of
 * indices stands for libvpx's MODE_INFO pointer grid), probabilities / counts are flat arrays per syntax
 * element, every tile column parses into a private segment of the lists (one thread each).
 */
#define _POSIX_C_SOURCE 200809L
#include <limits.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vp9hip_fe.h"

#include "vp9fe_tables.inc"

/* ---- names of the format's enumerations (libvpx/vp9/common/vp9_enums.h, vp9_blockd.h) ------------------- */
enum { INTRA_FRAME = 0, LAST_FRAME = 1, GOLDEN_FRAME = 2, ALTREF_FRAME = 3, NO_REF = -1 };
enum { DC_PRED = 0, TM_PRED = 9, NEARESTMV = 10, NEARMV = 11, ZEROMV = 12, NEWMV = 13 };
enum { BLOCK_4X4 = 0, BLOCK_4X8 = 1, BLOCK_8X4 = 2, BLOCK_8X8 = 3, BLOCK_64X64 = 12, BLOCK_INVALID = 13 };
enum { TX_4X4 = 0, TX_8X8 = 1, TX_16X16 = 2, TX_32X32 = 3 };
enum { ONLY_4X4 = 0, ALLOW_32X32 = 3, TX_MODE_SELECT = 4 };
enum { SINGLE_REFERENCE = 0, COMPOUND_REFERENCE = 1, REFERENCE_MODE_SELECT = 2 };
enum { SWITCHABLE_FILTERS = 3, SWITCHABLE = 4 };
enum { PARTITION_NONE = 0, PARTITION_HORZ = 1, PARTITION_VERT = 2, PARTITION_SPLIT = 3 };
enum { SEG_LVL_ALT_Q = 0, SEG_LVL_ALT_LF = 1, SEG_LVL_REF_FRAME = 2, SEG_LVL_SKIP = 3 };
enum { KEY_FRAME = 0, INTER_FRAME = 1 };
enum { MV_JOINT_ZERO = 0, MV_JOINT_HNZVZ = 1, MV_JOINT_HZVNZ = 2, MV_JOINT_HNZVNZ = 3 };
#define MAX_TILE_COLS 64
#define MV_LOW (-(1 << 14))
#define MV_UPP (1 << 14)

static const uint8_t kW8[13] = { 1, 1, 1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8 };  /* block width / height in 8x8 units */
static const uint8_t kH8[13] = { 1, 1, 1, 1, 2, 1, 2, 4, 2, 4, 8, 4, 8 };
static const uint8_t kModeLf[14] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 1 }; /* mode_lf_lut, vp9_loopfilter.c:207 */

/* ---- probabilities and counts: one flat array per syntax element ------------------------------------------ */
typedef struct {
  uint8_t y_mode[4][9], uv_mode[10][9], partition[16][3];
  uint8_t coef[4][2][2][6][6][3];
  uint8_t interp[4][2], inter_mode[7][3], intra_inter[4], comp_inter[5], single_ref[5][2], comp_ref[5];
  uint8_t tx8[2][1], tx16[2][2], tx32[2][3], skip[3];
  uint8_t mv_joints[3], mv_sign[2], mv_classes[2][10], mv_class0[2][1], mv_bits[2][10], mv_class0_fp[2][2][3],
      mv_fp[2][3], mv_class0_hp[2], mv_hp[2];
  uint8_t initialized;
} ProbCtx;

typedef struct {
  uint32_t y_mode[4][10], uv_mode[10][10], partition[16][4];
  uint32_t coef[4][2][2][6][6][4], eob_branch[4][2][2][6][6];
  uint32_t interp[4][3], inter_mode[7][4], intra_inter[4][2], comp_inter[5][2], single_ref[5][2][2], comp_ref[5][2];
  uint32_t tx8[2][2], tx16[2][3], tx32[2][4], skip[3][2];
  uint32_t mv_joints[4], mv_sign[2][2], mv_classes[2][11], mv_class0[2][2], mv_bits[2][10][2], mv_class0_fp[2][2][4],
      mv_fp[2][4], mv_class0_hp[2][2], mv_hp[2][2];
} Counts;

typedef struct {
  int8_t ref[2];
  int16_t mv[2][2];
} MvRef; /* one per 8x8 cell: what the next frame's candidate search reads (MV_REF, vp9_onyxc_int.h) */

typedef struct {
  int enabled, update_map, update_data, abs_delta, temporal_update;
  uint8_t tree_probs[7], pred_probs[3];
  uint8_t feature_mask[8];
  int16_t feature_data[8][4];
} Segmentation;

typedef struct {
  int width, height, ss_x, ss_y, bit_depth, valid;
} SlotInfo;

/* ---- boolean decoder (vpx_dsp/bitreader.h:60-140, bitreader.c:20-90) -------------------------------------- */
typedef struct {
  uint64_t value;
  unsigned range;
  int count;
  const uint8_t *buf, *end;
  int overrun;
} BoolDec;

static inline __attribute__((always_inline)) void bd_fill(BoolDec *r) {
  int shift = 64 - 8 - (r->count + 8);
  if (r->end - r->buf >= 8) { /* whole bytes that fit, in one load */
    uint64_t be;
    memcpy(&be, r->buf, 8);
    be = __builtin_bswap64(be);
    const int bits = (shift & ~7) + 8;
    r->value |= (be >> (64 - bits)) << (shift & 7);
    r->count += bits;
    r->buf += bits >> 3;
    return;
  }
  while (shift >= 0) {
    if (r->buf < r->end) {
      r->value |= (uint64_t)*r->buf++ << shift;
    } else {
      ++r->overrun; /* zeros past the end, as libvpx feeds them */
    }
    r->count += 8;
    shift -= 8;
  }
}

static inline __attribute__((always_inline)) int bd_read(BoolDec *r, int prob) {
  const unsigned split = (r->range * (unsigned)prob + (256 - (unsigned)prob)) >> 8;
  if (r->count < 0) bd_fill(r);
  const uint64_t bigsplit = (uint64_t)split << 56;
  unsigned range;
  int bit;
  if (r->value >= bigsplit) {
    range = r->range - split;
    r->value -= bigsplit;
    bit = 1;
  } else {
    range = split;
    bit = 0;
  }
  const int shift = __builtin_clz(range) - 24; /* vpx_norm[range]: range is 1..255 */
  r->range = range << shift;
  r->value <<= shift;
  r->count -= shift;
  return bit;
}
static inline int bd_bit(BoolDec *r) { return bd_read(r, 128); }
static int bd_literal(BoolDec *r, int bits) {
  int v = 0;
  for (int b = bits - 1; b >= 0; --b) v |= bd_bit(r) << b;
  return v;
}
static inline int bd_tree(BoolDec *r, const int8_t *tree, const uint8_t *probs) {
  int i = 0;
  while ((i = tree[i + bd_read(r, probs[i >> 1])]) > 0) {
  }
  return -i;
}
/* returns nonzero when the marker bit is set or there is no data */
static int bd_init(BoolDec *r, const uint8_t *data, size_t size) {
  if (size == 0) return 1;
  r->buf = data;
  r->end = data + size;
  r->value = 0;
  r->count = -8;
  r->range = 255;
  r->overrun = 0;
  bd_fill(r);
  return bd_bit(r) != 0;
}
/* libvpx: count > BD_VALUE_SIZE && count < LOTS_OF_BITS — more than the padding was consumed */
static inline int bd_error(const BoolDec *r) { return r->overrun > 8 + 2; }

/* ---- bit reader of the uncompressed header (vpx_dsp/bitreader_buffer.c) ------------------------------------ */
typedef struct {
  const uint8_t *buf;
  size_t bits, pos;
  int err;
} BitRd;
static int rb_bit(BitRd *r) {
  if (r->pos >= r->bits) {
    r->err = 1;
    return 0;
  }
  const int b = (r->buf[r->pos >> 3] >> (7 - (r->pos & 7))) & 1;
  ++r->pos;
  return b;
}
static int rb_lit(BitRd *r, int n) {
  int v = 0;
  for (int b = n - 1; b >= 0; --b) v |= rb_bit(r) << b;
  return v;
}
static int rb_signed(BitRd *r, int n) {
  const int v = rb_lit(r, n);
  return rb_bit(r) ? -v : v;
}

/* ---- per-frame header state ---------------------------------------------------------------------------- */
typedef struct {
  int profile, show_existing, frame_to_show;
  int frame_type, show_frame, error_res, intra_only, reset_frame_context;
  int refresh_flags, ref_idx[3];
  int allow_hp, interp_filter;
  int refresh_frame_context, frame_parallel, frame_context_idx;
  int filter_level, sharpness, mode_ref_delta_enabled;
  int base_qindex, y_dc_delta, uv_dc_delta, uv_ac_delta, lossless;
  int log2_tile_cols, log2_tile_rows;
  int tx_mode, reference_mode, comp_fixed_ref, comp_var_ref[2];
  size_t first_partition_size, header_bytes;
} FrameHdr;

typedef struct TileJob TileJob;

struct vp9hip_fe {
  char err[256];
  vp9hip_alloc_fn alloc;
  vp9hip_free_fn release;
  void *user;
  int max_threads;
  int trace;     /* VP9HIP_FE_TRACE: where a frame's parse time goes, printed when the front-end is destroyed */
  double tr_total, tr_head, tr_tiles, tr_tile_sum, tr_tile_max, tr_tail;
  int tr_frames;
  int narrow_slots; /* try int16 coefficient slots first (vp9hip_fe_set_narrow_slots) */
  int narrow_limit; /* 0, or (tests) a smaller magnitude that already counts as "does not fit" */
  int wide_frames;  /* frames that had to be parsed again with int32 slots */
  int checksums; /* VP9HIP_FE_CHECKSUMS: per-block checksum of eobs + coefficients in reserved2 (tests/test_fe_blocks.py) */

  /* stream state that outlives a frame */
  ProbCtx saved[4], fc;
  Segmentation seg;
  int8_t lf_ref_deltas[4], lf_mode_deltas[2];
  int ref_sign_bias[4];
  int ref_map[8];
  int prev_new_slot;
  SlotInfo slot[VP9HIP_FE_SLOTS];
  int width, height, ss_x, ss_y, bit_depth;  /* of the current / last frame (persist like VP9_COMMON's) */
  int mi_rows, mi_cols, sb_rows, sb_cols;
  int last_width, last_height, last_show_frame, last_frame_type, last_intra_only;
  int frame_type, intra_only, reset_frame_context, show_frame;
  int have_frame;  /* a frame has been decoded (frame_type etc. are meaningful) */
  int need_resync;
  uint8_t *seg_map[2];
  int seg_cur;
  size_t seg_cap;
  MvRef *mvs[2];
  int mv_cur, mv_rows[2], mv_cols[2]; /* allocated size of the two arrays */
  int mv_wr_rows[2], mv_wr_cols[2];    /* size of the frame that last wrote each (row pitch = its mi_cols) */
  size_t mv_cap[2];

  /* per frame */
  FrameHdr h;
  Counts counts;
  int use_prev_mvs;
  int16_t dq_y[8][2], dq_uv[8][2];
  uint8_t lvl[8][4][2];
  int ctx_cols;                 /* allocated columns (64-aligned, in 8x8 units) */
  uint8_t *above_nz[3];         /* entropy context per 4x4 column */
  uint8_t *above_part;          /* partition context per 8x8 column */
  vp9hip_block *seg_blocks;     /* per tile column segments: [mi_col_start * mi_rows ...) */
  uint32_t (*seg_off)[3];
  int32_t *grid;                /* index into seg_blocks per 8x8 cell, -1 = not decoded */
  int32_t *sb_count;            /* blocks per superblock */
  size_t cells_cap, sb_cap;
  vp9hip_block *out_blocks;     /* merged list (several tile columns) */
  uint32_t *out_off;
  int32_t *eob[3];
  int32_t *coef[3];
  /* the output arrays above are aliases of one of three sets used in rotation: a caller may still be packing frame
   * N - 1 and the device fetching the coefficients of frame N - 2 while frame N is parsed */
  struct OutSet {
    vp9hip_block *seg_blocks, *out_blocks;
    uint32_t (*seg_off)[3];
    uint32_t *out_off;
    int32_t *eob[3], *coef[3];
    size_t cells_cap, eob_cap[3], coef_cap[3]; /* of THIS set: a set only grows when its own turn comes (see ensure_frame_arrays) */
    vp9hip_coeff_region regions[3 * MAX_TILE_COLS];
  } sets[3];
  int set_idx;

  /* thread pool */
  pthread_t thr[16];
  int n_thr, pool_started, pool_stop;
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  TileJob *jobs;
  int n_jobs, next_job, done_jobs, epoch;
};

static int fe_fail(vp9hip_fe *fe, int rc, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(fe->err, sizeof(fe->err), fmt, ap);
  va_end(ap);
  return rc;
}
#define FE_FAIL(fe, ...) return fe_fail(fe, VP9HIP_EINVAL, __VA_ARGS__)

const char *vp9hip_fe_error(const vp9hip_fe *fe) { return fe ? fe->err : "null front-end"; }

/* ---- default contexts (vp9_setup_past_independence, vp9_entropymode.c:414-460) ---------------------------- */
static void default_probs(ProbCtx *p) {
  memcpy(p->coef, kDefCoef, sizeof(p->coef));
  memcpy(p->y_mode, kDefYMode, sizeof(p->y_mode));
  memcpy(p->uv_mode, kDefUvMode, sizeof(p->uv_mode));
  memcpy(p->partition, kDefPartition, sizeof(p->partition));
  memcpy(p->interp, kDefInterp, sizeof(p->interp));
  memcpy(p->inter_mode, kDefInterMode, sizeof(p->inter_mode));
  memcpy(p->intra_inter, kDefIntraInter, sizeof(p->intra_inter));
  memcpy(p->comp_inter, kDefCompInter, sizeof(p->comp_inter));
  memcpy(p->single_ref, kDefSingleRef, sizeof(p->single_ref));
  memcpy(p->comp_ref, kDefCompRef, sizeof(p->comp_ref));
  memcpy(p->tx8, kDefTx8, sizeof(p->tx8));
  memcpy(p->tx16, kDefTx16, sizeof(p->tx16));
  memcpy(p->tx32, kDefTx32, sizeof(p->tx32));
  memcpy(p->skip, kDefSkip, sizeof(p->skip));
  memcpy(p->mv_joints, kDefMvJoints, sizeof(p->mv_joints));
  memcpy(p->mv_sign, kDefMvSign, sizeof(p->mv_sign));
  memcpy(p->mv_classes, kDefMvClasses, sizeof(p->mv_classes));
  memcpy(p->mv_class0, kDefMvClass0, sizeof(p->mv_class0));
  memcpy(p->mv_bits, kDefMvBits, sizeof(p->mv_bits));
  memcpy(p->mv_class0_fp, kDefMvClass0Fp, sizeof(p->mv_class0_fp));
  memcpy(p->mv_fp, kDefMvFp, sizeof(p->mv_fp));
  memcpy(p->mv_class0_hp, kDefMvClass0Hp, sizeof(p->mv_class0_hp));
  memcpy(p->mv_hp, kDefMvHp, sizeof(p->mv_hp));
  p->initialized = 1;
}

static void seg_clear_features(Segmentation *s) {
  memset(s->feature_mask, 0, sizeof(s->feature_mask));
  memset(s->feature_data, 0, sizeof(s->feature_data));
}
static inline int seg_active(const Segmentation *s, int id, int feature) {
  return s->enabled && ((s->feature_mask[id] >> feature) & 1);
}

static void setup_past_independence(vp9hip_fe *fe) {
  FrameHdr *h = &fe->h;
  seg_clear_features(&fe->seg);
  fe->seg.abs_delta = 0;
  const size_t cells = (size_t)fe->mi_rows * fe->mi_cols;
  for (int k = 0; k < 2; ++k)
    if (fe->seg_map[k]) memset(fe->seg_map[k], 0, cells);
  fe->lf_ref_deltas[0] = 1;
  fe->lf_ref_deltas[1] = 0;
  fe->lf_ref_deltas[2] = -1;
  fe->lf_ref_deltas[3] = -1;
  fe->lf_mode_deltas[0] = fe->lf_mode_deltas[1] = 0;
  default_probs(&fe->fc);
  if (h->frame_type == KEY_FRAME || h->error_res || h->reset_frame_context == 3) {
    for (int i = 0; i < 4; ++i) fe->saved[i] = fe->fc;
  } else if (h->reset_frame_context == 2) {
    fe->saved[h->frame_context_idx] = fe->fc;
  }
  memset(fe->ref_sign_bias, 0, sizeof(fe->ref_sign_bias));
  h->frame_context_idx = 0;
}

/* ---- geometry ------------------------------------------------------------------------------------------ */
static void *fe_realloc_zero(void *old, size_t *cap, size_t want) {
  if (want <= *cap && old) return old;
  free(old);
  void *p = calloc(want ? want : 1, 1);
  *cap = p ? want : 0;
  return p;
}

/* resize_context_buffers (vp9_decodeframe.c:1705-1741) + what hangs off the frame size here */
static int set_frame_size(vp9hip_fe *fe, int width, int height) {
  if (width <= 0 || height <= 0 || width > 16384 || height > 16384) FE_FAIL(fe, "frame size %dx%d not supported", width, height);
  if (fe->width != width || fe->height != height) {
    const int mi_rows = (height + 7) >> 3, mi_cols = (width + 7) >> 3;
    const size_t cells = (size_t)mi_rows * mi_cols;
    if (cells > fe->seg_cap) {
      size_t c0 = fe->seg_cap, c1 = fe->seg_cap;
      fe->seg_map[0] = (uint8_t *)fe_realloc_zero(fe->seg_map[0], &c0, cells);
      fe->seg_map[1] = (uint8_t *)fe_realloc_zero(fe->seg_map[1], &c1, cells);
      if (!fe->seg_map[0] || !fe->seg_map[1]) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
      fe->seg_cap = cells;
    }
    fe->mi_rows = mi_rows;
    fe->mi_cols = mi_cols;
    fe->sb_rows = (mi_rows + 7) >> 3;
    fe->sb_cols = (mi_cols + 7) >> 3;
    /* vp9_init_context_buffers: the previous frame's segment map does not survive a size change */
    memset(fe->seg_map[fe->seg_cur ^ 1], 0, cells);
    fe->width = width;
    fe->height = height;
  }
  /* the frame's motion vector array (resize_mv_buffer: zeroed when it has to grow) */
  MvRef **mv = &fe->mvs[fe->mv_cur];
  if (!*mv || fe->mi_rows > fe->mv_rows[fe->mv_cur] || fe->mi_cols > fe->mv_cols[fe->mv_cur]) {
    const size_t want = (size_t)fe->mi_rows * fe->mi_cols * sizeof(MvRef);
    size_t cap = 0;
    *mv = (MvRef *)fe_realloc_zero(*mv, &cap, want);
    if (!*mv) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    fe->mv_rows[fe->mv_cur] = fe->mi_rows;
    fe->mv_cols[fe->mv_cur] = fe->mi_cols;
  }
  return VP9HIP_OK;
}

/* ---- uncompressed header (read_uncompressed_header, vp9_decodeframe.c:3114-3338) ----------------------- */
static int read_color_config(vp9hip_fe *fe, BitRd *rb) { /* read_bitdepth_colorspace_sampling :3054 */
  FrameHdr *h = &fe->h;
  fe->bit_depth = h->profile >= 2 ? (rb_bit(rb) ? 12 : 10) : 8;
  const int color_space = rb_lit(rb, 3);
  if (color_space != 7 /* sRGB */) {
    rb_bit(rb); /* color range */
    if (h->profile == 1 || h->profile == 3) {
      fe->ss_x = rb_bit(rb);
      fe->ss_y = rb_bit(rb);
      if (fe->ss_x == 1 && fe->ss_y == 1) FE_FAIL(fe, "4:2:0 colour not supported in profile 1 or 3");
      if (rb_bit(rb)) FE_FAIL(fe, "reserved bit set");
    } else {
      fe->ss_x = fe->ss_y = 1;
    }
  } else {
    if (h->profile == 1 || h->profile == 3) {
      fe->ss_x = fe->ss_y = 0;
      if (rb_bit(rb)) FE_FAIL(fe, "reserved bit set");
    } else {
      FE_FAIL(fe, "4:4:4 colour not supported in profile 0 or 2");
    }
  }
  if (fe->ss_x != fe->ss_y) FE_FAIL(fe, "4:2:2 / 4:4:0 subsampling is not supported by the reconstruction path");
  return VP9HIP_OK;
}

static int read_sync_code(BitRd *rb) { return rb_lit(rb, 8) == 0x49 && rb_lit(rb, 8) == 0x83 && rb_lit(rb, 8) == 0x42; }

static void read_render_size(BitRd *rb) {
  if (rb_bit(rb)) {
    rb_lit(rb, 16);
    rb_lit(rb, 16);
  }
}

static void tile_col_bits(int mi_cols, int *min_log2, int *max_log2) { /* vp9_get_tile_n_bits, vp9_tile_common.c:47 */
  const int sb_cols = (mi_cols + 7) >> 3;
  int mn = 0, mx = 1;
  while ((64 << mn) < sb_cols) ++mn;
  while ((sb_cols >> mx) >= 4) ++mx;
  *min_log2 = mn;
  *max_log2 = mx - 1;
}

/* a buffer no reference map entry names; not the previous frame's either — a caller may still be fetching that one
 * while this frame's kernels are queued */
static int find_free_slot(const vp9hip_fe *fe) {
  for (int s = 0; s < VP9HIP_FE_SLOTS; ++s) {
    int used = s == fe->prev_new_slot;
    for (int i = 0; i < 8; ++i) used |= fe->ref_map[i] == s;
    if (!used) return s;
  }
  return -1;
}

static int read_uncompressed_header(vp9hip_fe *fe, BitRd *rb, vp9hip_fe_frame *out) {
  FrameHdr *h = &fe->h;
  memset(h, 0, sizeof(*h));
  /* :3122-3123 — also ahead of a show_existing_frame */
  fe->last_frame_type = fe->frame_type;
  fe->last_intra_only = fe->intra_only;
  if (rb_lit(rb, 2) != 2) FE_FAIL(fe, "invalid frame marker");
  h->profile = rb_bit(rb);
  h->profile |= rb_bit(rb) << 1;
  if (h->profile > 2) h->profile += rb_bit(rb);
  if (h->profile > 3) FE_FAIL(fe, "unsupported bitstream profile");
  h->show_existing = rb_bit(rb);
  if (h->show_existing) {
    const int idx = rb_lit(rb, 3);
    h->frame_to_show = fe->ref_map[idx];
    if (h->frame_to_show < 0 || !fe->slot[h->frame_to_show].valid) FE_FAIL(fe, "buffer %d does not contain a decoded frame", idx);
    h->refresh_flags = 0;
    h->filter_level = 0;
    fe->show_frame = 1;
    return VP9HIP_OK;
  }
  h->frame_type = rb_bit(rb);
  h->show_frame = rb_bit(rb);
  h->error_res = rb_bit(rb);
  h->intra_only = fe->intra_only;                    /* persists over key frames, like cm->intra_only */
  h->reset_frame_context = fe->reset_frame_context;
  for (int i = 0; i < 3; ++i) h->ref_idx[i] = -1;
  int width = 0, height = 0;
  if (h->frame_type == KEY_FRAME) {
    if (!read_sync_code(rb)) FE_FAIL(fe, "invalid frame sync code");
    int rc = read_color_config(fe, rb);
    if (rc) return rc;
    h->refresh_flags = 0xff;
    width = rb_lit(rb, 16) + 1;
    height = rb_lit(rb, 16) + 1;
    if ((rc = set_frame_size(fe, width, height))) return rc;
    read_render_size(rb);
    if (fe->need_resync) {
      for (int i = 0; i < 8; ++i) fe->ref_map[i] = -1;
      fe->need_resync = 0;
    }
  } else {
    h->intra_only = h->show_frame ? 0 : rb_bit(rb);
    h->reset_frame_context = h->error_res ? 0 : rb_lit(rb, 2);
    if (h->intra_only) {
      if (!read_sync_code(rb)) FE_FAIL(fe, "invalid frame sync code");
      if (h->profile > 0) {
        int rc = read_color_config(fe, rb);
        if (rc) return rc;
      } else {
        fe->ss_x = fe->ss_y = 1;
        fe->bit_depth = 8;
      }
      h->refresh_flags = rb_lit(rb, 8);
      width = rb_lit(rb, 16) + 1;
      height = rb_lit(rb, 16) + 1;
      int rc = set_frame_size(fe, width, height);
      if (rc) return rc;
      read_render_size(rb);
      if (fe->need_resync) {
        for (int i = 0; i < 8; ++i) fe->ref_map[i] = -1;
        fe->need_resync = 0;
      }
    } else {
      if (fe->need_resync) FE_FAIL(fe, "key frame / intra-only frame required to reset decoder state");
      h->refresh_flags = rb_lit(rb, 8);
      for (int i = 0; i < 3; ++i) {
        const int slot = fe->ref_map[rb_lit(rb, 3)];
        if (slot < 0 || !fe->slot[slot].valid) FE_FAIL(fe, "reference %d is missing", i);
        h->ref_idx[i] = slot;
        fe->ref_sign_bias[LAST_FRAME + i] = rb_bit(rb);
      }
      /* setup_frame_size_with_refs :1781 */
      int found = 0;
      for (int i = 0; i < 3; ++i)
        if (rb_bit(rb)) {
          width = fe->slot[h->ref_idx[i]].width;
          height = fe->slot[h->ref_idx[i]].height;
          found = 1;
          break;
        }
      if (!found) {
        width = rb_lit(rb, 16) + 1;
        height = rb_lit(rb, 16) + 1;
      }
      int any_valid = 0;
      for (int i = 0; i < 3; ++i) {
        const SlotInfo *s = &fe->slot[h->ref_idx[i]];
        /* valid_ref_frame_size, vp9_onyxc_int.h: 2x larger .. 16x smaller */
        any_valid |= 2 * width >= s->width && 2 * height >= s->height && width <= 16 * s->width && height <= 16 * s->height;
        if (s->bit_depth != fe->bit_depth || s->ss_x != fe->ss_x || s->ss_y != fe->ss_y)
          FE_FAIL(fe, "referenced frame has incompatible colour format");
      }
      if (!any_valid) FE_FAIL(fe, "referenced frame has invalid size");
      int rc = set_frame_size(fe, width, height);
      if (rc) return rc;
      read_render_size(rb);
      h->allow_hp = rb_bit(rb);
      static const uint8_t literal_to_filter[4] = { 1, 0, 2, 3 };
      h->interp_filter = rb_bit(rb) ? SWITCHABLE : literal_to_filter[rb_lit(rb, 2)];
    }
  }
  if (!h->error_res) {
    h->refresh_frame_context = rb_bit(rb);
    h->frame_parallel = rb_bit(rb);
  } else {
    h->refresh_frame_context = 0;
    h->frame_parallel = 1;
  }
  h->frame_context_idx = rb_lit(rb, 2);
  if (h->frame_type == KEY_FRAME || h->intra_only || h->error_res) setup_past_independence(fe);

  /* setup_loopfilter :1610 */
  h->filter_level = rb_lit(rb, 6);
  h->sharpness = rb_lit(rb, 3);
  h->mode_ref_delta_enabled = rb_bit(rb);
  if (h->mode_ref_delta_enabled && rb_bit(rb)) {
    for (int i = 0; i < 4; ++i)
      if (rb_bit(rb)) fe->lf_ref_deltas[i] = (int8_t)rb_signed(rb, 6);
    for (int i = 0; i < 2; ++i)
      if (rb_bit(rb)) fe->lf_mode_deltas[i] = (int8_t)rb_signed(rb, 6);
  }
  /* setup_quantization :1640 */
  h->base_qindex = rb_lit(rb, 8);
  h->y_dc_delta = rb_bit(rb) ? rb_signed(rb, 4) : 0;
  h->uv_dc_delta = rb_bit(rb) ? rb_signed(rb, 4) : 0;
  h->uv_ac_delta = rb_bit(rb) ? rb_signed(rb, 4) : 0;
  h->lossless = h->base_qindex == 0 && h->y_dc_delta == 0 && h->uv_dc_delta == 0 && h->uv_ac_delta == 0;
  /* setup_segmentation :1560 */
  Segmentation *sg = &fe->seg;
  sg->update_map = 0;
  sg->update_data = 0;
  sg->enabled = rb_bit(rb);
  if (sg->enabled) {
    sg->update_map = rb_bit(rb);
    if (sg->update_map) {
      for (int i = 0; i < 7; ++i) sg->tree_probs[i] = rb_bit(rb) ? (uint8_t)rb_lit(rb, 8) : 255;
      sg->temporal_update = rb_bit(rb);
      for (int i = 0; i < 3; ++i) sg->pred_probs[i] = sg->temporal_update ? (rb_bit(rb) ? (uint8_t)rb_lit(rb, 8) : 255) : 255;
    }
    sg->update_data = rb_bit(rb);
    if (sg->update_data) {
      static const uint8_t bits[4] = { 8, 6, 2, 0 };
      static const int16_t maxv[4] = { 255, 63, 3, 0 };
      sg->abs_delta = rb_bit(rb);
      seg_clear_features(sg);
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) {
          int data = 0;
          if (rb_bit(rb)) {
            sg->feature_mask[i] |= (uint8_t)(1 << j);
            data = rb_lit(rb, bits[j]);
            if (data > maxv[j]) data = maxv[j];
            if (j < 2 && rb_bit(rb)) data = -data;
          }
          sg->feature_data[i][j] = (int16_t)data;
        }
    }
  }
  /* setup_tile_info :1857 */
  int mn, mx;
  tile_col_bits(fe->mi_cols, &mn, &mx);
  h->log2_tile_cols = mn;
  for (int ones = mx - mn; ones-- > 0 && rb_bit(rb);) ++h->log2_tile_cols;
  if (h->log2_tile_cols > 6) FE_FAIL(fe, "invalid number of tile columns");
  h->log2_tile_rows = rb_bit(rb);
  if (h->log2_tile_rows) h->log2_tile_rows += rb_bit(rb);
  h->first_partition_size = (size_t)rb_lit(rb, 16);
  if (rb->err) FE_FAIL(fe, "truncated uncompressed header");
  if (h->first_partition_size == 0) FE_FAIL(fe, "invalid header size");
  h->header_bytes = (rb->pos + 7) >> 3;
  (void)out;
  return VP9HIP_OK;
}

/* ---- compressed header (read_compressed_header :3340, vp9_dsubexp.c) ----------------------------------- */
static uint8_t g_inv_map[255];
static pthread_once_t g_inv_once = PTHREAD_ONCE_INIT;
static void build_inv_map(void) { /* the 20 coarse steps first, every other value after them in ascending order */
  int n = 0;
  uint8_t taken[256] = { 0 };
  for (int i = 0; i < 20; ++i) {
    g_inv_map[n++] = (uint8_t)(7 + 13 * i);
    taken[7 + 13 * i] = 1;
  }
  for (int v = 1; v <= 253 && n < 255; ++v)
    if (!taken[v]) g_inv_map[n++] = (uint8_t)v;
  while (n < 255) g_inv_map[n++] = 253;
}
static int inv_recenter(int v, int m) {
  if (v > 2 * m) return v;
  return (v & 1) ? m - ((v + 1) >> 1) : m + (v >> 1);
}
static void diff_update(BoolDec *r, uint8_t *p) {
  if (!bd_read(r, 252)) return;
  int delp;
  if (!bd_bit(r)) {
    delp = bd_literal(r, 4);
  } else if (!bd_bit(r)) {
    delp = bd_literal(r, 4) + 16;
  } else if (!bd_bit(r)) {
    delp = bd_literal(r, 5) + 32;
  } else {
    const int v = bd_literal(r, 7);
    delp = (v < 65 ? v : (v << 1) - 65 + bd_bit(r)) + 64;
  }
  if (delp > 254) delp = 254;
  const int v = g_inv_map[delp];
  int m = *p - 1;
  if ((m << 1) <= 255)
    *p = (uint8_t)(1 + inv_recenter(v, m));
  else
    *p = (uint8_t)(255 - inv_recenter(v, 255 - 1 - m));
}
static void update_mv_probs(BoolDec *r, uint8_t *p, int n) {
  for (int i = 0; i < n; ++i)
    if (bd_read(r, 252)) p[i] = (uint8_t)((bd_literal(r, 7) << 1) | 1);
}

static int compound_allowed(const vp9hip_fe *fe) { /* vp9_pred_common.c:16 */
  for (int i = 1; i < 3; ++i)
    if (fe->ref_sign_bias[i + 1] != fe->ref_sign_bias[1]) return 1;
  return 0;
}

static int read_compressed_header(vp9hip_fe *fe, const uint8_t *data, size_t size) {
  FrameHdr *h = &fe->h;
  ProbCtx *fc = &fe->fc;
  BoolDec r;
  pthread_once(&g_inv_once, build_inv_map);
  if (bd_init(&r, data, size)) FE_FAIL(fe, "invalid compressed header");
  if (h->lossless) {
    h->tx_mode = ONLY_4X4;
  } else {
    h->tx_mode = bd_literal(&r, 2);
    if (h->tx_mode == ALLOW_32X32) h->tx_mode += bd_bit(&r);
  }
  if (h->tx_mode == TX_MODE_SELECT) {
    for (int i = 0; i < 2; ++i) diff_update(&r, &fc->tx8[i][0]);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) diff_update(&r, &fc->tx16[i][j]);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) diff_update(&r, &fc->tx32[i][j]);
  }
  for (int tx = 0; tx <= kTxModeBiggest[h->tx_mode]; ++tx)
    if (bd_bit(&r))
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int k = 0; k < 6; ++k)
            for (int l = 0; l < (k == 0 ? 3 : 6); ++l)
              for (int m = 0; m < 3; ++m) diff_update(&r, &fc->coef[tx][i][j][k][l][m]);
  for (int k = 0; k < 3; ++k) diff_update(&r, &fc->skip[k]);
  if (!(h->frame_type == KEY_FRAME || h->intra_only)) {
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 3; ++j) diff_update(&r, &fc->inter_mode[i][j]);
    if (h->interp_filter == SWITCHABLE)
      for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 2; ++i) diff_update(&r, &fc->interp[j][i]);
    for (int i = 0; i < 4; ++i) diff_update(&r, &fc->intra_inter[i]);
    h->reference_mode = SINGLE_REFERENCE;
    if (compound_allowed(fe)) h->reference_mode = bd_bit(&r) ? (bd_bit(&r) ? REFERENCE_MODE_SELECT : COMPOUND_REFERENCE) : SINGLE_REFERENCE;
    if (h->reference_mode != SINGLE_REFERENCE) { /* vp9_setup_compound_reference_mode, vp9_pred_common.c:24 */
      if (fe->ref_sign_bias[LAST_FRAME] == fe->ref_sign_bias[GOLDEN_FRAME]) {
        h->comp_fixed_ref = ALTREF_FRAME;
        h->comp_var_ref[0] = LAST_FRAME;
        h->comp_var_ref[1] = GOLDEN_FRAME;
      } else if (fe->ref_sign_bias[LAST_FRAME] == fe->ref_sign_bias[ALTREF_FRAME]) {
        h->comp_fixed_ref = GOLDEN_FRAME;
        h->comp_var_ref[0] = LAST_FRAME;
        h->comp_var_ref[1] = ALTREF_FRAME;
      } else {
        h->comp_fixed_ref = LAST_FRAME;
        h->comp_var_ref[0] = GOLDEN_FRAME;
        h->comp_var_ref[1] = ALTREF_FRAME;
      }
    }
    if (h->reference_mode == REFERENCE_MODE_SELECT)
      for (int i = 0; i < 5; ++i) diff_update(&r, &fc->comp_inter[i]);
    if (h->reference_mode != COMPOUND_REFERENCE)
      for (int i = 0; i < 5; ++i) {
        diff_update(&r, &fc->single_ref[i][0]);
        diff_update(&r, &fc->single_ref[i][1]);
      }
    if (h->reference_mode != SINGLE_REFERENCE)
      for (int i = 0; i < 5; ++i) diff_update(&r, &fc->comp_ref[i]);
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 9; ++i) diff_update(&r, &fc->y_mode[j][i]);
    for (int j = 0; j < 16; ++j)
      for (int i = 0; i < 3; ++i) diff_update(&r, &fc->partition[j][i]);
    /* read_mv_probs :144 */
    update_mv_probs(&r, fc->mv_joints, 3);
    for (int i = 0; i < 2; ++i) {
      update_mv_probs(&r, &fc->mv_sign[i], 1);
      update_mv_probs(&r, fc->mv_classes[i], 10);
      update_mv_probs(&r, fc->mv_class0[i], 1);
      update_mv_probs(&r, fc->mv_bits[i], 10);
    }
    for (int i = 0; i < 2; ++i) {
      for (int j = 0; j < 2; ++j) update_mv_probs(&r, fc->mv_class0_fp[i][j], 3);
      update_mv_probs(&r, fc->mv_fp[i], 3);
    }
    if (h->allow_hp)
      for (int i = 0; i < 2; ++i) {
        update_mv_probs(&r, &fc->mv_class0_hp[i], 1);
        update_mv_probs(&r, &fc->mv_hp[i], 1);
      }
  }
  if (bd_error(&r)) FE_FAIL(fe, "compressed header is corrupted");
  return VP9HIP_OK;
}

/* ---- one tile column's parse state ------------------------------------------------------------------- */
typedef struct {
  const uint8_t *data;
  size_t size;
} TileBuf;

typedef struct TileCtx {
  vp9hip_fe *fe;
  BoolDec bd;
  int tile_col, col_start, col_end; /* 8x8 units */
  uint8_t left_nz[3][16];
  uint8_t left_part[8];
  Counts *counts; /* NULL: frame-parallel mode collects nothing */
  vp9hip_block *seg_first, *blk;
  uint32_t (*off)[3];
  int32_t *cf[3], *cf_base[3];
  int32_t scratch[1024];
  int corrupt;
  int narrow;           /* slots are written as int16 (vp9hip_coeff_layout.narrow) ... */
  uint32_t out_of_range; /* ... and this is non-zero once a coefficient did not fit */
  /* the block being read */
  vp9hip_block *cur;
  const vp9hip_block *above, *left;
  int mi_row, mi_col, bw8, bh8;
  int to_left, to_right, to_top, to_bottom; /* distance to the frame edges, 1/8 sample (mb_to_*_edge) */
} TileCtx;

struct TileJob {
  double seconds; /* VP9HIP_FE_TRACE */
  TileCtx tc;
  TileBuf buf[4]; /* per tile row */
  Counts counts;
  int64_t cf_start[3], cf_used[3];
  int n_blocks;
};

static inline const vp9hip_block *cell_block(const vp9hip_fe *fe, int mi_row, int mi_col) {
  const int32_t g = fe->grid[(size_t)mi_row * fe->mi_cols + mi_col];
  return g >= 0 ? &fe->seg_blocks[g] : NULL;
}
static inline int is_inter(const vp9hip_block *b) { return b->ref_frame[0] > INTRA_FRAME; }
static inline int has_second(const vp9hip_block *b) { return b->ref_frame[1] > INTRA_FRAME; }
#define SEG_ID(b) ((b)->reserved[0])
#define SEG_PRED(b) ((b)->reserved[1])

/* ---- contexts (vp9_pred_common.h / .c) ----------------------------------------------------------------- */
static int ctx_skip(const TileCtx *t) { return (t->above ? t->above->skip : 0) + (t->left ? t->left->skip : 0); }

static int ctx_intra_inter(const TileCtx *t) { /* vp9_pred_common.h:93 */
  const vp9hip_block *a = t->above, *l = t->left;
  if (a && l) {
    const int ai = !is_inter(a), li = !is_inter(l);
    return (ai && li) ? 3 : (ai || li);
  }
  if (a || l) return 2 * !is_inter(a ? a : l);
  return 0;
}

static int ctx_interp(const TileCtx *t) { /* vp9_pred_common.h:67 */
  const int lt = t->left ? t->left->interp_filter : SWITCHABLE_FILTERS;
  const int at = t->above ? t->above->interp_filter : SWITCHABLE_FILTERS;
  if (lt == at) return lt;
  if (lt == SWITCHABLE_FILTERS) return at;
  if (at == SWITCHABLE_FILTERS) return lt;
  return SWITCHABLE_FILTERS;
}

static int ctx_tx_size(const TileCtx *t, int max_tx) { /* vp9_pred_common.h:157 */
  const vp9hip_block *a = t->above, *l = t->left;
  int ac = (a && !a->skip) ? a->tx_size : max_tx;
  int lc = (l && !l->skip) ? l->tx_size : max_tx;
  if (!l) lc = ac;
  if (!a) ac = lc;
  return (ac + lc) > max_tx;
}

static int ctx_reference_mode(const TileCtx *t) { /* vp9_get_reference_mode_context, vp9_pred_common.c:42 */
  const vp9hip_block *a = t->above, *l = t->left;
  const int fixed = t->fe->h.comp_fixed_ref;
  if (a && l) {
    if (!has_second(a) && !has_second(l)) return (a->ref_frame[0] == fixed) ^ (l->ref_frame[0] == fixed);
    if (!has_second(a)) return 2 + (a->ref_frame[0] == fixed || !is_inter(a));
    if (!has_second(l)) return 2 + (l->ref_frame[0] == fixed || !is_inter(l));
    return 4;
  }
  if (a || l) {
    const vp9hip_block *e = a ? a : l;
    return has_second(e) ? 3 : e->ref_frame[0] == fixed;
  }
  return 1;
}

static int ctx_comp_ref(const TileCtx *t) { /* vp9_get_pred_context_comp_ref_p, vp9_pred_common.c:84 */
  const vp9hip_block *a = t->above, *l = t->left;
  const FrameHdr *h = &t->fe->h;
  const int var_idx = !t->fe->ref_sign_bias[h->comp_fixed_ref];
  const int v1 = h->comp_var_ref[1], v0 = h->comp_var_ref[0], fixed = h->comp_fixed_ref;
  if (a && l) {
    const int ai = !is_inter(a), li = !is_inter(l);
    if (ai && li) return 2;
    if (ai || li) {
      const vp9hip_block *e = ai ? l : a;
      return 1 + 2 * ((has_second(e) ? e->ref_frame[var_idx] : e->ref_frame[0]) != v1);
    }
    const int l_sg = !has_second(l), a_sg = !has_second(a);
    const int vrfa = a_sg ? a->ref_frame[0] : a->ref_frame[var_idx];
    const int vrfl = l_sg ? l->ref_frame[0] : l->ref_frame[var_idx];
    if (vrfa == vrfl && v1 == vrfa) return 0;
    if (l_sg && a_sg) {
      if ((vrfa == fixed && vrfl == v0) || (vrfl == fixed && vrfa == v0)) return 4;
      return vrfa == vrfl ? 3 : 1;
    }
    if (l_sg || a_sg) {
      const int vrfc = l_sg ? vrfa : vrfl, rfs = a_sg ? vrfa : vrfl;
      if (vrfc == v1 && rfs != v1) return 1;
      if (rfs == v1 && vrfc != v1) return 2;
      return 4;
    }
    return vrfa == vrfl ? 4 : 2;
  }
  if (a || l) {
    const vp9hip_block *e = a ? a : l;
    if (!is_inter(e)) return 2;
    return has_second(e) ? 4 * (e->ref_frame[var_idx] != v1) : 3 * (e->ref_frame[0] != v1);
  }
  return 2;
}

static int ctx_single_ref_p1(const TileCtx *t) { /* vp9_pred_common.c:166 */
  const vp9hip_block *a = t->above, *l = t->left;
#define USES_LAST(b) ((b)->ref_frame[0] == LAST_FRAME || (b)->ref_frame[1] == LAST_FRAME)
  if (a && l) {
    const int ai = !is_inter(a), li = !is_inter(l);
    if (ai && li) return 2;
    if (ai || li) {
      const vp9hip_block *e = ai ? l : a;
      return has_second(e) ? 1 + USES_LAST(e) : 4 * (e->ref_frame[0] == LAST_FRAME);
    }
    const int a2 = has_second(a), l2 = has_second(l);
    if (a2 && l2) return 1 + (USES_LAST(a) || USES_LAST(l));
    if (a2 || l2) {
      const vp9hip_block *single = a2 ? l : a, *comp = a2 ? a : l;
      return (single->ref_frame[0] == LAST_FRAME ? 3 : 0) + USES_LAST(comp);
    }
    return 2 * (a->ref_frame[0] == LAST_FRAME) + 2 * (l->ref_frame[0] == LAST_FRAME);
  }
  if (a || l) {
    const vp9hip_block *e = a ? a : l;
    if (!is_inter(e)) return 2;
    return has_second(e) ? 1 + USES_LAST(e) : 4 * (e->ref_frame[0] == LAST_FRAME);
  }
  return 2;
#undef USES_LAST
}

static int ctx_single_ref_p2(const TileCtx *t) { /* vp9_pred_common.c:232 */
  const vp9hip_block *a = t->above, *l = t->left;
#define USES_GOLD(b) ((b)->ref_frame[0] == GOLDEN_FRAME || (b)->ref_frame[1] == GOLDEN_FRAME)
  if (a && l) {
    const int ai = !is_inter(a), li = !is_inter(l);
    if (ai && li) return 2;
    if (ai || li) {
      const vp9hip_block *e = ai ? l : a;
      if (has_second(e)) return 1 + 2 * USES_GOLD(e);
      return e->ref_frame[0] == LAST_FRAME ? 3 : 4 * (e->ref_frame[0] == GOLDEN_FRAME);
    }
    const int a2 = has_second(a), l2 = has_second(l);
    const int a0 = a->ref_frame[0], a1 = a->ref_frame[1], l0 = l->ref_frame[0], l1 = l->ref_frame[1];
    if (a2 && l2) return (a0 == l0 && a1 == l1) ? 3 * (USES_GOLD(a) || USES_GOLD(l)) : 2;
    if (a2 || l2) {
      const int rfs = a2 ? l0 : a0;
      const int g = a2 ? USES_GOLD(a) : USES_GOLD(l);
      if (rfs == GOLDEN_FRAME) return 3 + g;
      if (rfs == ALTREF_FRAME) return g;
      return 1 + 2 * g;
    }
    if (a0 == LAST_FRAME && l0 == LAST_FRAME) return 3;
    if (a0 == LAST_FRAME || l0 == LAST_FRAME) return 4 * ((a0 == LAST_FRAME ? l0 : a0) == GOLDEN_FRAME);
    return 2 * (a0 == GOLDEN_FRAME) + 2 * (l0 == GOLDEN_FRAME);
  }
  if (a || l) {
    const vp9hip_block *e = a ? a : l;
    if (!is_inter(e) || (e->ref_frame[0] == LAST_FRAME && !has_second(e))) return 2;
    return has_second(e) ? 3 * USES_GOLD(e) : 4 * (e->ref_frame[0] == GOLDEN_FRAME);
  }
  return 2;
#undef USES_GOLD
}

/* Test hook (include/vp9hip_fe.h): the eight neighbour contexts of one (above, left) pair */
void vp9hip_fe_debug_contexts(const vp9hip_block *above, const vp9hip_block *left, const int32_t sign_bias[3], int max_tx,
                              int32_t out[8]) {
  vp9hip_fe *fe = (vp9hip_fe *)calloc(1, sizeof(*fe));
  TileCtx t;
  if (!fe) return;
  memset(&t, 0, sizeof(t));
  t.fe = fe;
  t.above = above;
  t.left = left;
  for (int i = 0; i < 3; ++i) fe->ref_sign_bias[1 + i] = sign_bias[i];
  /* vp9_setup_compound_reference_mode, as read_compressed_header does it */
  FrameHdr *h = &fe->h;
  if (fe->ref_sign_bias[LAST_FRAME] == fe->ref_sign_bias[GOLDEN_FRAME]) {
    h->comp_fixed_ref = ALTREF_FRAME;
    h->comp_var_ref[0] = LAST_FRAME;
    h->comp_var_ref[1] = GOLDEN_FRAME;
  } else if (fe->ref_sign_bias[LAST_FRAME] == fe->ref_sign_bias[ALTREF_FRAME]) {
    h->comp_fixed_ref = GOLDEN_FRAME;
    h->comp_var_ref[0] = LAST_FRAME;
    h->comp_var_ref[1] = ALTREF_FRAME;
  } else {
    h->comp_fixed_ref = LAST_FRAME;
    h->comp_var_ref[0] = GOLDEN_FRAME;
    h->comp_var_ref[1] = ALTREF_FRAME;
  }
  out[0] = ctx_skip(&t);
  out[1] = ctx_intra_inter(&t);
  out[2] = ctx_interp(&t);
  out[3] = ctx_tx_size(&t, max_tx);
  out[4] = ctx_reference_mode(&t);
  out[5] = ctx_comp_ref(&t);
  out[6] = ctx_single_ref_p1(&t);
  out[7] = ctx_single_ref_p2(&t);
  free(fe);
}

/* ---- segment ids (vp9_decodemv.c:93-176) ---------------------------------------------------------------- */
static void seg_cells_set(uint8_t *map, int cols, int mi_row, int mi_col, int x_mis, int y_mis, int id) {
  for (int y = 0; y < y_mis; ++y) memset(map + (size_t)(mi_row + y) * cols + mi_col, id, (size_t)x_mis);
}
static void seg_cells_copy(uint8_t *dst, const uint8_t *src, int cols, int mi_row, int mi_col, int x_mis, int y_mis) {
  for (int y = 0; y < y_mis; ++y) {
    const size_t o = (size_t)(mi_row + y) * cols + mi_col;
    memcpy(dst + o, src + o, (size_t)x_mis);
  }
}
static int read_segment_tree(TileCtx *t) { return bd_tree(&t->bd, kSegmentTree, t->fe->seg.tree_probs); }

static int read_intra_segment_id(TileCtx *t, int x_mis, int y_mis) {
  vp9hip_fe *fe = t->fe;
  if (!fe->seg.enabled) return 0;
  uint8_t *cur = fe->seg_map[fe->seg_cur];
  if (!fe->seg.update_map) {
    seg_cells_copy(cur, fe->seg_map[fe->seg_cur ^ 1], fe->mi_cols, t->mi_row, t->mi_col, x_mis, y_mis);
    return 0;
  }
  const int id = read_segment_tree(t);
  seg_cells_set(cur, fe->mi_cols, t->mi_row, t->mi_col, x_mis, y_mis, id);
  return id;
}

static int read_inter_segment_id(TileCtx *t, int x_mis, int y_mis) {
  vp9hip_fe *fe = t->fe;
  if (!fe->seg.enabled) return 0;
  uint8_t *cur = fe->seg_map[fe->seg_cur];
  const uint8_t *last = fe->seg_map[fe->seg_cur ^ 1];
  int predicted = 8;
  for (int y = 0; y < y_mis; ++y)
    for (int x = 0; x < x_mis; ++x) {
      const int v = last[(size_t)(t->mi_row + y) * fe->mi_cols + t->mi_col + x];
      if (v < predicted) predicted = v;
    }
  if (predicted > 7) predicted = 7;
  if (!fe->seg.update_map) {
    seg_cells_copy(cur, last, fe->mi_cols, t->mi_row, t->mi_col, x_mis, y_mis);
    return predicted;
  }
  int id;
  if (fe->seg.temporal_update) {
    const int ctx = (t->above ? SEG_PRED(t->above) : 0) + (t->left ? SEG_PRED(t->left) : 0);
    const int flag = bd_read(&t->bd, fe->seg.pred_probs[ctx]);
    SEG_PRED(t->cur) = (uint8_t)flag;
    id = flag ? predicted : read_segment_tree(t);
  } else {
    id = read_segment_tree(t);
  }
  seg_cells_set(cur, fe->mi_cols, t->mi_row, t->mi_col, x_mis, y_mis, id);
  return id;
}

static int read_skip(TileCtx *t, int segment_id) {
  if (seg_active(&t->fe->seg, segment_id, SEG_LVL_SKIP)) return 1;
  const int ctx = ctx_skip(t);
  const int skip = bd_read(&t->bd, t->fe->fc.skip[ctx]);
  if (t->counts) ++t->counts->skip[ctx][skip];
  return skip;
}

static int read_tx_size(TileCtx *t, int allow_select) { /* vp9_decodemv.c:65-91 */
  const FrameHdr *h = &t->fe->h;
  const int bsize = t->cur->sb_type, max_tx = kMaxTxSize[bsize];
  if (!(allow_select && h->tx_mode == TX_MODE_SELECT && bsize >= BLOCK_8X8))
    return max_tx < kTxModeBiggest[h->tx_mode] ? max_tx : kTxModeBiggest[h->tx_mode];
  const int ctx = ctx_tx_size(t, max_tx);
  const ProbCtx *fc = &t->fe->fc;
  const uint8_t *p = max_tx == TX_8X8 ? fc->tx8[ctx] : max_tx == TX_16X16 ? fc->tx16[ctx] : fc->tx32[ctx];
  int tx = bd_read(&t->bd, p[0]);
  if (tx != TX_4X4 && max_tx >= TX_16X16) {
    tx += bd_read(&t->bd, p[1]);
    if (tx != TX_8X8 && max_tx >= TX_32X32) tx += bd_read(&t->bd, p[2]);
  }
  if (t->counts) {
    if (max_tx == TX_8X8)
      ++t->counts->tx8[ctx][tx];
    else if (max_tx == TX_16X16)
      ++t->counts->tx16[ctx][tx];
    else
      ++t->counts->tx32[ctx][tx];
  }
  return tx;
}

/* ---- intra frames (read_intra_frame_mode_info, vp9_decodemv.c:192) ----------------------------------------- */
static int kf_left_mode(const vp9hip_block *cur, const vp9hip_block *left, int b) { /* vp9_blockd.c: left / above_block_mode */
  if (b == 0 || b == 2) {
    if (!left || is_inter(left)) return DC_PRED;
    return left->sb_type < BLOCK_8X8 ? left->sub_mode[b + 1] : left->mode;
  }
  return cur->sub_mode[b - 1];
}
static int kf_above_mode(const vp9hip_block *cur, const vp9hip_block *above, int b) {
  if (b == 0 || b == 1) {
    if (!above || is_inter(above)) return DC_PRED;
    return above->sb_type < BLOCK_8X8 ? above->sub_mode[b + 2] : above->mode;
  }
  return cur->sub_mode[b - 2];
}
static int read_kf_mode(TileCtx *t, int b) {
  return bd_tree(&t->bd, kIntraModeTree, kKfYMode[kf_above_mode(t->cur, t->above, b)][kf_left_mode(t->cur, t->left, b)]);
}

static void read_intra_frame_mode_info(TileCtx *t, int x_mis, int y_mis) {
  vp9hip_block *b = t->cur;
  SEG_ID(b) = (uint8_t)read_intra_segment_id(t, x_mis, y_mis);
  b->skip = (uint8_t)read_skip(t, SEG_ID(b));
  b->tx_size = (uint8_t)read_tx_size(t, 1);
  b->ref_frame[0] = INTRA_FRAME;
  b->ref_frame[1] = NO_REF;
  switch (b->sb_type) {
    case BLOCK_4X4:
      for (int i = 0; i < 4; ++i) b->sub_mode[i] = (uint8_t)read_kf_mode(t, i);
      break;
    case BLOCK_4X8:
      b->sub_mode[0] = b->sub_mode[2] = (uint8_t)read_kf_mode(t, 0);
      b->sub_mode[1] = b->sub_mode[3] = (uint8_t)read_kf_mode(t, 1);
      break;
    case BLOCK_8X4:
      b->sub_mode[0] = b->sub_mode[1] = (uint8_t)read_kf_mode(t, 0);
      b->sub_mode[2] = b->sub_mode[3] = (uint8_t)read_kf_mode(t, 2);
      break;
    default:
      b->sub_mode[0] = b->sub_mode[1] = b->sub_mode[2] = b->sub_mode[3] = (uint8_t)read_kf_mode(t, 0);
  }
  b->mode = b->sub_mode[3];
  b->uv_mode = (uint8_t)bd_tree(&t->bd, kIntraModeTree, kKfUvMode[b->mode]);
  b->interp_filter = 0;
}

/* ---- inter frames -------------------------------------------------------------------------------------- */
static int read_y_mode(TileCtx *t, int group) {
  const int m = bd_tree(&t->bd, kIntraModeTree, t->fe->fc.y_mode[group]);
  if (t->counts) ++t->counts->y_mode[group][m];
  return m;
}

static void read_intra_block_mode_info(TileCtx *t) { /* vp9_decodemv.c:347 */
  vp9hip_block *b = t->cur;
  switch (b->sb_type) {
    case BLOCK_4X4:
      for (int i = 0; i < 4; ++i) b->sub_mode[i] = (uint8_t)read_y_mode(t, 0);
      break;
    case BLOCK_4X8:
      b->sub_mode[0] = b->sub_mode[2] = (uint8_t)read_y_mode(t, 0);
      b->sub_mode[1] = b->sub_mode[3] = (uint8_t)read_y_mode(t, 0);
      break;
    case BLOCK_8X4:
      b->sub_mode[0] = b->sub_mode[1] = (uint8_t)read_y_mode(t, 0);
      b->sub_mode[2] = b->sub_mode[3] = (uint8_t)read_y_mode(t, 0);
      break;
    default:
      b->sub_mode[0] = b->sub_mode[1] = b->sub_mode[2] = b->sub_mode[3] = (uint8_t)read_y_mode(t, kSizeGroup[b->sb_type]);
  }
  b->mode = b->sub_mode[3];
  b->uv_mode = (uint8_t)bd_tree(&t->bd, kIntraModeTree, t->fe->fc.uv_mode[b->mode]);
  if (t->counts) ++t->counts->uv_mode[b->mode][b->uv_mode];
  b->interp_filter = SWITCHABLE_FILTERS; /* what the filter context of later blocks reads for an intra block */
  b->ref_frame[0] = INTRA_FRAME;
  b->ref_frame[1] = NO_REF;
}

static void read_ref_frames(TileCtx *t, int segment_id) { /* vp9_decodemv.c:290 */
  vp9hip_fe *fe = t->fe;
  vp9hip_block *b = t->cur;
  const FrameHdr *h = &fe->h;
  if (seg_active(&fe->seg, segment_id, SEG_LVL_REF_FRAME)) {
    b->ref_frame[0] = (int8_t)fe->seg.feature_data[segment_id][SEG_LVL_REF_FRAME];
    b->ref_frame[1] = NO_REF;
    return;
  }
  int mode = h->reference_mode;
  if (mode == REFERENCE_MODE_SELECT) {
    const int ctx = ctx_reference_mode(t);
    mode = bd_read(&t->bd, fe->fc.comp_inter[ctx]);
    if (t->counts) ++t->counts->comp_inter[ctx][mode];
  }
  if (mode == COMPOUND_REFERENCE) {
    const int idx = fe->ref_sign_bias[h->comp_fixed_ref];
    const int ctx = ctx_comp_ref(t);
    const int bit = bd_read(&t->bd, fe->fc.comp_ref[ctx]);
    if (t->counts) ++t->counts->comp_ref[ctx][bit];
    b->ref_frame[idx] = (int8_t)h->comp_fixed_ref;
    b->ref_frame[!idx] = (int8_t)h->comp_var_ref[bit];
  } else {
    const int ctx0 = ctx_single_ref_p1(t);
    const int bit0 = bd_read(&t->bd, fe->fc.single_ref[ctx0][0]);
    if (t->counts) ++t->counts->single_ref[ctx0][0][bit0];
    if (bit0) {
      const int ctx1 = ctx_single_ref_p2(t);
      const int bit1 = bd_read(&t->bd, fe->fc.single_ref[ctx1][1]);
      if (t->counts) ++t->counts->single_ref[ctx1][1][bit1];
      b->ref_frame[0] = bit1 ? ALTREF_FRAME : GOLDEN_FRAME;
    } else {
      b->ref_frame[0] = LAST_FRAME;
    }
    b->ref_frame[1] = NO_REF;
  }
}

/* motion vector candidates (dec_find_mv_refs, vp9_decodemv.c:469-612) */
typedef struct {
  int16_t mv[2][2]; /* [i] = { row, col } */
  int n, early, done;
} MvList;

static inline void cand_add(MvList *l, int row, int col) {
  if (l->done) return;
  if (l->n) {
    if (row != l->mv[0][0] || col != l->mv[0][1]) {
      l->mv[l->n][0] = (int16_t)row;
      l->mv[l->n][1] = (int16_t)col;
      ++l->n;
      l->done = 1;
    }
  } else {
    l->mv[0][0] = (int16_t)row;
    l->mv[0][1] = (int16_t)col;
    l->n = 1;
    l->done = l->early;
  }
}

static inline int pos_inside(const TileCtx *t, int row, int col) { /* is_inside, vp9_mvref_common.h:279 */
  return !(t->mi_row + row < 0 || t->mi_col + col < t->col_start || t->mi_row + row >= t->fe->mi_rows || t->mi_col + col >= t->col_end);
}

static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int find_mv_refs(TileCtx *t, int mode, int ref_frame, int block, int16_t out[2][2]) {
  vp9hip_fe *fe = t->fe;
  const int8_t(*pos)[2] = kMvRefPos[t->cur->sb_type];
  const int *bias = fe->ref_sign_bias;
  const MvRef *prev = fe->use_prev_mvs ? &fe->mvs[fe->mv_cur ^ 1][(size_t)t->mi_row * fe->mi_cols + t->mi_col] : NULL;
  MvList l;
  memset(&l, 0, sizeof(l));
  l.early = mode != NEARMV;
  int different_ref_found = 0, i = 0;
  if (block >= 0) {
    for (; i < 2; ++i) {
      if (!pos_inside(t, pos[i][0], pos[i][1])) continue;
      const vp9hip_block *c = cell_block(fe, t->mi_row + pos[i][0], t->mi_col + pos[i][1]);
      if (!c) continue;
      different_ref_found = 1;
      for (int w = 0; w < 2; ++w)
        if (c->ref_frame[w] == ref_frame) {
          if (c->sb_type < BLOCK_8X8) {
            const int sb = kIdxColToSub[block][pos[i][1] == 0];
            cand_add(&l, c->sub_mv[sb][w][0], c->sub_mv[sb][w][1]);
          } else {
            cand_add(&l, c->mv[w][0], c->mv[w][1]);
          }
          break;
        }
    }
  }
  for (; i < 8 && !l.done; ++i) {
    if (!pos_inside(t, pos[i][0], pos[i][1])) continue;
    const vp9hip_block *c = cell_block(fe, t->mi_row + pos[i][0], t->mi_col + pos[i][1]);
    if (!c) continue;
    different_ref_found = 1;
    if (c->ref_frame[0] == ref_frame)
      cand_add(&l, c->mv[0][0], c->mv[0][1]);
    else if (c->ref_frame[1] == ref_frame)
      cand_add(&l, c->mv[1][0], c->mv[1][1]);
  }
  if (prev && !l.done) {
    if (prev->ref[0] == ref_frame)
      cand_add(&l, prev->mv[0][0], prev->mv[0][1]);
    else if (prev->ref[1] == ref_frame)
      cand_add(&l, prev->mv[1][0], prev->mv[1][1]);
  }
  if (different_ref_found && !l.done) {
    for (i = 0; i < 8 && !l.done; ++i) {
      if (!pos_inside(t, pos[i][0], pos[i][1])) continue;
      const vp9hip_block *c = cell_block(fe, t->mi_row + pos[i][0], t->mi_col + pos[i][1]);
      if (!c || !is_inter(c)) continue;
      if (c->ref_frame[0] != ref_frame) {
        const int s = bias[c->ref_frame[0]] != bias[ref_frame] ? -1 : 1;
        cand_add(&l, s * c->mv[0][0], s * c->mv[0][1]);
      }
      if (has_second(c) && c->ref_frame[1] != ref_frame && (c->mv[1][0] != c->mv[0][0] || c->mv[1][1] != c->mv[0][1])) {
        const int s = bias[c->ref_frame[1]] != bias[ref_frame] ? -1 : 1;
        cand_add(&l, s * c->mv[1][0], s * c->mv[1][1]);
      }
    }
  }
  if (prev && !l.done) {
    if (prev->ref[0] != ref_frame && prev->ref[0] > INTRA_FRAME) {
      const int s = bias[prev->ref[0]] != bias[ref_frame] ? -1 : 1;
      cand_add(&l, s * prev->mv[0][0], s * prev->mv[0][1]);
    }
    if (prev->ref[1] > INTRA_FRAME && prev->ref[1] != ref_frame && (prev->mv[1][0] != prev->mv[0][0] || prev->mv[1][1] != prev->mv[0][1])) {
      const int s = bias[prev->ref[1]] != bias[ref_frame] ? -1 : 1;
      cand_add(&l, s * prev->mv[1][0], s * prev->mv[1][1]);
    }
  }
  const int count = l.done ? l.n : (mode == NEARMV ? 2 : 1);
  for (int k = 0; k < 2; ++k) {
    out[k][0] = l.mv[k][0];
    out[k][1] = l.mv[k][1];
    if (k < count) { /* clamp_mv_ref: 16 samples beyond the frame */
      out[k][1] = (int16_t)iclamp(out[k][1], t->to_left - 128, t->to_right + 128);
      out[k][0] = (int16_t)iclamp(out[k][0], t->to_top - 128, t->to_bottom + 128);
    }
  }
  return count;
}

static inline int mv_use_hp(const int16_t mv[2]) { return (abs(mv[0]) >> 3) < 8 && (abs(mv[1]) >> 3) < 8; }
static inline void lower_mv_precision(int16_t mv[2], int allow_hp) {
  if (allow_hp && mv_use_hp(mv)) return;
  if (mv[0] & 1) mv[0] += (mv[0] > 0 ? -1 : 1);
  if (mv[1] & 1) mv[1] += (mv[1] > 0 ? -1 : 1);
}

static int read_mv_component(TileCtx *t, int comp, int use_hp) { /* vp9_decodemv.c:237 */
  const ProbCtx *fc = &t->fe->fc;
  BoolDec *r = &t->bd;
  const int sign = bd_read(r, fc->mv_sign[comp]);
  const int cls = bd_tree(r, kMvClassTree, fc->mv_classes[comp]);
  int d, mag;
  if (cls == 0) {
    d = bd_read(r, fc->mv_class0[comp][0]);
    mag = 0;
  } else {
    d = 0;
    for (int i = 0; i < cls; ++i) d |= bd_read(r, fc->mv_bits[comp][i]) << i;
    mag = 2 << (cls + 2);
  }
  const int fr = bd_tree(r, kMvFpTree, cls == 0 ? fc->mv_class0_fp[comp][d] : fc->mv_fp[comp]);
  const int hp = use_hp ? bd_read(r, cls == 0 ? fc->mv_class0_hp[comp] : fc->mv_hp[comp]) : 1;
  mag += ((d << 3) | (fr << 1) | hp) + 1;
  return sign ? -mag : mag;
}

static void count_mv_component(Counts *c, int comp, int v) { /* inc_mv_component, vp9_entropymv.c:109 */
  const int s = v < 0;
  ++c->mv_sign[comp][s];
  const int z = (s ? -v : v) - 1;
  int cls = 10;
  if (z < 2 * 4096) {
    const int k = z >> 3;
    cls = k < 2 ? 0 : 31 - __builtin_clz((unsigned)k);
  }
  ++c->mv_classes[comp][cls];
  const int o = z - (cls ? 2 << (cls + 2) : 0);
  const int d = o >> 3, f = (o >> 1) & 3, e = o & 1;
  if (cls == 0) {
    ++c->mv_class0[comp][d];
    ++c->mv_class0_fp[comp][d][f];
    ++c->mv_class0_hp[comp][e];
  } else {
    for (int i = 0; i < cls; ++i) ++c->mv_bits[comp][i][(d >> i) & 1];
    ++c->mv_fp[comp][f];
    ++c->mv_hp[comp][e];
  }
}

static int read_mv(TileCtx *t, int16_t mv[2], const int16_t ref[2]) { /* vp9_decodemv.c:270; returns is_mv_valid */
  const int joint = bd_tree(&t->bd, kMvJointTree, t->fe->fc.mv_joints);
  const int use_hp = t->fe->h.allow_hp && mv_use_hp(ref);
  int dr = 0, dc = 0;
  if (joint == MV_JOINT_HZVNZ || joint == MV_JOINT_HNZVNZ) dr = read_mv_component(t, 0, use_hp);
  if (joint == MV_JOINT_HNZVZ || joint == MV_JOINT_HNZVNZ) dc = read_mv_component(t, 1, use_hp);
  if (t->counts) {
    ++t->counts->mv_joints[joint];
    if (dr) count_mv_component(t->counts, 0, dr);
    if (dc) count_mv_component(t->counts, 1, dc);
  }
  const int row = ref[0] + dr, col = ref[1] + dc;
  mv[0] = (int16_t)row;
  mv[1] = (int16_t)col;
  return row > MV_LOW && row < MV_UPP && col > MV_LOW && col < MV_UPP;
}

static int read_inter_mode(TileCtx *t, int ctx) {
  const int m = bd_tree(&t->bd, kInterModeTree, t->fe->fc.inter_mode[ctx]);
  if (t->counts) ++t->counts->inter_mode[ctx][m];
  return NEARESTMV + m;
}

/* assign_mv, vp9_decodemv.c:403: returns 0 on an invalid vector */
static int assign_mv(TileCtx *t, int mode, int16_t mv[2][2], int16_t ref_mv[2][2], int16_t near_nearest[2][2], int compound) {
  int ok = 1;
  switch (mode) {
    case NEWMV:
      for (int i = 0; i < 1 + compound; ++i) ok &= read_mv(t, mv[i], ref_mv[i]);
      break;
    case NEARMV:
    case NEARESTMV: memcpy(mv, near_nearest, sizeof(int16_t) * 4); break;
    case ZEROMV: memset(mv, 0, sizeof(int16_t) * 4); break;
    default: return 0;
  }
  return ok;
}

/* append_sub8x8_mvs_for_idx, vp9_decodemv.c:614 */
static void sub8x8_candidate(TileCtx *t, int b_mode, int block, int ref, int16_t best[2]) {
  vp9hip_block *b = t->cur;
  int16_t list[2][2];
#define SAME(a, c) ((a)[0] == (c)[0] && (a)[1] == (c)[1])
  switch (block) {
    case 0: {
      const int n = find_mv_refs(t, b_mode, b->ref_frame[ref], block, list);
      best[0] = list[n - 1][0];
      best[1] = list[n - 1][1];
      break;
    }
    case 1:
    case 2:
      if (b_mode == NEARESTMV) {
        best[0] = b->sub_mv[0][ref][0];
        best[1] = b->sub_mv[0][ref][1];
      } else {
        find_mv_refs(t, b_mode, b->ref_frame[ref], block, list);
        best[0] = best[1] = 0;
        for (int n = 0; n < 2; ++n)
          if (!SAME(b->sub_mv[0][ref], list[n])) {
            best[0] = list[n][0];
            best[1] = list[n][1];
            break;
          }
      }
      break;
    default:
      if (b_mode == NEARESTMV) {
        best[0] = b->sub_mv[2][ref][0];
        best[1] = b->sub_mv[2][ref][1];
      } else {
        best[0] = best[1] = 0;
        if (!SAME(b->sub_mv[2][ref], b->sub_mv[1][ref])) {
          best[0] = b->sub_mv[1][ref][0];
          best[1] = b->sub_mv[1][ref][1];
          break;
        }
        if (!SAME(b->sub_mv[2][ref], b->sub_mv[0][ref])) {
          best[0] = b->sub_mv[0][ref][0];
          best[1] = b->sub_mv[0][ref][1];
          break;
        }
        find_mv_refs(t, b_mode, b->ref_frame[ref], block, list);
        for (int n = 0; n < 2; ++n)
          if (!SAME(b->sub_mv[2][ref], list[n])) {
            best[0] = list[n][0];
            best[1] = list[n][1];
            break;
          }
      }
  }
#undef SAME
}

static void read_inter_block_mode_info(TileCtx *t) { /* vp9_decodemv.c:701 */
  vp9hip_fe *fe = t->fe;
  vp9hip_block *b = t->cur;
  const FrameHdr *h = &fe->h;
  const int bsize = b->sb_type;
  int16_t best_ref[2][2] = { { 0, 0 }, { 0, 0 } };
  read_ref_frames(t, SEG_ID(b));
  const int compound = has_second(b);
  for (int r = 0; r < 1 + compound; ++r)
    if (b->ref_frame[r] < LAST_FRAME || b->ref_frame[r] > ALTREF_FRAME || fe->h.ref_idx[b->ref_frame[r] - 1] < 0) {
      t->corrupt = 1;
      b->ref_frame[0] = LAST_FRAME;
      b->ref_frame[1] = NO_REF;
      return;
    }
  /* get_mode_context, vp9_decodemv.c:681 */
  int counter = 0;
  {
    const int8_t(*pos)[2] = kMvRefPos[bsize];
    for (int i = 0; i < 2; ++i)
      if (pos_inside(t, pos[i][0], pos[i][1])) {
        const vp9hip_block *c = cell_block(fe, t->mi_row + pos[i][0], t->mi_col + pos[i][1]);
        if (c) counter += kMode2Counter[c->mode];
      }
  }
  const int mode_ctx = kCounterToCtx[counter];
  if (seg_active(&fe->seg, SEG_ID(b), SEG_LVL_SKIP)) {
    b->mode = ZEROMV;
    if (bsize < BLOCK_8X8) {
      t->corrupt = 1;
      return;
    }
  } else if (bsize >= BLOCK_8X8) {
    b->mode = (uint8_t)read_inter_mode(t, mode_ctx);
  }
  if (h->interp_filter == SWITCHABLE) {
    const int ctx = ctx_interp(t);
    b->interp_filter = (uint8_t)bd_tree(&t->bd, kInterpTree, fe->fc.interp[ctx]);
    if (t->counts) ++t->counts->interp[ctx][b->interp_filter];
  } else {
    b->interp_filter = (uint8_t)h->interp_filter;
  }
  if (bsize < BLOCK_8X8) {
    const int n4w = bsize == BLOCK_4X4 || bsize == BLOCK_4X8 ? 1 : 2; /* sub-blocks are n4w x n4h 4x4 units */
    const int n4h = bsize == BLOCK_4X4 || bsize == BLOCK_8X4 ? 1 : 2;
    int b_mode = NEARESTMV, got_new = 0;
    int16_t best_sub[2][2] = { { 0, 0 }, { 0, 0 } };
    for (int idy = 0; idy < 2; idy += n4h)
      for (int idx = 0; idx < 2; idx += n4w) {
        const int j = idy * 2 + idx;
        b_mode = read_inter_mode(t, mode_ctx);
        if (b_mode == NEARESTMV || b_mode == NEARMV) {
          for (int r = 0; r < 1 + compound; ++r) sub8x8_candidate(t, b_mode, j, r, best_sub[r]);
        } else if (b_mode == NEWMV && !got_new) {
          for (int r = 0; r < 1 + compound; ++r) {
            int16_t list[2][2];
            find_mv_refs(t, NEWMV, b->ref_frame[r], -1, list);
            lower_mv_precision(list[0], h->allow_hp);
            best_ref[r][0] = list[0][0];
            best_ref[r][1] = list[0][1];
          }
          got_new = 1;
        }
        if (!assign_mv(t, b_mode, b->sub_mv[j], best_ref, best_sub, compound)) {
          t->corrupt = 1;
          idy = 2;
          break;
        }
        if (n4h == 2) memcpy(b->sub_mv[j + 2], b->sub_mv[j], sizeof(b->sub_mv[j]));
        if (n4w == 2) memcpy(b->sub_mv[j + 1], b->sub_mv[j], sizeof(b->sub_mv[j]));
      }
    b->mode = (uint8_t)b_mode;
    memcpy(b->mv, b->sub_mv[3], sizeof(b->mv));
  } else {
    if (b->mode != ZEROMV)
      for (int r = 0; r < 1 + compound; ++r) {
        int16_t list[2][2];
        const int n = find_mv_refs(t, b->mode, b->ref_frame[r], -1, list);
        lower_mv_precision(list[n - 1], h->allow_hp);
        best_ref[r][0] = list[n - 1][0];
        best_ref[r][1] = list[n - 1][1];
      }
    if (!assign_mv(t, b->mode, b->mv, best_ref, best_ref, compound)) t->corrupt = 1;
  }
}

static void read_inter_frame_mode_info(TileCtx *t, int x_mis, int y_mis) { /* vp9_decodemv.c:788 */
  vp9hip_fe *fe = t->fe;
  vp9hip_block *b = t->cur;
  SEG_ID(b) = (uint8_t)read_inter_segment_id(t, x_mis, y_mis);
  b->skip = (uint8_t)read_skip(t, SEG_ID(b));
  int inter;
  if (seg_active(&fe->seg, SEG_ID(b), SEG_LVL_REF_FRAME)) {
    inter = fe->seg.feature_data[SEG_ID(b)][SEG_LVL_REF_FRAME] != INTRA_FRAME;
  } else {
    const int ctx = ctx_intra_inter(t);
    inter = bd_read(&t->bd, fe->fc.intra_inter[ctx]);
    if (t->counts) ++t->counts->intra_inter[ctx][inter];
  }
  b->tx_size = (uint8_t)read_tx_size(t, !b->skip || !inter);
  if (inter)
    read_inter_block_mode_info(t);
  else
    read_intra_block_mode_info(t);
}

/* ---- coefficient tokens (decode_coefs / vp9_decode_block_tokens, vp9_detokenize.c:123-333) ------------- */
static int read_coefs(TileCtx *t, int type, int tx, const int16_t *dq, int ctx, const int16_t *scan, const int16_t *nb, int is_inter_blk) {
  vp9hip_fe *fe = t->fe;
  BoolDec local = t->bd, *r = &local; /* value / range / count in registers: the byte stores below would force reloads */
  const int max_eob = 16 << (tx << 1);
  const uint8_t(*probs)[6][3] = fe->fc.coef[tx][type][is_inter_blk];
  uint32_t(*cnt)[6][4] = t->counts ? t->counts->coef[tx][type][is_inter_blk] : NULL;
  uint32_t(*eobb)[6] = t->counts ? t->counts->eob_branch[tx][type][is_inter_blk] : NULL;
  const uint8_t *band_tr = tx == TX_4X4 ? kBand4x4 : kBand8x8Plus;
  const int dq_shift = tx == TX_32X32;
  const int bd = fe->bit_depth;
  const uint8_t *cat6 = bd == 12 ? kCat6ProbHigh12 : bd == 10 ? kCat6ProbHigh12 + 2 : kCat6Prob;
  const int cat6_bits = bd == 12 ? 18 : bd == 10 ? 16 : 14;
  uint8_t cache[32 * 32];
  int32_t *out = t->scratch;
  int c = 0, dqv = dq[0];
#define READ_BITS(pr, n, dst)                                         \
  do {                                                                \
    dst = 0;                                                          \
    for (int k_ = 0; k_ < (n); ++k_) dst = (dst << 1) | bd_read(r, (pr)[k_]); \
  } while (0)
  while (c < max_eob) {
    int band = band_tr[c];
    const uint8_t *p = probs[band][ctx];
    if (eobb) ++eobb[band][ctx];
    if (!bd_read(r, p[0])) {
      if (cnt) ++cnt[band][ctx][3];
      break;
    }
    while (!bd_read(r, p[1])) {
      if (cnt) ++cnt[band][ctx][0];
      dqv = dq[1];
      cache[scan[c]] = 0;
      ++c;
      if (c >= max_eob) { /* zeros up to the end: no end-of-block token */
        t->bd = local;
        return c;
      }
      ctx = (1 + cache[nb[2 * c]] + cache[nb[2 * c + 1]]) >> 1;
      band = band_tr[c];
      p = probs[band][ctx];
    }
    int64_t val;
    if (bd_read(r, p[2])) {
      const uint8_t *pp = kPareto8[p[2] - 1];
      if (cnt) ++cnt[band][ctx][2];
      if (bd_read(r, pp[0])) {
        int extra;
        if (bd_read(r, pp[3])) {
          cache[scan[c]] = 5;
          if (bd_read(r, pp[5])) {
            if (bd_read(r, pp[7])) {
              READ_BITS(cat6, cat6_bits, extra);
              val = 67 + extra;
            } else {
              READ_BITS(kCat5Prob, 5, extra);
              val = 35 + extra;
            }
          } else if (bd_read(r, pp[6])) {
            READ_BITS(kCat4Prob, 4, extra);
            val = 19 + extra;
          } else {
            READ_BITS(kCat3Prob, 3, extra);
            val = 11 + extra;
          }
        } else {
          cache[scan[c]] = 4;
          if (bd_read(r, pp[4])) {
            READ_BITS(kCat2Prob, 2, extra);
            val = 7 + extra;
          } else {
            READ_BITS(kCat1Prob, 1, extra);
            val = 5 + extra;
          }
        }
      } else if (bd_read(r, pp[1])) {
        cache[scan[c]] = 3;
        val = 3 + bd_read(r, pp[2]);
      } else {
        cache[scan[c]] = 2;
        val = 2;
      }
    } else {
      if (cnt) ++cnt[band][ctx][1];
      cache[scan[c]] = 1;
      val = 1;
    }
    const int v = (int)((val * dqv) >> dq_shift);
    out[scan[c]] = bd_bit(r) ? -v : v;
    ++c;
    ctx = (1 + cache[nb[2 * c]] + cache[nb[2 * c + 1]]) >> 1;
    dqv = dq[1];
  }
#undef READ_BITS
  t->bd = local;
  return c;
}

static const int16_t *scan_of(int tx, int tx_type, const int16_t **nb) {
  const int sel = kScanSel[tx][tx_type];
  switch (tx) {
    case TX_4X4: *nb = sel == 1 ? kNb4_1 : sel == 2 ? kNb4_2 : kNb4_0; return sel == 1 ? kScan4_1 : sel == 2 ? kScan4_2 : kScan4_0;
    case TX_8X8: *nb = sel == 1 ? kNb8_1 : sel == 2 ? kNb8_2 : kNb8_0; return sel == 1 ? kScan8_1 : sel == 2 ? kScan8_2 : kScan8_0;
    case TX_16X16: *nb = sel == 1 ? kNb16_1 : sel == 2 ? kNb16_2 : kNb16_0; return sel == 1 ? kScan16_1 : sel == 2 ? kScan16_2 : kScan16_0;
    default: *nb = kNb32_0; return kScan32_0;
  }
}

/* every transform block of a non-skip block, plane by plane (detoken_block, vp9_decodeframe.c:919-1024);
 * returns the sum of the eobs */
static int read_block_tokens(TileCtx *t) {
  vp9hip_fe *fe = t->fe;
  vp9hip_block *b = t->cur;
  const int inter = is_inter(b), sub8 = b->sb_type < BLOCK_8X8, lossless = fe->h.lossless;
  int eobtotal = 0;
  uint32_t checksum = 0; /* over eobs and coefficients, for the comparison with the reference's parse (tests/test_fe_blocks.py) */
  for (int p = 0; p < 3; ++p) {
    const int ss = p ? fe->ss_x : 0;
    const int tx = p ? kUvTxSize[b->sb_type][b->tx_size][fe->ss_x][fe->ss_y] : b->tx_size;
    const int n4w = (t->bw8 * 2) >> ss ? (t->bw8 * 2) >> ss : 1, n4h = (t->bh8 * 2) >> ss ? (t->bh8 * 2) >> ss : 1;
    const int mw = n4w + (t->to_right >= 0 ? 0 : t->to_right >> (5 + ss));
    const int mh = n4h + (t->to_bottom >= 0 ? 0 : t->to_bottom >> (5 + ss));
    const int lim_w = t->to_right >= 0 ? 0 : mw, lim_h = t->to_bottom >= 0 ? 0 : mh; /* xd->max_blocks_wide / high */
    const int step = 1 << tx, n = 4 << tx;
    uint8_t *above = fe->above_nz[p] + ((t->mi_col * 2) >> ss);
    uint8_t *left = t->left_nz[p] + (((t->mi_row & 7) * 2) >> ss);
    const int16_t *dq = p ? fe->dq_uv[SEG_ID(b)] : fe->dq_y[SEG_ID(b)];
    const int x0 = (t->mi_col * 8) >> ss, y0 = (t->mi_row * 8) >> ss;
    const int estride = ((fe->ctx_cols * 8) >> ss) >> 2;
    for (int row = 0; row < mh; row += step)
      for (int col = 0; col < mw; col += step) {
        int tx_type = 0;
        if (!inter && p == 0 && !lossless) {
          const int mode = sub8 ? b->sub_mode[(row << 1) + col] : b->mode;
          tx_type = kIntraModeTxType[mode];
        }
        const int16_t *nb;
        const int16_t *scan = scan_of(tx, tx_type, &nb);
        int ca = 0, cl = 0;
        for (int i = 0; i < step; ++i) {
          ca |= above[col + i];
          cl |= left[row + i];
        }
        const int eob = read_coefs(t, p > 0, tx, dq, (ca != 0) + (cl != 0), scan, nb, inter);
        /* contexts of positions beyond the frame edge read as empty (get_ctx_shift, vp9_detokenize.c:255) */
        for (int i = 0; i < step; ++i) {
          above[col + i] = (uint8_t)(eob > 0 && !(lim_w && col + i >= lim_w));
          left[row + i] = (uint8_t)(eob > 0 && !(lim_h && row + i >= lim_h));
        }
        fe->eob[p][(size_t)((y0 >> 2) + row) * estride + (x0 >> 2) + col] = eob;
        uint32_t sum = 0;
        if (eob > 0) {
          const int ext = vp9hip_coeff_extent(eob, (p || tx == TX_32X32) ? 0 : tx_type, tx);
          if (fe->checksums)
            for (int i = 0; i < ext; ++i) sum += (uint32_t)t->scratch[i] * (uint32_t)(i + 1);
          if (t->narrow) {
            /* int16 slots at the same offsets (in coefficients): the array is an int16 array for this frame */
            int16_t *d16 = (int16_t *)fe->coef[p] + (t->cf[p] - fe->coef[p]);
            uint32_t bad = 0;
            for (int i = 0; i < ext; ++i) {
              const int32_t v = t->scratch[i];
              d16[i] = (int16_t)v;
              bad |= (uint32_t)(v + 32768) >> 16;
              if (fe->narrow_limit) bad |= (uint32_t)((v < 0 ? -v : v) >= fe->narrow_limit);
            }
            t->out_of_range |= bad;
          } else {
            memcpy(t->cf[p], t->scratch, sizeof(int32_t) * (size_t)ext);
          }
          t->cf[p] += ext;
          /* the clearing rule of vp9_decodeframe.c:960-967 — what the extent is defined by */
          if (eob == 1)
            t->scratch[0] = 0;
          else
            memset(t->scratch, 0, sizeof(int32_t) * (size_t)(ext < n * n ? ext : n * n));
        }
        checksum = checksum * 1000003u + sum + (uint32_t)eob * 7919u;
        eobtotal += eob;
      }
  }
  memcpy(b->reserved2, &checksum, 4);
  return eobtotal;
}

/* ---- blocks and partitions (decode_block :1198, decode_partition :1386) -------------------------------- */
static void decode_block(TileCtx *t, int mi_row, int mi_col, int bsize) {
  vp9hip_fe *fe = t->fe;
  const FrameHdr *h = &fe->h;
  const int bw8 = kW8[bsize], bh8 = kH8[bsize];
  const int x_mis = bw8 < fe->mi_cols - mi_col ? bw8 : fe->mi_cols - mi_col;
  const int y_mis = bh8 < fe->mi_rows - mi_row ? bh8 : fe->mi_rows - mi_row;
  vp9hip_block *b = t->blk++;
  memset(b, 0, sizeof(*b));
  b->mi_row = (int16_t)mi_row;
  b->mi_col = (int16_t)mi_col;
  b->sb_type = (uint8_t)bsize;
  const int32_t index = (int32_t)(b - fe->seg_blocks);
  for (int k = 0; k < 3; ++k) t->off[index][k] = (uint32_t)(t->cf[k] - fe->coef[k]);
  for (int y = 0; y < y_mis; ++y)
    for (int x = 0; x < x_mis; ++x) fe->grid[(size_t)(mi_row + y) * fe->mi_cols + mi_col + x] = index;
  t->cur = b;
  t->mi_row = mi_row;
  t->mi_col = mi_col;
  t->bw8 = bw8;
  t->bh8 = bh8;
  t->to_top = -(mi_row * 64);
  t->to_bottom = (fe->mi_rows - bh8 - mi_row) * 64;
  t->to_left = -(mi_col * 64);
  t->to_right = (fe->mi_cols - bw8 - mi_col) * 64;
  t->above = mi_row > 0 ? cell_block(fe, mi_row - 1, mi_col) : NULL;
  t->left = mi_col > t->col_start ? cell_block(fe, mi_row, mi_col - 1) : NULL;
  const int intra_frame = h->frame_type == KEY_FRAME || h->intra_only;
  if (intra_frame) {
    read_intra_frame_mode_info(t, x_mis, y_mis);
  } else {
    read_inter_frame_mode_info(t, x_mis, y_mis);
    MvRef *mv = &fe->mvs[fe->mv_cur][(size_t)mi_row * fe->mi_cols + mi_col];
    MvRef v;
    v.ref[0] = b->ref_frame[0];
    v.ref[1] = b->ref_frame[1];
    memcpy(v.mv, b->mv, sizeof(v.mv));
    for (int y = 0; y < y_mis; ++y)
      for (int x = 0; x < x_mis; ++x) mv[(size_t)y * fe->mi_cols + x] = v;
  }
  b->reserved[2] = b->skip; /* the flag as parsed */
  if (b->skip) {
    for (int p = 0; p < 3; ++p) { /* dec_reset_skip_context :802 */
      const int ss = p ? fe->ss_x : 0;
      const int n4w = (bw8 * 2) >> ss ? (bw8 * 2) >> ss : 1, n4h = (bh8 * 2) >> ss ? (bh8 * 2) >> ss : 1;
      memset(fe->above_nz[p] + ((mi_col * 2) >> ss), 0, (size_t)n4w);
      memset(t->left_nz[p] + (((mi_row & 7) * 2) >> ss), 0, (size_t)n4h);
    }
  } else {
    const int eobtotal = read_block_tokens(t);
    /* stock libvpx (vp9_decodeframe.c of v1.9: "skip loopfilter"): an inter block of 8x8 or more without a coded
     * coefficient counts as skipped from here on — for the loop filter and for the contexts of later blocks */
    if (is_inter(b) && bsize >= BLOCK_8X8 && eobtotal == 0) b->skip = 1;
  }
  b->filter_level = h->filter_level ? fe->lvl[SEG_ID(b)][b->ref_frame[0]][kModeLf[b->mode]] : 0;
  if (bd_error(&t->bd)) t->corrupt = 1;
}

static void decode_partition(TileCtx *t, int mi_row, int mi_col, int bsl /* log2 of the size in 8x8 units */) {
  vp9hip_fe *fe = t->fe;
  if (mi_row >= fe->mi_rows || mi_col >= fe->mi_cols) return;
  static const uint8_t bsize_of[4] = { 3, 6, 9, 12 };
  const int bsize = bsize_of[bsl], num8 = 1 << bsl, hbs = num8 >> 1;
  const int has_rows = (mi_row + hbs) < fe->mi_rows, has_cols = (mi_col + hbs) < fe->mi_cols;
  uint8_t *ap = fe->above_part + mi_col, *lp = t->left_part + (mi_row & 7);
  const int ctx = (((*lp >> bsl) & 1) * 2 + ((*ap >> bsl) & 1)) + bsl * 4;
  const int intra_frame = fe->h.frame_type == KEY_FRAME || fe->h.intra_only;
  const uint8_t *probs = intra_frame ? kKfPartition[ctx] : fe->fc.partition[ctx];
  int part;
  if (has_rows && has_cols)
    part = bd_tree(&t->bd, kPartitionTree, probs);
  else if (!has_rows && has_cols)
    part = bd_read(&t->bd, probs[1]) ? PARTITION_SPLIT : PARTITION_HORZ;
  else if (has_rows && !has_cols)
    part = bd_read(&t->bd, probs[2]) ? PARTITION_SPLIT : PARTITION_VERT;
  else
    part = PARTITION_SPLIT;
  if (t->counts) ++t->counts->partition[ctx][part];
  const int subsize = kSubsize[part][bsize];
  if (subsize >= BLOCK_INVALID) {
    t->corrupt = 1;
    return;
  }
  if (!hbs) {
    decode_block(t, mi_row, mi_col, subsize);
  } else {
    switch (part) {
      case PARTITION_NONE: decode_block(t, mi_row, mi_col, subsize); break;
      case PARTITION_HORZ:
        decode_block(t, mi_row, mi_col, subsize);
        if (has_rows) decode_block(t, mi_row + hbs, mi_col, subsize);
        break;
      case PARTITION_VERT:
        decode_block(t, mi_row, mi_col, subsize);
        if (has_cols) decode_block(t, mi_row, mi_col + hbs, subsize);
        break;
      default:
        decode_partition(t, mi_row, mi_col, bsl - 1);
        decode_partition(t, mi_row, mi_col + hbs, bsl - 1);
        decode_partition(t, mi_row + hbs, mi_col, bsl - 1);
        decode_partition(t, mi_row + hbs, mi_col + hbs, bsl - 1);
    }
  }
  if (bsl == 0 || part != PARTITION_SPLIT) {
    memset(ap, kPartCtxAbove[subsize], (size_t)num8);
    memset(lp, kPartCtxLeft[subsize], (size_t)num8);
  }
}

static int tile_offset(int idx, int mis, int log2) { /* get_tile_offset, vp9_tile_common.c:18 */
  const int sb = (mis + 7) >> 3;
  const int off = ((idx * sb) >> log2) << 3;
  return off < mis ? off : mis;
}

/* one tile column, its tile rows top to bottom (they share the above context) */
static double fe_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void run_tile_job(vp9hip_fe *fe, TileJob *job) {
  TileCtx *t = &job->tc;
  const double t_start = fe->trace ? fe_now() : 0.0;
  const int tile_rows = 1 << fe->h.log2_tile_rows;
  for (int tr = 0; tr < tile_rows && !t->corrupt; ++tr) {
    if (bd_init(&t->bd, job->buf[tr].data, job->buf[tr].size)) {
      t->corrupt = 1;
      break;
    }
    const int row_start = tile_offset(tr, fe->mi_rows, fe->h.log2_tile_rows);
    const int row_end = tile_offset(tr + 1, fe->mi_rows, fe->h.log2_tile_rows);
    for (int mi_row = row_start; mi_row < row_end && !t->corrupt; mi_row += 8) {
      memset(t->left_nz, 0, sizeof(t->left_nz));
      memset(t->left_part, 0, sizeof(t->left_part));
      for (int mi_col = t->col_start; mi_col < t->col_end; mi_col += 8) {
        vp9hip_block *before = t->blk;
        decode_partition(t, mi_row, mi_col, 3);
        fe->sb_count[(mi_row >> 3) * fe->sb_cols + (mi_col >> 3)] = (int32_t)(t->blk - before);
      }
    }
  }
  job->n_blocks = (int)(t->blk - t->seg_first);
  for (int p = 0; p < 3; ++p) job->cf_used[p] = t->cf[p] - t->cf_base[p];
  if (fe->trace) job->seconds = fe_now() - t_start;
}

/* ---- thread pool: tile columns of a frame in parallel -------------------------------------------------- */
static void *pool_main(void *arg) {
  vp9hip_fe *fe = (vp9hip_fe *)arg;
  int seen = 0;
  pthread_mutex_lock(&fe->mu);
  for (;;) {
    while (!fe->pool_stop && (fe->epoch == seen || fe->next_job >= fe->n_jobs)) {
      if (fe->epoch != seen && fe->next_job >= fe->n_jobs) seen = fe->epoch;
      pthread_cond_wait(&fe->cv_work, &fe->mu);
    }
    if (fe->pool_stop) break;
    const int j = fe->next_job++;
    pthread_mutex_unlock(&fe->mu);
    run_tile_job(fe, &fe->jobs[j]);
    pthread_mutex_lock(&fe->mu);
    if (++fe->done_jobs == fe->n_jobs) pthread_cond_signal(&fe->cv_done);
  }
  pthread_mutex_unlock(&fe->mu);
  return NULL;
}

static void run_jobs(vp9hip_fe *fe, int n_jobs) {
  int want = fe->max_threads > 0 ? fe->max_threads : n_jobs;
  if (want > n_jobs) want = n_jobs;
  if (want > 16) want = 16;
  if (want <= 1) {
    for (int j = 0; j < n_jobs; ++j) run_tile_job(fe, &fe->jobs[j]);
    return;
  }
  pthread_mutex_lock(&fe->mu);
  while (fe->n_thr < want - 1) { /* the calling thread takes jobs as well */
    if (pthread_create(&fe->thr[fe->n_thr], NULL, pool_main, fe)) break;
    ++fe->n_thr;
  }
  fe->n_jobs = n_jobs;
  fe->next_job = 0;
  fe->done_jobs = 0;
  ++fe->epoch;
  pthread_cond_broadcast(&fe->cv_work);
  for (;;) {
    if (fe->next_job < fe->n_jobs) {
      const int j = fe->next_job++;
      pthread_mutex_unlock(&fe->mu);
      run_tile_job(fe, &fe->jobs[j]);
      pthread_mutex_lock(&fe->mu);
      ++fe->done_jobs;
    } else if (fe->done_jobs < fe->n_jobs) {
      pthread_cond_wait(&fe->cv_done, &fe->mu);
    } else {
      break;
    }
  }
  pthread_mutex_unlock(&fe->mu);
}

/* ---- backward adaptation (vpx_dsp/prob.h:47-90, prob.c:17-47) ------------------------------------------- */
static inline uint8_t binary_prob(uint32_t n0, uint32_t den) {
  const int p = (int)(((uint64_t)n0 * 256 + (den >> 1)) / den);
  return (uint8_t)(p > 255 ? 255 : (p < 1 ? 1 : p));
}
static inline uint8_t weighted(int p1, int p2, int factor) { return (uint8_t)((p1 * (256 - factor) + p2 * factor + 128) >> 8); }
static uint8_t merge_coef(uint8_t pre, uint32_t c0, uint32_t c1, uint32_t sat, uint32_t max_factor) {
  const uint32_t den = c0 + c1;
  const uint8_t prob = den ? binary_prob(c0, den) : 128;
  const uint32_t count = den < sat ? den : sat;
  return weighted(pre, prob, (int)(max_factor * count / sat));
}
static uint8_t merge_mode(uint8_t pre, uint32_t c0, uint32_t c1) {
  static const uint8_t factor[21] = { 0, 6, 12, 19, 25, 32, 38, 44, 51, 57, 64, 70, 76, 83, 89, 96, 102, 108, 115, 121, 128 };
  const uint32_t den = c0 + c1;
  if (!den) return pre;
  return weighted(pre, binary_prob(c0, den), factor[den < 20 ? den : 20]);
}
static uint32_t merge_tree_at(int i, const int8_t *tree, const uint8_t *pre, const uint32_t *counts, uint8_t *probs) {
  const int l = tree[i], r = tree[i + 1];
  const uint32_t lc = l <= 0 ? counts[-l] : merge_tree_at(l, tree, pre, counts, probs);
  const uint32_t rc = r <= 0 ? counts[-r] : merge_tree_at(r, tree, pre, counts, probs);
  probs[i >> 1] = merge_mode(pre[i >> 1], lc, rc);
  return lc + rc;
}
static void merge_tree(const int8_t *tree, const uint8_t *pre, const uint32_t *counts, uint8_t *probs) {
  merge_tree_at(0, tree, pre, counts, probs);
}

static void adapt_probs(vp9hip_fe *fe) {
  const FrameHdr *h = &fe->h;
  const ProbCtx *pre = &fe->saved[h->frame_context_idx];
  ProbCtx *fc = &fe->fc;
  const Counts *c = &fe->counts;
  const int intra_frame = h->frame_type == KEY_FRAME || h->intra_only;
  /* vp9_adapt_coef_probs, vp9_entropy.c:1082 */
  const uint32_t factor = intra_frame ? 112 : (fe->last_frame_type == KEY_FRAME ? 128 : 112), sat = 24;
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 6; ++k)
          for (int l = 0; l < (k == 0 ? 3 : 6); ++l) {
            const uint32_t *n = c->coef[t][i][j][k][l];
            const uint32_t neob = n[3], eb = c->eob_branch[t][i][j][k][l];
            fc->coef[t][i][j][k][l][0] = merge_coef(pre->coef[t][i][j][k][l][0], neob, eb - neob, sat, factor);
            fc->coef[t][i][j][k][l][1] = merge_coef(pre->coef[t][i][j][k][l][1], n[0], n[1] + n[2], sat, factor);
            fc->coef[t][i][j][k][l][2] = merge_coef(pre->coef[t][i][j][k][l][2], n[1], n[2], sat, factor);
          }
  if (intra_frame) return;
  /* vp9_adapt_mode_probs, vp9_entropymode.c:340 */
  for (int i = 0; i < 4; ++i) fc->intra_inter[i] = merge_mode(pre->intra_inter[i], c->intra_inter[i][0], c->intra_inter[i][1]);
  for (int i = 0; i < 5; ++i) fc->comp_inter[i] = merge_mode(pre->comp_inter[i], c->comp_inter[i][0], c->comp_inter[i][1]);
  for (int i = 0; i < 5; ++i) fc->comp_ref[i] = merge_mode(pre->comp_ref[i], c->comp_ref[i][0], c->comp_ref[i][1]);
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 2; ++j) fc->single_ref[i][j] = merge_mode(pre->single_ref[i][j], c->single_ref[i][j][0], c->single_ref[i][j][1]);
  for (int i = 0; i < 7; ++i) merge_tree(kInterModeTree, pre->inter_mode[i], c->inter_mode[i], fc->inter_mode[i]);
  for (int i = 0; i < 4; ++i) merge_tree(kIntraModeTree, pre->y_mode[i], c->y_mode[i], fc->y_mode[i]);
  for (int i = 0; i < 10; ++i) merge_tree(kIntraModeTree, pre->uv_mode[i], c->uv_mode[i], fc->uv_mode[i]);
  for (int i = 0; i < 16; ++i) merge_tree(kPartitionTree, pre->partition[i], c->partition[i], fc->partition[i]);
  if (h->interp_filter == SWITCHABLE)
    for (int i = 0; i < 4; ++i) merge_tree(kInterpTree, pre->interp[i], c->interp[i], fc->interp[i]);
  if (h->tx_mode == TX_MODE_SELECT)
    for (int i = 0; i < 2; ++i) {
      const uint32_t *a = c->tx8[i], *b = c->tx16[i], *d = c->tx32[i];
      fc->tx8[i][0] = merge_mode(pre->tx8[i][0], a[0], a[1]);
      fc->tx16[i][0] = merge_mode(pre->tx16[i][0], b[0], b[1] + b[2]);
      fc->tx16[i][1] = merge_mode(pre->tx16[i][1], b[1], b[2]);
      fc->tx32[i][0] = merge_mode(pre->tx32[i][0], d[0], d[1] + d[2] + d[3]);
      fc->tx32[i][1] = merge_mode(pre->tx32[i][1], d[1], d[2] + d[3]);
      fc->tx32[i][2] = merge_mode(pre->tx32[i][2], d[2], d[3]);
    }
  for (int i = 0; i < 3; ++i) fc->skip[i] = merge_mode(pre->skip[i], c->skip[i][0], c->skip[i][1]);
  /* vp9_adapt_mv_probs, vp9_entropymv.c:162 */
  merge_tree(kMvJointTree, pre->mv_joints, c->mv_joints, fc->mv_joints);
  for (int i = 0; i < 2; ++i) {
    fc->mv_sign[i] = merge_mode(pre->mv_sign[i], c->mv_sign[i][0], c->mv_sign[i][1]);
    merge_tree(kMvClassTree, pre->mv_classes[i], c->mv_classes[i], fc->mv_classes[i]);
    merge_tree(kMvClass0Tree, pre->mv_class0[i], c->mv_class0[i], fc->mv_class0[i]);
    for (int j = 0; j < 10; ++j) fc->mv_bits[i][j] = merge_mode(pre->mv_bits[i][j], c->mv_bits[i][j][0], c->mv_bits[i][j][1]);
    for (int j = 0; j < 2; ++j) merge_tree(kMvFpTree, pre->mv_class0_fp[i][j], c->mv_class0_fp[i][j], fc->mv_class0_fp[i][j]);
    merge_tree(kMvFpTree, pre->mv_fp[i], c->mv_fp[i], fc->mv_fp[i]);
    if (h->allow_hp) {
      fc->mv_class0_hp[i] = merge_mode(pre->mv_class0_hp[i], c->mv_class0_hp[i][0], c->mv_class0_hp[i][1]);
      fc->mv_hp[i] = merge_mode(pre->mv_hp[i], c->mv_hp[i][0], c->mv_hp[i][1]);
    }
  }
}

static void add_counts(Counts *dst, const Counts *src) {
  uint32_t *d = (uint32_t *)dst;
  const uint32_t *s = (const uint32_t *)src;
  for (size_t i = 0; i < sizeof(Counts) / sizeof(uint32_t); ++i) d[i] += s[i];
}

/* ---- public entry points ------------------------------------------------------------------------------- */
int vp9hip_fe_create(vp9hip_fe **out, vp9hip_alloc_fn alloc, vp9hip_free_fn release, void *user, int threads) {
  if (!out || (alloc == NULL) != (release == NULL)) return VP9HIP_EINVAL;
  vp9hip_fe *fe = (vp9hip_fe *)calloc(1, sizeof(*fe));
  if (!fe) return VP9HIP_ENOMEM;
  fe->alloc = alloc;
  fe->release = release;
  fe->user = user;
  fe->max_threads = threads;
  fe->need_resync = 1;
  fe->checksums = getenv("VP9HIP_FE_CHECKSUMS") != NULL;
  fe->trace = getenv("VP9HIP_FE_TRACE") != NULL;
  for (int i = 0; i < 8; ++i) fe->ref_map[i] = -1;
  fe->prev_new_slot = -1;
  pthread_mutex_init(&fe->mu, NULL);
  pthread_cond_init(&fe->cv_work, NULL);
  pthread_cond_init(&fe->cv_done, NULL);
  *out = fe;
  return VP9HIP_OK;
}

static void coef_free_set(vp9hip_fe *fe, int k, int p) {
  struct OutSet *s = &fe->sets[k];
  if (s->coef[p]) {
    if (fe->release)
      fe->release(fe->user, s->coef[p]);
    else
      free(s->coef[p]);
  }
  s->coef[p] = NULL;
  s->coef_cap[p] = 0;
}

static void coef_free(vp9hip_fe *fe, int p) {
  for (int k = 0; k < 3; ++k) coef_free_set(fe, k, p);
  fe->coef[p] = NULL;
}

static void use_set(vp9hip_fe *fe, int k) {
  const struct OutSet *s = &fe->sets[k];
  fe->set_idx = k;
  fe->seg_blocks = s->seg_blocks;
  fe->out_blocks = s->out_blocks;
  fe->seg_off = s->seg_off;
  fe->out_off = s->out_off;
  for (int p = 0; p < 3; ++p) {
    fe->eob[p] = s->eob[p];
    fe->coef[p] = s->coef[p];
  }
}

void vp9hip_fe_set_narrow_slots(vp9hip_fe *fe, int on) {
  if (!fe) return;
  fe->narrow_slots = on != 0;
  fe->narrow_limit = on > 1 ? on : 0; /* (tests: the fall-back to int32 slots on ordinary streams) */
}
int vp9hip_fe_wide_frames(const vp9hip_fe *fe) { return fe ? fe->wide_frames : 0; }

void vp9hip_fe_destroy(vp9hip_fe *fe) {
  if (!fe) return;
  if (fe->trace && fe->tr_frames)
    fprintf(stderr,
            "vp9hip_fe: %d frames; per frame %.3f ms: headers + set-up %.3f, tile columns %.3f (longest column %.3f, all columns "
            "together %.3f), merge + adaptation + state %.3f\n",
            fe->tr_frames, 1e3 * fe->tr_total / fe->tr_frames, 1e3 * fe->tr_head / fe->tr_frames, 1e3 * fe->tr_tiles / fe->tr_frames,
            1e3 * fe->tr_tile_max / fe->tr_frames, 1e3 * fe->tr_tile_sum / fe->tr_frames, 1e3 * fe->tr_tail / fe->tr_frames);
  pthread_mutex_lock(&fe->mu);
  fe->pool_stop = 1;
  pthread_cond_broadcast(&fe->cv_work);
  pthread_mutex_unlock(&fe->mu);
  for (int i = 0; i < fe->n_thr; ++i) pthread_join(fe->thr[i], NULL);
  pthread_mutex_destroy(&fe->mu);
  pthread_cond_destroy(&fe->cv_work);
  pthread_cond_destroy(&fe->cv_done);
  for (int p = 0; p < 3; ++p) {
    coef_free(fe, p);
    for (int k = 0; k < 3; ++k) free(fe->sets[k].eob[p]);
    free(fe->above_nz[p]);
  }
  for (int k = 0; k < 3; ++k) {
    free(fe->sets[k].seg_blocks);
    free(fe->sets[k].seg_off);
    free(fe->sets[k].out_blocks);
    free(fe->sets[k].out_off);
  }
  free(fe->above_part);
  free(fe->grid);
  free(fe->sb_count);
  free(fe->seg_map[0]);
  free(fe->seg_map[1]);
  free(fe->mvs[0]);
  free(fe->mvs[1]);
  free(fe->jobs);
  free(fe);
}

int vp9hip_fe_split_superframe(const uint8_t *data, size_t size, uint32_t sizes[8]) {
  if (!data || !size) return 0;
  sizes[0] = (uint32_t)size;
  const uint8_t marker = data[size - 1];
  if ((marker & 0xe0) != 0xc0) return 1;
  const int frames = (marker & 7) + 1, mag = ((marker >> 3) & 3) + 1;
  const size_t index_sz = 2 + (size_t)mag * frames;
  if (size < index_sz || data[size - index_sz] != marker) return 1;
  const uint8_t *x = data + size - index_sz + 1;
  size_t total = 0;
  for (int i = 0; i < frames; ++i) {
    uint32_t s = 0;
    for (int j = 0; j < mag; ++j) s |= (uint32_t)(*x++) << (j * 8);
    sizes[i] = s;
    total += s;
  }
  if (total + index_sz > size) {
    sizes[0] = (uint32_t)size;
    return 1;
  }
  return frames;
}

static int ensure_frame_arrays(vp9hip_fe *fe) {
  const int ctx_cols = (fe->mi_cols + 7) & ~7, ctx_rows = (fe->mi_rows + 7) & ~7;
  const size_t cells = (size_t)fe->mi_rows * fe->mi_cols;
  if (ctx_cols != fe->ctx_cols || !fe->above_part) {
    for (int p = 0; p < 3; ++p) {
      free(fe->above_nz[p]);
      fe->above_nz[p] = (uint8_t *)malloc((size_t)ctx_cols * 2 + 32);
      if (!fe->above_nz[p]) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    }
    free(fe->above_part);
    fe->above_part = (uint8_t *)malloc((size_t)ctx_cols + 16);
    if (!fe->above_part) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    fe->ctx_cols = ctx_cols;
  }
  /* The three output sets rotate: the one written now was handed out three frames ago, and vp9hip_fe.h promises a
   * frame's arrays for that call and the next two — the caller may still be packing the frame before this one and
   * the device still fetching the coefficients of the one before that.  So only THIS set may grow: a frame larger
   * than any before it (a key frame, an intra-only frame or a reference of another size in mid-stream) leaves the
   * other two sets, and whatever still reads them, alone; each set catches up when its own turn comes. */
  struct OutSet *s = &fe->sets[(fe->set_idx + 1) % 3];
  if (cells > s->cells_cap) {
    free(s->seg_blocks);
    free(s->seg_off);
    free(s->out_blocks);
    free(s->out_off);
    s->seg_blocks = (vp9hip_block *)malloc(sizeof(vp9hip_block) * (cells + 1));
    s->seg_off = (uint32_t(*)[3])malloc(sizeof(uint32_t) * 3 * (cells + 1));
    s->out_blocks = (vp9hip_block *)malloc(sizeof(vp9hip_block) * (cells + 1));
    s->out_off = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (cells + 1));
    s->cells_cap = (s->seg_blocks && s->seg_off && s->out_blocks && s->out_off) ? cells : 0;
    if (!s->cells_cap) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
  }
  /* scratch of the parse itself (not handed out) */
  if (cells > fe->cells_cap) {
    free(fe->grid);
    fe->grid = (int32_t *)malloc(sizeof(int32_t) * (cells + 1));
    fe->cells_cap = fe->grid ? cells : 0;
    if (!fe->cells_cap) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
  }
  if ((size_t)(ctx_cols >> 3) * (ctx_rows >> 3) > fe->sb_cap || !fe->sb_count) {
    free(fe->sb_count);
    fe->sb_cap = 0;
    fe->sb_count = (int32_t *)malloc(sizeof(int32_t) * ((size_t)(ctx_cols >> 3) * (ctx_rows >> 3) + 1));
    if (!fe->sb_count) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    fe->sb_cap = (size_t)(ctx_cols >> 3) * (ctx_rows >> 3);
  }
  for (int p = 0; p < 3; ++p) {
    const int ss = p ? fe->ss_x : 0;
    const size_t pw = (size_t)(ctx_cols * 8) >> ss, ph = (size_t)(ctx_rows * 8) >> ss;
    const size_t want_e = (pw >> 2) * (ph >> 2), want_c = pw * ph + 64;
    if (want_e > s->eob_cap[p]) {
      free(s->eob[p]);
      s->eob[p] = (int32_t *)malloc(sizeof(int32_t) * want_e);
      s->eob_cap[p] = s->eob[p] ? want_e : 0;
      if (!s->eob[p]) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    }
    if (want_c > s->coef_cap[p]) {
      coef_free_set(fe, (int)(s - fe->sets), p);
      s->coef[p] = (int32_t *)(fe->alloc ? fe->alloc(fe->user, sizeof(int32_t) * want_c) : malloc(sizeof(int32_t) * want_c));
      s->coef_cap[p] = s->coef[p] ? want_c : 0;
      if (!s->coef[p]) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
    }
  }
  use_set(fe, (fe->set_idx + 1) % 3);
  return VP9HIP_OK;
}

int vp9hip_fe_parse(vp9hip_fe *fe, const uint8_t *data, size_t size, vp9hip_fe_frame *out) {
  if (!fe || !out) return VP9HIP_EINVAL;
  if (!data || !size) FE_FAIL(fe, "empty frame");
  memset(out, 0, sizeof(*out));
  const double tr0 = fe->trace ? fe_now() : 0.0;
  BitRd rb = { data, size * 8, 0, 0 };
  int rc = read_uncompressed_header(fe, &rb, out);
  if (rc) return rc;
  FrameHdr *h = &fe->h;
  if (h->show_existing) {
    out->show_existing = 1;
    out->show_slot = h->frame_to_show;
    out->show_frame = 1;
    const SlotInfo *s = &fe->slot[h->frame_to_show];
    out->params.width = s->width;
    out->params.height = s->height;
    out->params.ss_x = s->ss_x;
    out->params.ss_y = s->ss_y;
    out->params.bit_depth = s->bit_depth;
    out->params.hbd = s->bit_depth > 8;
    /* vp9_receive_compressed_data tail: sizes and the frame counter move on, the rest of the state stays */
    fe->last_width = fe->width;
    fe->last_height = fe->height;
    return VP9HIP_OK;
  }
  const int intra_frame = h->frame_type == KEY_FRAME || h->intra_only;
  if (h->header_bytes + h->first_partition_size > size) FE_FAIL(fe, "truncated packet or corrupt header length");
  if ((rc = ensure_frame_arrays(fe))) return rc;

  /* vp9_decode_frame :3507 */
  fe->use_prev_mvs = !h->error_res && fe->width == fe->last_width && fe->height == fe->last_height && !fe->last_intra_only &&
                     fe->last_show_frame && fe->last_frame_type != KEY_FRAME && fe->have_frame && fe->mvs[fe->mv_cur ^ 1] &&
                     /* (implied by the size test above; the arrays themselves only ever grow, so their allocated size
                      * says nothing — it kept the previous frame's vectors out after a switch to a smaller size) */
                     fe->mv_wr_rows[fe->mv_cur ^ 1] == fe->mi_rows && fe->mv_wr_cols[fe->mv_cur ^ 1] == fe->mi_cols;
  fe->fc = fe->saved[h->frame_context_idx];
  if (!fe->fc.initialized) FE_FAIL(fe, "uninitialized entropy context");
  if ((rc = read_compressed_header(fe, data + h->header_bytes, h->first_partition_size))) return rc;

  /* loop-filter levels per (segment, reference, mode class) and thresholds; dequantisers per segment */
  int32_t seg_lf_on[8], seg_lf[8];
  for (int s = 0; s < 8; ++s) {
    seg_lf_on[s] = seg_active(&fe->seg, s, SEG_LVL_ALT_LF);
    seg_lf[s] = fe->seg.feature_data[s][SEG_LVL_ALT_LF];
  }
  if (h->filter_level)
    vp9hip_lf_frame_init(h->filter_level, h->sharpness, seg_lf_on, seg_lf, fe->seg.abs_delta, h->mode_ref_delta_enabled, fe->lf_ref_deltas,
                         fe->lf_mode_deltas, fe->lvl, &out->lf_thresh);
  {
    const int b = fe->bit_depth == 8 ? 0 : fe->bit_depth == 10 ? 1 : 2;
    for (int s = 0; s < 8; ++s) { /* setup_segmentation_dequant :1655, vp9_get_qindex */
      int q = h->base_qindex;
      if (seg_active(&fe->seg, s, SEG_LVL_ALT_Q)) {
        const int d = fe->seg.feature_data[s][SEG_LVL_ALT_Q];
        q = iclamp(fe->seg.abs_delta ? d : q + d, 0, 255);
      }
      fe->dq_y[s][0] = kDcQ[b][iclamp(q + h->y_dc_delta, 0, 255)];
      fe->dq_y[s][1] = kAcQ[b][iclamp(q, 0, 255)];
      fe->dq_uv[s][0] = kDcQ[b][iclamp(q + h->uv_dc_delta, 0, 255)];
      fe->dq_uv[s][1] = kAcQ[b][iclamp(q + h->uv_ac_delta, 0, 255)];
    }
  }

  /* tiles: sizes in front of all but the last (get_tile_buffers :1910) */
  const int tile_cols = 1 << h->log2_tile_cols, tile_rows = 1 << h->log2_tile_rows;
  TileJob *jobs = (TileJob *)realloc(fe->jobs, sizeof(TileJob) * (size_t)tile_cols);
  if (!jobs) return fe_fail(fe, VP9HIP_ENOMEM, "out of memory");
  fe->jobs = jobs;
  memset(jobs, 0, sizeof(TileJob) * (size_t)tile_cols);
  {
    const uint8_t *p = data + h->header_bytes + h->first_partition_size, *end = data + size;
    for (int r = 0; r < tile_rows; ++r)
      for (int c = 0; c < tile_cols; ++c) {
        size_t sz;
        if (r == tile_rows - 1 && c == tile_cols - 1) {
          sz = (size_t)(end - p);
        } else {
          if (end - p < 4) FE_FAIL(fe, "truncated packet or corrupt tile length");
          sz = ((size_t)p[0] << 24) | ((size_t)p[1] << 16) | ((size_t)p[2] << 8) | p[3];
          p += 4;
          if (sz > (size_t)(end - p)) FE_FAIL(fe, "truncated packet or corrupt tile size");
        }
        jobs[c].buf[r].data = p;
        jobs[c].buf[r].size = sz;
        p += sz;
      }
  }
  const int collect = !h->frame_parallel;
  const int ctx_rows = (fe->mi_rows + 7) & ~7;
  double tr1 = 0.0, tr2 = 0.0;
  int total = 0;
  /* Coefficient slots: int16 first when the caller asked for them (half the bytes on the way to the device).  The
   * reference keeps 32-bit coefficients (vpx_dsp/vpx_dsp_common.h:36-37; vp9_detokenize.c:243-250 stores whatever
   * (value * dequantiser) / 2 gives), so a frame in which a single coefficient does not fit is parsed AGAIN with
   * int32 slots: the tile jobs only write per-frame state (their lists, the segment map and motion vectors of THIS
   * frame, their counts), so running them twice gives the same result as running them once. */
  for (int narrow = fe->narrow_slots; ; narrow = 0) {
    /* per-frame context reset (decode_tiles :2358-2362) */
    for (int p = 0; p < 3; ++p) memset(fe->above_nz[p], 0, (size_t)fe->ctx_cols * 2 + 32);
    memset(fe->above_part, 0, (size_t)fe->ctx_cols + 16);
    memset(fe->grid, 0xff, sizeof(int32_t) * (size_t)fe->mi_rows * fe->mi_cols);
    for (int c = 0; c < tile_cols; ++c) {
      TileCtx *t = &jobs[c].tc;
      memset(t, 0, sizeof(*t));
      memset(&jobs[c].counts, 0, sizeof(jobs[c].counts));
      t->fe = fe;
      t->narrow = narrow;
      t->tile_col = c;
      t->col_start = tile_offset(c, fe->mi_cols, h->log2_tile_cols);
      t->col_end = tile_offset(c + 1, fe->mi_cols, h->log2_tile_cols);
      t->counts = collect ? &jobs[c].counts : NULL;
      t->seg_first = t->blk = fe->seg_blocks + (size_t)t->col_start * fe->mi_rows;
      t->off = fe->seg_off;
      for (int p = 0; p < 3; ++p) {
        const int ss = p ? fe->ss_x : 0;
        jobs[c].cf_start[p] = (int64_t)((t->col_start * 8) >> ss) * ((ctx_rows * 8) >> ss);
        t->cf[p] = t->cf_base[p] = fe->coef[p] + jobs[c].cf_start[p];
      }
    }
    tr1 = fe->trace ? fe_now() : 0.0;
    run_jobs(fe, tile_cols);
    tr2 = fe->trace ? fe_now() : 0.0;
    int corrupt = 0;
    uint32_t out_of_range = 0;
    total = 0;
    for (int c = 0; c < tile_cols; ++c) {
      corrupt |= jobs[c].tc.corrupt;
      out_of_range |= jobs[c].tc.out_of_range;
      total += jobs[c].n_blocks;
    }
    if (corrupt) {
      fe->need_resync = 1;
      FE_FAIL(fe, "decode failed: frame data is corrupted");
    }
    out->layout.narrow = narrow;
    if (!narrow || !out_of_range) break;
    ++fe->wide_frames;
  }

  /* one list in decode order: superblock raster order (= the serial loop of decode_tiles :2388-2430) */
  const vp9hip_block *blocks = fe->seg_blocks;
  const uint32_t *offs = &fe->seg_off[0][0];
  if (tile_cols > 1) {
    int cur[MAX_TILE_COLS], tcol[2048];
    for (int c = 0; c < tile_cols; ++c) {
      cur[c] = 0;
      for (int s = jobs[c].tc.col_start >> 3; s < (jobs[c].tc.col_end + 7) >> 3 && s < 2048; ++s) tcol[s] = c;
    }
    int k = 0;
    for (int r = 0; r < fe->sb_rows; ++r)
      for (int s = 0; s < fe->sb_cols; ++s) {
        const int c = tcol[s], n = fe->sb_count[r * fe->sb_cols + s];
        const size_t from = (size_t)(jobs[c].tc.seg_first - fe->seg_blocks) + cur[c];
        memcpy(fe->out_blocks + k, fe->seg_blocks + from, sizeof(vp9hip_block) * (size_t)n);
        memcpy(fe->out_off + 3 * (size_t)k, &fe->seg_off[from][0], sizeof(uint32_t) * 3 * (size_t)n);
        cur[c] += n;
        k += n;
      }
    blocks = fe->out_blocks;
    offs = fe->out_off;
  }

  /* backward adaptation and context refresh (vp9_decode_frame :3566-3586) */
  if (collect) {
    memset(&fe->counts, 0, sizeof(fe->counts));
    for (int c = 0; c < tile_cols; ++c) add_counts(&fe->counts, &jobs[c].counts);
    if (!h->error_res) adapt_probs(fe);
  }
  if (h->refresh_frame_context) fe->saved[h->frame_context_idx] = fe->fc;

  /* the frame's buffer and the reference map (read_uncompressed_header :3258-3276, swap_frame_buffers) */
  const int new_slot = find_free_slot(fe);
  if (new_slot < 0) FE_FAIL(fe, "no free frame buffer");
  out->new_slot = new_slot;
  for (int i = 0; i < 3; ++i) {
    out->ref_slot[i] = intra_frame ? -1 : h->ref_idx[i];
    if (!intra_frame) {
      out->params.ref_width[i] = fe->slot[h->ref_idx[i]].width;
      out->params.ref_height[i] = fe->slot[h->ref_idx[i]].height;
    }
  }
  SlotInfo *ns = &fe->slot[new_slot];
  ns->width = fe->width;
  ns->height = fe->height;
  ns->ss_x = fe->ss_x;
  ns->ss_y = fe->ss_y;
  ns->bit_depth = fe->bit_depth;
  ns->valid = 1;
  for (int i = 0; i < 8; ++i)
    if ((h->refresh_flags >> i) & 1) fe->ref_map[i] = new_slot;
  fe->prev_new_slot = new_slot;

  /* stream state for the next frame (vp9_receive_compressed_data :473-488) */
  fe->last_show_frame = h->show_frame;
  fe->mv_wr_rows[fe->mv_cur] = fe->mi_rows;
  fe->mv_wr_cols[fe->mv_cur] = fe->mi_cols;
  fe->mv_cur ^= 1;
  if (fe->seg.enabled) fe->seg_cur ^= 1;
  fe->last_width = fe->width;
  fe->last_height = fe->height;
  fe->frame_type = h->frame_type;
  fe->intra_only = h->intra_only;
  fe->reset_frame_context = h->reset_frame_context;
  fe->show_frame = h->show_frame;
  fe->have_frame = 1;

  /* what the frame driver takes */
  out->show_frame = h->show_frame;
  out->key_frame = h->frame_type == KEY_FRAME;
  out->intra_only = h->intra_only;
  out->error_resilient = h->error_res;
  out->refresh_flags = h->refresh_flags;
  out->filter_level = h->filter_level;
  out->sharpness = h->sharpness;
  out->tile_cols = tile_cols;
  out->tile_rows = tile_rows;
  vp9hip_frame_params *P = &out->params;
  P->width = fe->width;
  P->height = fe->height;
  P->ss_x = fe->ss_x;
  P->ss_y = fe->ss_y;
  P->bit_depth = fe->bit_depth;
  P->hbd = fe->bit_depth > 8;
  P->lossless = h->lossless;
  P->log2_tile_cols = h->log2_tile_cols;
  P->build_lf_masks = h->filter_level != 0;
  out->blocks = blocks;
  out->n_blocks = total;
  vp9hip_coeff_layout *L = &out->layout;
  int64_t base = 0;
  for (int p = 0; p < 3; ++p) {
    const int ss = p ? fe->ss_x : 0;
    L->eob[p] = fe->eob[p];
    L->eob_stride[p] = ((fe->ctx_cols * 8) >> ss) >> 2;
    L->plane_base[p] = base;
    base += (int64_t)fe->sets[fe->set_idx].coef_cap[p];
    out->dqcoeff[p] = fe->coef[p];
  }
  L->eob_shift = 2;
  L->block_off = offs;
  L->total = base;
  L->compact = 1;
  int nr = 0;
  for (int c = 0; c < tile_cols; ++c)
    for (int p = 0; p < 3; ++p) {
      vp9hip_coeff_region *g = &fe->sets[fe->set_idx].regions[nr++];
      g->plane = p;
      g->reserved = 0;
      g->start = jobs[c].cf_start[p];
      g->count = jobs[c].cf_used[p];
      out->coeff_count += g->count;
    }
  L->regions = fe->sets[fe->set_idx].regions;
  L->n_regions = nr;
  if (fe->trace) {
    const double tr3 = fe_now();
    double mx = 0, sum = 0;
    for (int c = 0; c < tile_cols; ++c) {
      sum += jobs[c].seconds;
      if (jobs[c].seconds > mx) mx = jobs[c].seconds;
    }
    fe->tr_total += tr3 - tr0;
    fe->tr_head += tr1 - tr0;
    fe->tr_tiles += tr2 - tr1;
    fe->tr_tail += tr3 - tr2;
    fe->tr_tile_sum += sum;
    fe->tr_tile_max += mx;
    ++fe->tr_frames;
  }
  return VP9HIP_OK;
}
