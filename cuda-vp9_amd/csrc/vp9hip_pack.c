/*
 * vp9hip_pack.c — host packers (plain C99, no HIP): decoded mode information -> work lists.
 * See include/vp9hip_pack.h for the list of reference functions each part restates; file:line
 * citations below are relative to /root/reference/libvpx/.
 */
#define _POSIX_C_SOURCE 200809L
#include "vp9hip_pack.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- block geometry (vp9/common/vp9_common_data.c: num_4x4_blocks_{wide,high}_lookup) ------- */
static const uint8_t kW4[13] = { 1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16 };
static const uint8_t kH4[13] = { 1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16 };
/* intra_mode_to_tx_type_lookup (vp9/common/vp9_reconintra.c:24-35): DC V H D45 D135 D117 D153 D207 D63 TM */
static const uint8_t kModeToTxType[10] = { 0, 1, 2, 0, 3, 1, 2, 2, 1, 3 };

/* the classes of vp9hip_inter_pred_batch (include/vp9hip.h) */
int vp9hip_inter_class(int w, int h, int unscaled) {
  if (unscaled) {
    if (w == 4 && (h == 4 || h == 8)) return h == 4 ? 0 : 1;
    if (w == 8 && (h == 4 || h == 8 || h == 16)) return h == 4 ? 2 : (h == 8 ? 3 : 4);
    if (w == 16 && (h == 8 || h == 16 || h == 32)) return h == 8 ? 5 : (h == 16 ? 6 : 7);
    if (w == 32 && (h == 16 || h == 32 || h == 64)) return h == 16 ? 8 : (h == 32 ? 9 : 10);
    if (w == 64 && (h == 32 || h == 64)) return h == 32 ? 11 : 12;
  }
  return VP9HIP_INTER_CLASSES - 1;
}

#define MAX_ISLAND_TASKS 4096
#define ISLAND_GROUP_TASKS 192

typedef struct {
  void *p;
  size_t cap; /* bytes */
} vec;

struct pk_pool;
struct pk_job_s;

struct vp9hip_packer {
  char err[256];
  vp9hip_alloc_fn alloc;
  vp9hip_free_fn release;
  void *alloc_user;
  struct pk_pool *pool;
  struct pk_job_s *job;
  int threads;
  int32_t own_base;    /* see the intra dependency maps */
  size_t own_cells[3];
  vec inter, inter_sorted, txb, txb_sorted, intra, intra_isl, intra_big, islands, wave_off, big_wave_start;
  vec level, parent, comp_id, comp_size, order_a, order_b, count;
  vec lvl_map[3], own_map[3];
  vec lfm, lf_raw, rows_expected, lf_skip, comp_box, row_pos;
};

/* Contents are kept when a vector grows (some are appended to across passes).  Only the vectors the
 * frame driver copies to the device (vec_is_output) come from the caller's allocator — page-locked
 * memory is expensive to obtain, so those also start large and grow with headroom. */
static int vec_is_output(const vp9hip_packer *pk, const vec *v);
static int vec_reserve_pk(vp9hip_packer *pk, vec *v, size_t bytes) {
  if (bytes <= v->cap) return 0;
  const int out = pk->alloc && vec_is_output(pk, v);
  size_t ncap = v->cap ? v->cap : (out ? (size_t)1 << 16 : 4096);
  while (ncap < bytes + (out ? bytes >> 1 : 0)) ncap *= 2;
  void *np;
  if (out) {
    np = pk->alloc(pk->alloc_user, ncap);
    if (!np) return -1;
    if (v->p) {
      memcpy(np, v->p, v->cap);
      pk->release(pk->alloc_user, v->p);
    }
  } else {
    np = realloc(v->p, ncap);
    if (!np) return -1;
  }
  v->p = np;
  v->cap = ncap;
  return 0;
}
#define vec_reserve(v, bytes) vec_reserve_pk(pk, (v), (bytes))

static int vec_is_output(const vp9hip_packer *pk, const vec *v) {
  return v == &pk->inter_sorted || v == &pk->txb_sorted || v == &pk->intra_isl || v == &pk->intra_big || v == &pk->islands ||
         v == &pk->wave_off || v == &pk->lfm || v == &pk->rows_expected;
}

int vp9hip_packer_create_ex(vp9hip_packer **out, vp9hip_alloc_fn alloc, vp9hip_free_fn release, void *user) {
  if (!out || (alloc != NULL) != (release != NULL)) return VP9HIP_EINVAL;
  *out = (vp9hip_packer *)calloc(1, sizeof(vp9hip_packer));
  if (!*out) return VP9HIP_ENOMEM;
  (*out)->alloc = alloc;
  (*out)->release = release;
  (*out)->alloc_user = user;
  return VP9HIP_OK;
}

int vp9hip_packer_create(vp9hip_packer **out) { return vp9hip_packer_create_ex(out, NULL, NULL, NULL); }

static void pk_pool_destroy(struct pk_pool *pl);

void vp9hip_packer_destroy(vp9hip_packer *pk) {
  if (!pk) return;
  pk_pool_destroy(pk->pool);
  free(pk->job);
  vec *all[] = { &pk->inter,    &pk->inter_sorted, &pk->txb,       &pk->txb_sorted, &pk->intra,    &pk->intra_isl,
                 &pk->intra_big, &pk->islands,     &pk->wave_off,  &pk->big_wave_start, &pk->level, &pk->parent,
                 &pk->comp_id,  &pk->comp_size,    &pk->order_a,   &pk->order_b,    &pk->count,    &pk->lvl_map[0],
                 &pk->lvl_map[1], &pk->lvl_map[2], &pk->own_map[0], &pk->own_map[1], &pk->own_map[2], &pk->lfm,
                 &pk->lf_raw,   &pk->rows_expected, &pk->lf_skip,   &pk->comp_box,   &pk->row_pos };
  for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) {
    if (!all[i]->p) continue;
    if (pk->release && vec_is_output(pk, all[i])) pk->release(pk->alloc_user, all[i]->p);
    else free(all[i]->p);
  }
  free(pk);
}

const char *vp9hip_packer_error(const vp9hip_packer *pk) { return pk ? pk->err : "null packer"; }

#define PK_FAIL(pk, code, ...)                         \
  do {                                                 \
    snprintf((pk)->err, sizeof((pk)->err), __VA_ARGS__); \
    return (code);                                     \
  } while (0)

/* ---- small libvpx helpers ---------------------------------------------------------------- */

/* uv_txsize_lookup[bsize][tx][ss_x][ss_y] for ss = (1,1) and (0,0) (vp9_common_data.c): the
 * largest transform that fits the chroma block, capped by the luma transform size. */
static int uv_tx_size(int sb_type, int tx_size, int ss) {
  if (sb_type < 3) return 0;
  if (!ss) return tx_size;
  int w4 = kW4[sb_type] >> 1, h4 = kH4[sb_type] >> 1;
  int m = w4 < h4 ? w4 : h4, lg = 0;
  if (m < 1) m = 1;
  while ((2 << lg) <= m) ++lg;
  return tx_size < lg ? tx_size : lg;
}

/* get_tile_offset / vp9_tile_set_col (vp9/common/vp9_tile_common.c:17-32): first mi column of the
 * tile that contains mi_col. */
static int tile_col_start(int mi_col, int mi_cols, int log2_tile_cols) {
  const int sb_cols = (mi_cols + 7) >> 3;
  const int n = 1 << log2_tile_cols;
  int start = 0;
  for (int t = 1; t < n; ++t) {
    int off = ((t * sb_cols) >> log2_tile_cols) << 3;
    if (off > mi_cols) off = mi_cols;
    if (off <= mi_col) start = off;
  }
  return start;
}

/* round_mv_comp_q4 / round_mv_comp_q2 (vp9_reconinter.c:57-63, 73-79) */
static int round_q4(int v) { return (v < 0 ? v - 2 : v + 2) / 4; }

typedef struct {
  int scaled;
  int x_scale_fp, y_scale_fp; /* REF_SCALE_SHIFT = 14 (vp9_scale.h:21) */
  int x_step_q4, y_step_q4;
} scale_factors;

static int scaled_x(int val, const scale_factors *sf) { return (int)((int64_t)val * sf->x_scale_fp >> 14); }
static int scaled_y(int val, const scale_factors *sf) { return (int)((int64_t)val * sf->y_scale_fp >> 14); }

/* vp9_setup_scale_factors_for_frame (vp9/common/vp9_scale.c:46-77); valid_ref_frame_size (:20 of
 * vp9_scale.h: 2*this >= other, this <= 16*other) */
static int setup_scale(scale_factors *sf, int other_w, int other_h, int this_w, int this_h) {
  if (!(2 * this_w >= other_w && 2 * this_h >= other_h && this_w <= 16 * other_w && this_h <= 16 * other_h)) return -1;
  sf->x_scale_fp = (other_w << 14) / this_w;
  sf->y_scale_fp = (other_h << 14) / this_h;
  sf->x_step_q4 = scaled_x(16, sf);
  sf->y_step_q4 = scaled_y(16, sf);
  sf->scaled = other_w != this_w || other_h != this_h; /* vp9_is_scaled: scale_fp != 1<<14 */
  if (!sf->scaled) sf->x_scale_fp = sf->y_scale_fp = 1 << 14;
  return 0;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ---- loop-filter masks -------------------------------------------------------------------- */

typedef struct {
  uint64_t left_y[4], above_y[4], int_4x4_y;
  uint16_t left_uv[4], above_uv[4], int_4x4_uv;
  uint8_t lfl_y[64];
} lfm_raw;

/* vp9_build_mask (vp9/common/vp9_loopfilter.c:1528-1608); the prediction/size masks are the
 * rectangles its tables spell out (:80-195). */
static void lf_build_mask(lfm_raw *lfm, const vp9hip_block *b, int skip) {
  const int level = b->filter_level;
  if (!level) return;
  const int w8 = kW4[b->sb_type] > 1 ? kW4[b->sb_type] >> 1 : 1;
  const int h8 = kH4[b->sb_type] > 1 ? kH4[b->sb_type] >> 1 : 1;
  const int r0 = b->mi_row & 7, c0 = b->mi_col & 7;
  const int shift_y = c0 + (r0 << 3), shift_uv = (c0 >> 1) + ((r0 >> 1) << 2);
  const int build_uv = !(r0 & 1) && !(c0 & 1); /* first_block_in_16x16 */
  const int txy = b->tx_size, txuv = uv_tx_size(b->sb_type, b->tx_size, 1);
  const int wuv = w8 > 1 ? w8 >> 1 : 1, huv = h8 > 1 ? h8 >> 1 : 1;
  uint64_t size_mask = 0, left_pred = 0;
  uint32_t size_mask_uv = 0, left_pred_uv = 0;
  for (int r = 0; r < h8; ++r) {
    memset(&lfm->lfl_y[shift_y + 8 * r], level, (size_t)w8);
    size_mask |= ((1ull << w8) - 1) << (8 * r);
    left_pred |= 1ull << (8 * r);
  }
  for (int r = 0; r < huv; ++r) {
    size_mask_uv |= ((1u << wuv) - 1) << (4 * r);
    left_pred_uv |= 1u << (4 * r);
  }
  const uint64_t above_pred = (1ull << w8) - 1;
  const uint32_t above_pred_uv = (1u << wuv) - 1;
  lfm->above_y[txy] |= above_pred << shift_y;
  lfm->left_y[txy] |= left_pred << shift_y;
  if (build_uv) {
    lfm->above_uv[txuv] |= (uint16_t)(above_pred_uv << shift_uv);
    lfm->left_uv[txuv] |= (uint16_t)(left_pred_uv << shift_uv);
  }
  if (skip && b->ref_frame[0] > 0) return;
  {
    /* above_64x64_txform_mask / left_64x64_txform_mask (:39-78): rows / columns at multiples of the
     * transform size */
    const int t8 = (4 << txy) >> 3 ? (4 << txy) >> 3 : 1;
    uint64_t rows_on = 0, cols_on = 0;
    for (int r = 0; r < 8; r += t8) rows_on |= 0xffull << (8 * r);
    for (int c = 0; c < 8; c += t8) cols_on |= 0x0101010101010101ull << c;
    lfm->above_y[txy] |= (size_mask & rows_on) << shift_y;
    lfm->left_y[txy] |= (size_mask & cols_on) << shift_y;
  }
  if (build_uv) {
    const int tu = (4 << txuv) >> 3 ? (4 << txuv) >> 3 : 1;
    uint32_t rows_uv = 0, cols_uv = 0;
    for (int r = 0; r < 4; r += tu) rows_uv |= 0xfu << (4 * r);
    for (int c = 0; c < 4; c += tu) cols_uv |= 0x1111u << c;
    lfm->above_uv[txuv] |= (uint16_t)((size_mask_uv & rows_uv) << shift_uv);
    lfm->left_uv[txuv] |= (uint16_t)((size_mask_uv & cols_uv) << shift_uv);
  }
  if (txy == 0) lfm->int_4x4_y |= size_mask << shift_y;
  if (build_uv && txuv == 0) lfm->int_4x4_uv |= (uint16_t)((size_mask_uv & 0xffff) << shift_uv);
}

/* vp9_adjust_mask (vp9/common/vp9_loopfilter.c:766-880) */
static void lf_adjust_mask(const lfm_raw *in, vp9hip_lfm *out, int mi_row, int mi_col, int mi_rows, int mi_cols) {
  uint64_t ly[4], ay[4], iy = in->int_4x4_y;
  uint16_t luv[4], auv[4], iuv = in->int_4x4_uv;
  memcpy(ly, in->left_y, sizeof(ly));
  memcpy(ay, in->above_y, sizeof(ay));
  memcpy(luv, in->left_uv, sizeof(luv));
  memcpy(auv, in->above_uv, sizeof(auv));
  const uint64_t LB = 0x1111111111111111ull, AB = 0x000000ff000000ffull;
  ly[2] |= ly[3];
  ay[2] |= ay[3];
  luv[2] |= luv[3];
  auv[2] |= auv[3];
  ly[1] |= ly[0] & LB;
  ly[0] &= ~LB;
  ay[1] |= ay[0] & AB;
  ay[0] &= ~AB;
  luv[1] |= luv[0] & 0x1111;
  luv[0] &= (uint16_t)~0x1111;
  auv[1] |= auv[0] & 0x000f;
  auv[0] &= (uint16_t)~0x000f;
  const int rows = mi_rows - mi_row, cols = mi_cols - mi_col;
  if (rows < 8) {
    const uint64_t my = (1ull << (rows << 3)) - 1;
    const uint16_t muv = (uint16_t)((1u << (((rows + 1) >> 1) << 2)) - 1);
    for (int k = 0; k < 3; ++k) {
      ly[k] &= my;
      ay[k] &= my;
      luv[k] &= muv;
      auv[k] &= muv;
    }
    iy &= my;
    iuv &= muv;
    if (rows == 1) {
      auv[1] |= auv[2];
      auv[2] = 0;
    }
    if (rows == 5) {
      auv[1] |= auv[2] & 0xff00;
      auv[2] &= (uint16_t)~(auv[2] & 0xff00);
    }
  }
  if (cols < 8) {
    const uint64_t my = ((1ull << cols) - 1) * 0x0101010101010101ull;
    const uint16_t muv = (uint16_t)(((1u << ((cols + 1) >> 1)) - 1) * 0x1111);
    const uint16_t muv_int = (uint16_t)(((1u << (cols >> 1)) - 1) * 0x1111);
    for (int k = 0; k < 3; ++k) {
      ly[k] &= my;
      ay[k] &= my;
      luv[k] &= muv;
      auv[k] &= muv;
    }
    iy &= my;
    iuv &= muv_int;
    if (cols == 1) {
      luv[1] |= luv[2];
      luv[2] = 0;
    }
    if (cols == 5) {
      luv[1] |= luv[2] & 0xcccc;
      luv[2] &= (uint16_t)~(luv[2] & 0xcccc);
    }
  }
  if (mi_col == 0) {
    for (int k = 0; k < 3; ++k) {
      ly[k] &= 0xfefefefefefefefeull;
      luv[k] &= 0xeeee;
    }
  }
  memset(out, 0, sizeof(*out));
  for (int k = 0; k < 3; ++k) {
    out->left_y[k] = ly[k];
    out->above_y[k] = ay[k];
    out->left_uv[k] = luv[k];
    out->above_uv[k] = auv[k];
  }
  out->int_4x4_y = iy;
  out->int_4x4_uv = iuv;
  memcpy(out->lfl_y, in->lfl_y, 64);
}

int vp9hip_lf_adjust_masks(const vp9hip_lfm *raw, int sb_rows, int sb_cols, int mi_rows, int mi_cols, vp9hip_lfm *out) {
  if (!raw || !out || sb_rows <= 0 || sb_cols <= 0 || mi_rows <= 0 || mi_cols <= 0 || sb_rows != (mi_rows + 7) >> 3 ||
      sb_cols != (mi_cols + 7) >> 3)
    return VP9HIP_EINVAL;
  for (int r = 0; r < sb_rows; ++r)
    for (int c = 0; c < sb_cols; ++c) {
      const vp9hip_lfm *m = &raw[(size_t)r * sb_cols + c];
      lfm_raw tmp;
      memcpy(tmp.left_y, m->left_y, sizeof(tmp.left_y));
      memcpy(tmp.above_y, m->above_y, sizeof(tmp.above_y));
      tmp.int_4x4_y = m->int_4x4_y;
      memcpy(tmp.left_uv, m->left_uv, sizeof(tmp.left_uv));
      memcpy(tmp.above_uv, m->above_uv, sizeof(tmp.above_uv));
      tmp.int_4x4_uv = m->int_4x4_uv;
      memcpy(tmp.lfl_y, m->lfl_y, 64);
      lf_adjust_mask(&tmp, &out[(size_t)r * sb_cols + c], r * 8, c * 8, mi_rows, mi_cols);
    }
  return VP9HIP_OK;
}

void vp9hip_lf_frame_init(int default_lvl, int sharpness, const int32_t seg_enabled[8], const int32_t seg_data[8],
                          int abs_delta, int mode_ref_delta_enabled, const int8_t ref_deltas[4],
                          const int8_t mode_deltas[2], uint8_t out_lvl[8][4][2], vp9hip_lf_thresh *out_thresh) {
  /* update_sharpness + vp9_loop_filter_init (vp9_loopfilter.c:212-250) */
  if (out_thresh) {
    for (int lvl = 0; lvl < 64; ++lvl) {
      int lim = lvl >> ((sharpness > 0) + (sharpness > 4));
      if (sharpness > 0 && lim > 9 - sharpness) lim = 9 - sharpness;
      if (lim < 1) lim = 1;
      out_thresh->lim[lvl] = (uint8_t)lim;
      out_thresh->mblim[lvl] = (uint8_t)(2 * (lvl + 2) + lim);
      out_thresh->hev_thr[lvl] = (uint8_t)(lvl >> 4);
    }
  }
  if (!out_lvl) return;
  /* vp9_loop_filter_frame_init (:252-295) */
  const int scale = 1 << (default_lvl >> 5);
  for (int s = 0; s < 8; ++s) {
    int lvl_seg = default_lvl;
    if (seg_enabled && seg_enabled[s]) lvl_seg = clampi(abs_delta ? seg_data[s] : default_lvl + seg_data[s], 0, 63);
    if (!mode_ref_delta_enabled) {
      memset(out_lvl[s], lvl_seg, 8);
    } else {
      out_lvl[s][0][0] = (uint8_t)clampi(lvl_seg + ref_deltas[0] * scale, 0, 63);
      out_lvl[s][0][1] = 0; /* never read for intra (mode_lf_lut is 0 for intra modes); libvpx leaves it */
      for (int ref = 1; ref < 4; ++ref)
        for (int m = 0; m < 2; ++m)
          out_lvl[s][ref][m] = (uint8_t)clampi(lvl_seg + ref_deltas[ref] * scale + mode_deltas[m] * scale, 0, 63);
    }
  }
}

/* ---- stable counting sort of fixed-size records by a small key ----------------------------- */
static void counting_sort(const void *src, void *dst, size_t rec, const int32_t *key, int n, int n_keys,
                          int32_t *count /* n_keys + 1 */) {
  memset(count, 0, sizeof(int32_t) * (size_t)(n_keys + 1));
  for (int i = 0; i < n; ++i) ++count[key[i] + 1];
  for (int k = 0; k < n_keys; ++k) count[k + 1] += count[k];
  for (int i = 0; i < n; ++i)
    memcpy((char *)dst + (size_t)count[key[i]]++ * rec, (const char *)src + (size_t)i * rec, rec);
}

/* Order of the islands = order in which their workgroups are dispatched.  The loop filter running beside the
 * walk (vp9hip_intra_islands_lf) reaches superblock (r, c) after about (r + c) superblock steps and needs every
 * island around it finished by then, and an island takes its depth in waves: earliest deadline first, deadline =
 * (first row + first column) steps minus the island's own duration (a step of the filter is about 2.2 wave times
 * of the walk; measured).  In front of that: the group g = max(first superblock row - 1, 0) —
 * filter row r is placed behind the islands of groups <= r in the fused launch's grid, which are all the islands
 * that touch superblock rows <= r + 1, i.e. all it ever waits for. */
static int island_group(const vp9hip_intra_island *x) {
  const int r = (int)(x->reserved & 255);
  return r > 0 ? r - 1 : 0;
}
static int island_key(const vp9hip_intra_island *x) {
  const int r = (int)(x->reserved & 255), c = (int)((x->reserved >> 16) & 255);
  return 56 * (r + c) - 60 * (int)x->n_waves;
}
static int island_deadline_first(const void *a, const void *b) {
  const vp9hip_intra_island *x = (const vp9hip_intra_island *)a, *y = (const vp9hip_intra_island *)b;
  const int gx = island_group(x), gy = island_group(y);
  if (gx != gy) return gx < gy ? -1 : 1;
  const int kx = island_key(x), ky = island_key(y);
  if (kx != ky) return kx < ky ? -1 : 1;
  return x->task_start < y->task_start ? -1 : (x->task_start > y->task_start);
}

/* per component (later: per island group): the bounding box of its blocks in every plane + the number of full
 * 32x32 transforms — what VP9HIP_ISLAND_FITS looks at */
typedef struct {
  int32_t box[3][4]; /* x0, y0, x1, y1 in samples of the plane */
  int32_t n_tx32;
} comp_box;
static void comp_box_init(comp_box *b) {
  for (int p = 0; p < 3; ++p) {
    b->box[p][0] = b->box[p][1] = INT_MAX;
    b->box[p][2] = b->box[p][3] = 0;
  }
  b->n_tx32 = 0;
}
static void comp_box_union(comp_box *d, const comp_box *a, const comp_box *b) {
  for (int p = 0; p < 3; ++p) {
    d->box[p][0] = a->box[p][0] < b->box[p][0] ? a->box[p][0] : b->box[p][0];
    d->box[p][1] = a->box[p][1] < b->box[p][1] ? a->box[p][1] : b->box[p][1];
    d->box[p][2] = a->box[p][2] > b->box[p][2] ? a->box[p][2] : b->box[p][2];
    d->box[p][3] = a->box[p][3] > b->box[p][3] ? a->box[p][3] : b->box[p][3];
  }
  d->n_tx32 = a->n_tx32 + b->n_tx32;
}

static int uf_find(int32_t *parent, int a) {
  while (parent[a] != a) {
    parent[a] = parent[parent[a]];
    a = parent[a];
  }
  return a;
}


static double pk_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + 1e-3 * (double)ts.tv_nsec;
}

/* ---- a small fork/join pool (the packer's passes over the block list are split into ranges of whole
 * superblocks; SURVEY §8f-2/f-3: the host side of the path must not be the bottleneck) ------------- */
#include <pthread.h>
#include <unistd.h>

#define PK_MAX_THREADS 16

typedef void (*pk_fn)(void *arg, int tid);

struct pk_pool {
  int n; /* workers besides the caller */
  pthread_t th[PK_MAX_THREADS];
  pthread_mutex_t mu;
  pthread_cond_t cv_go, cv_done;
  unsigned gen;
  int pending, stop;
  pk_fn fn;
  void *arg;
  int ids[PK_MAX_THREADS];
  struct pk_pool *self[PK_MAX_THREADS];
};

typedef struct {
  struct pk_pool *pool;
  int tid;
} pk_worker_arg;

static void *pk_worker(void *argp) {
  pk_worker_arg *wa = (pk_worker_arg *)argp;
  struct pk_pool *pl = wa->pool;
  const int tid = wa->tid;
  unsigned seen = 0;
  free(wa);
  for (;;) {
    pthread_mutex_lock(&pl->mu);
    while (pl->gen == seen && !pl->stop) pthread_cond_wait(&pl->cv_go, &pl->mu);
    if (pl->stop) {
      pthread_mutex_unlock(&pl->mu);
      return NULL;
    }
    seen = pl->gen;
    pk_fn fn = pl->fn;
    void *arg = pl->arg;
    pthread_mutex_unlock(&pl->mu);
    fn(arg, tid);
    pthread_mutex_lock(&pl->mu);
    if (--pl->pending == 0) pthread_cond_signal(&pl->cv_done);
    pthread_mutex_unlock(&pl->mu);
  }
}

static struct pk_pool *pk_pool_create(int workers) {
  struct pk_pool *pl = (struct pk_pool *)calloc(1, sizeof(*pl));
  if (!pl) return NULL;
  pthread_mutex_init(&pl->mu, NULL);
  pthread_cond_init(&pl->cv_go, NULL);
  pthread_cond_init(&pl->cv_done, NULL);
  for (int i = 0; i < workers && i < PK_MAX_THREADS - 1; ++i) {
    pk_worker_arg *wa = (pk_worker_arg *)malloc(sizeof(*wa));
    if (!wa) break;
    wa->pool = pl;
    wa->tid = i + 1;
    if (pthread_create(&pl->th[pl->n], NULL, pk_worker, wa)) {
      free(wa);
      break;
    }
    ++pl->n;
  }
  return pl;
}

static void pk_pool_destroy(struct pk_pool *pl) {
  if (!pl) return;
  pthread_mutex_lock(&pl->mu);
  pl->stop = 1;
  pthread_cond_broadcast(&pl->cv_go);
  pthread_mutex_unlock(&pl->mu);
  for (int i = 0; i < pl->n; ++i) pthread_join(pl->th[i], NULL);
  pthread_mutex_destroy(&pl->mu);
  pthread_cond_destroy(&pl->cv_go);
  pthread_cond_destroy(&pl->cv_done);
  free(pl);
}

/* fn(arg, tid) for tid 0..threads-1; the caller is tid 0 */
static void pk_parallel(struct pk_pool *pl, int threads, pk_fn fn, void *arg) {
  if (!pl || threads <= 1) {
    for (int t = 0; t < threads; ++t) fn(arg, t);
    return;
  }
  /* workers beyond `threads` return at once (their range is empty) */
  pthread_mutex_lock(&pl->mu);
  pl->fn = fn;
  pl->arg = arg;
  pl->pending = pl->n;
  ++pl->gen;
  pthread_cond_broadcast(&pl->cv_go);
  pthread_mutex_unlock(&pl->mu);
  fn(arg, 0);
  pthread_mutex_lock(&pl->mu);
  while (pl->pending) pthread_cond_wait(&pl->cv_done, &pl->mu);
  pthread_mutex_unlock(&pl->mu);
}

/* ---- the per-range passes -------------------------------------------------------------------- */
typedef struct pk_job_s {
  /* inputs */
  vp9hip_packer *pk;
  const vp9hip_frame_params *P;
  const vp9hip_block *blocks;
  const vp9hip_coeff_layout *coeffs;
  int n_blocks, threads, ss, mi_cols, mi_rows, sb_cols, sb_rows, have_eobs;
  int range[PK_MAX_THREADS + 1]; /* block ranges, split at superblock boundaries */
  const void *sf;                /* scale_factors[3] */
  const int *sf_valid;
  /* per-range results of pass 0 */
  int32_t n_inter[PK_MAX_THREADS], n_tx_ub[PK_MAX_THREADS], n_intra[PK_MAX_THREADS];
  int64_t coeff[PK_MAX_THREADS][3];
  /* offsets handed to pass 1 */
  int32_t inter_off[PK_MAX_THREADS], txb_off[PK_MAX_THREADS], intra_off[PK_MAX_THREADS];
  int64_t run_off[PK_MAX_THREADS][3];
  int64_t coeff_base[3], coeff_total;
  /* per-range results of pass 1 */
  int32_t n_txb[PK_MAX_THREADS], hist_inter[PK_MAX_THREADS][VP9HIP_INTER_CLASSES], hist_txb[PK_MAX_THREADS][4];
  uint32_t refs_used[PK_MAX_THREADS];
  /* sorted positions handed to the scatter */
  int32_t pos_inter[PK_MAX_THREADS][VP9HIP_INTER_CLASSES], pos_txb[PK_MAX_THREADS][4];
  /* arrays */
  vp9hip_inter_task *it, *it_sorted;
  vp9hip_txb *tb, *tb_sorted;
  vp9hip_intra_task *ia;
  int32_t *key;
  uint8_t *lf_skip;
  void *lf_raw; /* lfm_raw[sb_rows * sb_cols] or NULL */
  vp9hip_lfm *lfm;
  /* errors */
  int err[PK_MAX_THREADS];
  char errmsg[PK_MAX_THREADS][160];
} pk_job;

#define JOB_FAIL(j, tid, ...)                                        \
  do {                                                               \
    (j)->err[tid] = VP9HIP_EINVAL;                                   \
    snprintf((j)->errmsg[tid], sizeof((j)->errmsg[tid]), __VA_ARGS__); \
    return;                                                          \
  } while (0)

/* vp9hip_pack.h: rows / coefficients of a transform block's slot that can be non-zero */
int vp9hip_coeff_rows(int eob, int tx_type, int tx_size) {
  const int n = 4 << tx_size;
  if (eob <= 0) return 0;
  if (eob == 1) return 1;
  if (tx_type == 0 && tx_size <= 2 && eob <= 10) return 4;
  if (tx_size == 3 && eob <= 34) return 8;
  return n;
}
int vp9hip_coeff_extent(int eob, int tx_type, int tx_size) {
  return vp9hip_coeff_rows(eob, tx_type, tx_size) * (4 << tx_size);
}

/* pass 0: validate, count (exact but for the residual records, whose number depends on the eobs: upper bound) */
static void pk_pass0(void *argp, int tid) {
  pk_job *j = (pk_job *)argp;
  if (tid >= j->threads) return;
  const int ss = j->ss, mi_cols = j->mi_cols, mi_rows = j->mi_rows;
  int32_t n_inter = 0, n_tx_ub = 0, n_intra = 0;
  int64_t cc[3] = { 0, 0, 0 };
  for (int i = j->range[tid]; i < j->range[tid + 1]; ++i) {
    const vp9hip_block *b = &j->blocks[i];
    if (b->sb_type > 12 || b->tx_size > 3 || b->mi_row < 0 || b->mi_col < 0 || b->mi_row >= mi_rows || b->mi_col >= mi_cols)
      JOB_FAIL(j, tid, "vp9hip_pack_frame: block %d out of range (sb_type %d tx %d at mi %d,%d)", i, b->sb_type, b->tx_size,
               b->mi_row, b->mi_col);
    const int bw8 = kW4[b->sb_type] > 1 ? kW4[b->sb_type] >> 1 : 1, bh8 = kH4[b->sb_type] > 1 ? kH4[b->sb_type] >> 1 : 1;
    const int to_right = (mi_cols - bw8 - b->mi_col) * 64, to_bottom = (mi_rows - bh8 - b->mi_row) * 64;
    const int inter = b->ref_frame[0] > 0;
    if (inter) {
      for (int r = 0; r < 1 + (b->ref_frame[1] > 0); ++r) {
        const int k = b->ref_frame[r] - 1;
        if (k < 0 || k > 2 || !j->sf_valid[k])
          JOB_FAIL(j, tid, "vp9hip_pack_frame: block %d uses reference %d which has no valid size", i, k + 1);
      }
      if (b->interp_filter > 3) JOB_FAIL(j, tid, "vp9hip_pack_frame: block %d bad interp_filter", i);
      n_inter += b->sb_type < 3 ? (4 + 2 * (ss ? 1 : 4)) : 3;
    } else if (b->mode > 9 || b->uv_mode > 9) {
      JOB_FAIL(j, tid, "vp9hip_pack_frame: block %d bad intra mode", i);
    }
    for (int p = 0; p < 3; ++p) {
      const int s = p ? ss : 0;
      const int n4w = (bw8 * 2) >> s ? (bw8 * 2) >> s : 1, n4h = (bh8 * 2) >> s ? (bh8 * 2) >> s : 1;
      const int tx = p ? uv_tx_size(b->sb_type, b->tx_size, ss) : b->tx_size;
      const int mw = n4w + (to_right >= 0 ? 0 : to_right >> (5 + s)), mh = n4h + (to_bottom >= 0 ? 0 : to_bottom >> (5 + s));
      const int step = 1 << tx;
      if (mw <= 0 || mh <= 0) continue;
      const int cnt = ((mw + step - 1) / step) * ((mh + step - 1) / step);
      if (!inter) n_intra += cnt;
      else if (!b->skip) n_tx_ub += cnt;
      if (!b->skip) cc[p] += (int64_t)cnt * (16 << (2 * tx));
    }
  }
  j->n_inter[tid] = n_inter;
  j->n_tx_ub[tid] = n_tx_ub;
  j->n_intra[tid] = n_intra;
  memcpy(j->coeff[tid], cc, sizeof(cc));
}

/* pass 1: the records of a range of blocks, in decode order, into the range's slice of the unsorted
 * arrays; class histograms for the scatter; loop-filter masks of the range's superblocks */
static void pk_pass1(void *argp, int tid) {
  pk_job *j = (pk_job *)argp;
  if (tid >= j->threads) return;
  const vp9hip_frame_params *P = j->P;
  const vp9hip_block *blocks = j->blocks;
  const vp9hip_coeff_layout *coeffs = j->coeffs;
  const scale_factors *sf = (const scale_factors *)j->sf;
  const int ss = j->ss, mi_cols = j->mi_cols, mi_rows = j->mi_rows, have_eobs = j->have_eobs;
  vp9hip_inter_task *it = j->it;
  vp9hip_txb *tb = j->tb;
  vp9hip_intra_task *ia = j->ia;
  int32_t *key = j->key;
  uint8_t *lf_skip = j->lf_skip;
  lfm_raw *raw = (lfm_raw *)j->lf_raw;
  int ni = j->inter_off[tid], nt = j->txb_off[tid], na = j->intra_off[tid];
  int64_t run[3] = { j->run_off[tid][0], j->run_off[tid][1], j->run_off[tid][2] };
  uint32_t refs_used = 0;
  int32_t hist_inter[VP9HIP_INTER_CLASSES] = { 0 }, hist_txb[4] = { 0, 0, 0, 0 }; /* local: no shared cache lines */
  if (raw && j->range[tid] < j->range[tid + 1]) {
    /* the range is a run of whole superblocks in decode order: clear exactly the records it owns */
    int prev = -1;
    for (int i = j->range[tid]; i < j->range[tid + 1]; ++i) {
      const int sb = (blocks[i].mi_row >> 3) * j->sb_cols + (blocks[i].mi_col >> 3);
      if (sb != prev) {
        memset(&raw[sb], 0, sizeof(lfm_raw));
        prev = sb;
      }
    }
  }
  for (int i = j->range[tid]; i < j->range[tid + 1]; ++i) {
    const vp9hip_block *b = &blocks[i];
    const int bw8 = kW4[b->sb_type] > 1 ? kW4[b->sb_type] >> 1 : 1, bh8 = kH4[b->sb_type] > 1 ? kH4[b->sb_type] >> 1 : 1;
    /* set_mi_row_col (vp9/common/vp9_onyxc_int.h): distances to the frame edges in 1/8 sample */
    const int to_left = -(b->mi_col * 64), to_right = (mi_cols - bw8 - b->mi_col) * 64;
    const int to_top = -(b->mi_row * 64), to_bottom = (mi_rows - bh8 - b->mi_row) * 64;
    const int inter = b->ref_frame[0] > 0;
    const int sub8 = b->sb_type < 3;
    const int left_mi = b->mi_col > tile_col_start(b->mi_col, mi_cols, P->log2_tile_cols);

    if (inter) {
      const int compound = b->ref_frame[1] > 0;
      for (int p = 0; p < 3; ++p) {
        const int s = p ? ss : 0;
        const int n4w = (bw8 * 2) >> s ? (bw8 * 2) >> s : 1, n4h = (bh8 * 2) >> s ? (bh8 * 2) >> s : 1;
        const int bw = 4 * n4w, bh = 4 * n4h; /* plane block size as passed to dec_build_inter_predictors */
        const int nby = sub8 ? n4h : 1, nbx = sub8 ? n4w : 1;
        const int x_start = (b->mi_col * 8) >> s, y_start = (b->mi_row * 8) >> s;
        for (int by = 0; by < nby; ++by)
          for (int bx = 0; bx < nbx; ++bx) {
            vp9hip_inter_task *t = &it[ni];
            const int x = 4 * bx, y = 4 * by, blk = by * nbx + bx;
            int any_scaled = 0;
            memset(t, 0, sizeof(*t));
            t->dst_x = (int16_t)(x_start + x);
            t->dst_y = (int16_t)(y_start + y);
            t->w = (uint8_t)(sub8 ? 4 : bw);
            t->h = (uint8_t)(sub8 ? 4 : bh);
            t->plane = (uint8_t)p;
            t->flags = (uint8_t)(compound | (b->interp_filter << 1));
            for (int r = 0; r < 1 + compound; ++r) {
              const int k = b->ref_frame[r] - 1;
              const scale_factors *f = &sf[k];
              int mvr, mvc;
              if (!sub8) {
                mvr = b->mv[r][0];
                mvc = b->mv[r][1];
              } else if (!s) { /* average_split_mvs, ss_idx 0 */
                mvr = b->sub_mv[blk][r][0];
                mvc = b->sub_mv[blk][r][1];
              } else { /* ss_idx 3: mi_mv_pred_q4 */
                mvr = round_q4(b->sub_mv[0][r][0] + b->sub_mv[1][r][0] + b->sub_mv[2][r][0] + b->sub_mv[3][r][0]);
                mvc = round_q4(b->sub_mv[0][r][1] + b->sub_mv[1][r][1] + b->sub_mv[2][r][1] + b->sub_mv[3][r][1]);
              }
              refs_used |= 1u << k;
              t->ref[r] = (uint8_t)k;
              if (!f->scaled) {
                /* vp9_decodeframe.c:620-632: no clamp; q4 position = 16 * sample + mv in 1/16 */
                t->pos_x[r] = ((x_start + x) << 4) + mvc * (1 << (1 - s));
                t->pos_y[r] = ((y_start + y) << 4) + mvr * (1 << (1 - s));
                t->step_x[r] = t->step_y[r] = 16;
              } else {
                /* vp9_decodeframe.c:566-619: clamp_mv_to_umv_border_sb, then scale */
                const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16;
                const int spel_top = (4 + bh) << 4, spel_bottom = spel_top - 16;
                int q4r = (int16_t)(mvr * (1 << (1 - s))), q4c = (int16_t)(mvc * (1 << (1 - s)));
                q4c = clampi(q4c, to_left * (1 << (1 - s)) - spel_left, to_right * (1 << (1 - s)) + spel_right);
                q4r = clampi(q4r, to_top * (1 << (1 - s)) - spel_top, to_bottom * (1 << (1 - s)) + spel_bottom);
                /* vp9_scale_mv (vp9_scale.c:37-44) is given LUMA-grid block coordinates + the plane offset */
                const int x_off_q4 = scaled_x((b->mi_col * 8 + x) << 4, f) & 15;
                const int y_off_q4 = scaled_y((b->mi_row * 8 + y) << 4, f) & 15;
                const int smv_c = scaled_x(q4c, f) + x_off_q4, smv_r = scaled_y(q4r, f) + y_off_q4;
                t->pos_x[r] = (scaled_x(x_start + x, f) << 4) + smv_c;
                t->pos_y[r] = (scaled_y(y_start + y, f) << 4) + smv_r;
                t->step_x[r] = (uint8_t)f->x_step_q4;
                t->step_y[r] = (uint8_t)f->y_step_q4;
                any_scaled = 1;
              }
            }
            if (!compound) {
              t->step_x[1] = t->step_y[1] = 16;
            }
            {
              const int cls = vp9hip_inter_class(t->w, t->h, !any_scaled);
              key[ni] = cls;
              ++hist_inter[cls];
            }
            ++ni;
          }
      }
    }

    /* transform blocks: vp9_foreach_transformed_block_in_plane order, clipped to the frame */
    int eobtotal = 0;
    const uint32_t *boff = (coeffs && coeffs->block_off) ? &coeffs->block_off[3 * (size_t)i] : NULL;
    const int compact = boff && coeffs->compact;
    int64_t brun[3] = { 0, 0, 0 };
    for (int p = 0; p < 3; ++p) {
      const int s = p ? ss : 0;
      const int n4w = (bw8 * 2) >> s ? (bw8 * 2) >> s : 1, n4h = (bh8 * 2) >> s ? (bh8 * 2) >> s : 1;
      const int tx = p ? uv_tx_size(b->sb_type, b->tx_size, ss) : b->tx_size;
      const int mw = n4w + (to_right >= 0 ? 0 : to_right >> (5 + s)), mh = n4h + (to_bottom >= 0 ? 0 : to_bottom >> (5 + s));
      const int step = 1 << tx, nn = 16 << (2 * tx);
      const int x_start = (b->mi_col * 8) >> s, y_start = (b->mi_row * 8) >> s;
      for (int row = 0; row < mh; row += step)
        for (int col = 0; col < mw; col += step) {
          const int x = x_start + 4 * col, y = y_start + 4 * row;
          int eob = 0;
          uint32_t off = 0;
          if (!b->skip) {
            if (coeffs && coeffs->eob[p])
              eob = coeffs->eob[p][(size_t)(y >> coeffs->eob_shift) * coeffs->eob_stride[p] + (x >> coeffs->eob_shift)];
            else if (P->assume_coded)
              eob = 1;
            if (eob < 0 || eob > nn) JOB_FAIL(j, tid, "vp9hip_pack_frame: bad eob %d (block %d plane %d)", eob, i, p);
            eobtotal += eob;
            off = (uint32_t)(j->coeff_base[p] + run[p]);
            if (boff) off = (uint32_t)(j->coeff_base[p] + boff[p] + brun[p]);
            int ext = nn;
            if (compact) { /* the slot holds the rows the clearing rule leaves (vp9hip_coeff_extent), nothing at eob 0 */
              int txt = 0;
              if (!inter && !P->lossless && p == 0 && tx < 3) {
                const int m = sub8 ? b->sub_mode[(row << 1) + col] : b->mode;
                txt = m <= 9 ? kModeToTxType[m] : 0;
              }
              ext = vp9hip_coeff_extent(eob, txt, tx);
            }
            if (eob > 0 && (int64_t)off + ext > j->coeff_total)
              JOB_FAIL(j, tid, "vp9hip_pack_frame: coefficient slot of block %d plane %d ends past the buffer", i, p);
            brun[p] += ext;
            run[p] += nn;
          }
          if (inter) {
            if (eob > 0) {
              vp9hip_txb *r = &tb[nt++];
              ++hist_txb[tx];
              memset(r, 0, sizeof(*r));
              r->coeff_off = off;
              r->x = (uint16_t)x;
              r->y = (uint16_t)y;
              r->plane = (uint8_t)p;
              r->tx_size = (uint8_t)tx;
              r->tx_type = P->lossless ? 0x80 : 0;
              r->eob = (uint16_t)eob;
            }
          } else {
            vp9hip_intra_task *r = &ia[na++];
            int mode = p ? b->uv_mode : b->mode;
            if (sub8 && p == 0) mode = b->sub_mode[(row << 1) + col];
            if (mode > 9) JOB_FAIL(j, tid, "vp9hip_pack_frame: block %d bad intra sub-mode", i);
            memset(r, 0, sizeof(*r));
            r->coeff_off = off;
            r->x = (uint16_t)x;
            r->y = (uint16_t)y;
            r->plane = (uint8_t)p;
            r->tx_size = (uint8_t)tx;
            r->tx_type = P->lossless ? 0x80 : (uint8_t)((p || tx == 3) ? 0 : kModeToTxType[mode]);
            r->mode = (uint8_t)mode;
            r->eob = (uint16_t)eob;
            /* vp9_predict_intra_block (vp9_reconintra.c:409-415) */
            r->flags = (uint8_t)((row > 0 || b->mi_row > 0) | ((col > 0 || left_mi) << 1) | (((col + step) < n4w) << 2));
          }
        }
    }
    lf_skip[i] = (uint8_t)(b->skip || (have_eobs && inter && !sub8 && eobtotal == 0));
    if (raw) lf_build_mask(&raw[(size_t)(b->mi_row >> 3) * j->sb_cols + (b->mi_col >> 3)], b, lf_skip[i]);
  }

  j->n_txb[tid] = nt - j->txb_off[tid];
  j->refs_used[tid] = refs_used;
  memcpy(j->hist_inter[tid], hist_inter, sizeof(hist_inter));
  memcpy(j->hist_txb[tid], hist_txb, sizeof(hist_txb));
  if (ni != j->inter_off[tid] + j->n_inter[tid] || na != j->intra_off[tid] + j->n_intra[tid] ||
      run[0] != j->run_off[tid][0] + j->coeff[tid][0] || run[1] != j->run_off[tid][1] + j->coeff[tid][1] ||
      run[2] != j->run_off[tid][2] + j->coeff[tid][2])
    JOB_FAIL(j, tid, "vp9hip_pack_frame: internal count mismatch");
  if (raw) {
    int prev = -1;
    for (int i = j->range[tid]; i < j->range[tid + 1]; ++i) {
      const int r = blocks[i].mi_row >> 3, c = blocks[i].mi_col >> 3, sb = r * j->sb_cols + c;
      if (sb != prev) {
        lf_adjust_mask(&raw[sb], &j->lfm[sb], r * 8, c * 8, mi_rows, mi_cols);
        prev = sb;
      }
    }
  }
}

/* scatter: every range moves its records to their final, class-sorted places (a stable counting sort
 * whose counting was done in pass 1) */
static void pk_scatter(void *argp, int tid) {
  pk_job *j = (pk_job *)argp;
  if (tid >= j->threads) return;
  int32_t pos[VP9HIP_INTER_CLASSES];
  memcpy(pos, j->pos_inter[tid], sizeof(pos));
  for (int i = j->inter_off[tid], e = i + j->n_inter[tid]; i < e; ++i) j->it_sorted[pos[j->key[i]]++] = j->it[i];
  memcpy(pos, j->pos_txb[tid], sizeof(int32_t) * 4);
  for (int i = j->txb_off[tid], e = i + j->n_txb[tid]; i < e; ++i) j->tb_sorted[pos[j->tb[i].tx_size]++] = j->tb[i];
}

/* ---- the frame packer ---------------------------------------------------------------------- */
#define PK_MARK(name)                                                      \
  do {                                                                     \
    if (trace) {                                                           \
      const double n_ = pk_now();                                          \
      fprintf(stderr, "  pack: %-28s %8.1f us\n", name, n_ - t_mark);       \
      t_mark = n_;                                                         \
    }                                                                      \
  } while (0)

int vp9hip_pack_frame(vp9hip_packer *pk, const vp9hip_frame_params *P, const vp9hip_block *blocks, int n_blocks,
                      const vp9hip_coeff_layout *coeffs, vp9hip_packed *out) {
  if (!pk) return VP9HIP_EINVAL;
  if (!P || !out || n_blocks < 0 || (n_blocks && !blocks)) PK_FAIL(pk, VP9HIP_EINVAL, "vp9hip_pack_frame: null argument");
  if (P->width <= 0 || P->height <= 0 || P->width > 16384 || P->height > 16384)
    PK_FAIL(pk, VP9HIP_EINVAL, "vp9hip_pack_frame: bad frame size %dx%d", P->width, P->height);
  if (P->ss_x != P->ss_y || (P->ss_x != 0 && P->ss_x != 1))
    PK_FAIL(pk, VP9HIP_EINVAL, "vp9hip_pack_frame: only 4:2:0 and 4:4:4 are supported (ss %d,%d)", P->ss_x, P->ss_y);
  const int trace = getenv("VP9HIP_PACK_TRACE") != NULL;
  double t_mark = trace ? pk_now() : 0.0;
  const int ss = P->ss_x;
  const int aw = (P->width + 7) & ~7, ah = (P->height + 7) & ~7;
  const int mi_cols = aw >> 3, mi_rows = ah >> 3;
  const int sb_cols = (mi_cols + 7) >> 3, sb_rows = (mi_rows + 7) >> 3;
  const int paw[3] = { aw, aw >> ss, aw >> ss }, pah[3] = { ah, ah >> ss, ah >> ss };
  memset(out, 0, sizeof(*out));

  scale_factors sf[3];
  memset(sf, 0, sizeof(sf));
  int sf_valid[3] = { 0, 0, 0 };
  for (int r = 0; r < 3; ++r)
    if (P->ref_width[r] > 0 && P->ref_height[r] > 0)
      sf_valid[r] = setup_scale(&sf[r], P->ref_width[r], P->ref_height[r], P->width, P->height) == 0;

  /* ---- ranges of whole superblocks, one per thread ------------------------------------------------ */
  pk_job *j = pk->job;
  if (!j) {
    j = pk->job = (pk_job *)calloc(1, sizeof(pk_job));
    if (!j) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
  }
  int threads = pk->threads;
  if (threads <= 0) {
    const char *e = getenv("VP9HIP_PACK_THREADS");
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    threads = e ? atoi(e) : (nc >= 16 ? 8 : nc >= 8 ? 4 : nc >= 4 ? 2 : 1);
    if (threads < 1) threads = 1;
    if (threads > PK_MAX_THREADS) threads = PK_MAX_THREADS;
    pk->threads = threads;
  }
  if (n_blocks < 2048) threads = 1; /* not worth a fork/join */
  if (threads > 1 && !pk->pool) pk->pool = pk_pool_create(pk->threads - 1);
  if (threads > 1 && (!pk->pool || pk->pool->n + 1 < threads)) threads = pk->pool ? pk->pool->n + 1 : 1;
  j->pk = pk;
  j->P = P;
  j->blocks = blocks;
  j->coeffs = coeffs;
  j->n_blocks = n_blocks;
  j->threads = threads;
  j->ss = ss;
  j->mi_cols = mi_cols;
  j->mi_rows = mi_rows;
  j->sb_cols = sb_cols;
  j->sb_rows = sb_rows;
  j->have_eobs = coeffs && coeffs->eob[0];
  j->sf = sf;
  j->sf_valid = sf_valid;
  memset(j->err, 0, sizeof(j->err));
  j->range[0] = 0;
  for (int t = 1; t < threads; ++t) {
    int i = (int)((int64_t)n_blocks * t / threads);
    if (i < j->range[t - 1]) i = j->range[t - 1];
    /* forward to the first block of the next superblock */
    while (i > 0 && i < n_blocks && (blocks[i].mi_row >> 3) == (blocks[i - 1].mi_row >> 3) &&
           (blocks[i].mi_col >> 3) == (blocks[i - 1].mi_col >> 3))
      ++i;
    j->range[t] = i;
  }
  j->range[threads] = n_blocks;

  /* ---- pass 0: sizes ---------------------------------------------------------------------- */
  pk_parallel(pk->pool, threads, pk_pass0, j);
  for (int t = 0; t < threads; ++t)
    if (j->err[t]) PK_FAIL(pk, j->err[t], "%s", j->errmsg[t]);
  size_t n_inter = 0, n_tx = 0, n_intra_total = 0;
  int64_t coeff_count[3] = { 0, 0, 0 };
  for (int t = 0; t < threads; ++t) {
    j->inter_off[t] = (int32_t)n_inter;
    j->txb_off[t] = (int32_t)n_tx;
    j->intra_off[t] = (int32_t)n_intra_total;
    for (int p = 0; p < 3; ++p) {
      j->run_off[t][p] = coeff_count[p];
      coeff_count[p] += j->coeff[t][p];
    }
    n_inter += (size_t)j->n_inter[t];
    n_tx += (size_t)j->n_tx_ub[t];
    n_intra_total += (size_t)j->n_intra[t];
  }
  PK_MARK("pass 0 (sizes)");
  out->coeff_base[0] = 0;
  out->coeff_base[1] = coeff_count[0];
  out->coeff_base[2] = coeff_count[0] + coeff_count[1];
  out->coeff_total = coeff_count[0] + coeff_count[1] + coeff_count[2];
  memcpy(out->coeff_count, coeff_count, sizeof(coeff_count));
  if (coeffs && coeffs->block_off) {
    /* slots placed by the caller (one region per tile column): offsets come with the blocks */
    memcpy(out->coeff_base, coeffs->plane_base, sizeof(out->coeff_base));
    out->coeff_total = coeffs->total;
  }
  if (out->coeff_total > (int64_t)UINT32_MAX) PK_FAIL(pk, VP9HIP_EINVAL, "vp9hip_pack_frame: too many coefficients");
  memcpy(j->coeff_base, out->coeff_base, sizeof(j->coeff_base));
  j->coeff_total = out->coeff_total;
  if (coeffs && coeffs->compact && !coeffs->block_off)
    PK_FAIL(pk, VP9HIP_EINVAL, "vp9hip_pack_frame: the compact coefficient layout needs block_off");

  const size_t n_sb = (size_t)sb_rows * sb_cols;
  if (vec_reserve(&pk->inter, (n_inter + 1) * sizeof(vp9hip_inter_task)) ||
      vec_reserve(&pk->inter_sorted, (n_inter + 1) * sizeof(vp9hip_inter_task)) ||
      vec_reserve(&pk->txb, (n_tx + 1) * sizeof(vp9hip_txb)) || vec_reserve(&pk->txb_sorted, (n_tx + 1) * sizeof(vp9hip_txb)) ||
      vec_reserve(&pk->intra, (n_intra_total + 1) * sizeof(vp9hip_intra_task)) ||
      vec_reserve(&pk->order_a, (n_tx + n_inter + n_intra_total + 1) * sizeof(int32_t)) ||
      vec_reserve(&pk->lf_skip, (size_t)n_blocks + 1) ||
      (P->build_lf_masks && (vec_reserve(&pk->lf_raw, n_sb * sizeof(lfm_raw)) || vec_reserve(&pk->lfm, n_sb * sizeof(vp9hip_lfm)))))
    PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
  vp9hip_inter_task *it = (vp9hip_inter_task *)pk->inter.p;
  vp9hip_txb *tb = (vp9hip_txb *)pk->txb.p;
  vp9hip_intra_task *ia = (vp9hip_intra_task *)pk->intra.p;
  int32_t *key = (int32_t *)pk->order_a.p;
  j->it = it;
  j->it_sorted = (vp9hip_inter_task *)pk->inter_sorted.p;
  j->tb = tb;
  j->tb_sorted = (vp9hip_txb *)pk->txb_sorted.p;
  j->ia = ia;
  j->key = key;
  j->lf_skip = (uint8_t *)pk->lf_skip.p;
  j->lf_raw = P->build_lf_masks ? pk->lf_raw.p : NULL;
  j->lfm = P->build_lf_masks ? (vp9hip_lfm *)pk->lfm.p : NULL;
  if (P->build_lf_masks) {
    /* superblocks no block list entry starts in (none in a well-formed frame) still get a defined record */
    if ((size_t)n_blocks < n_sb) memset(pk->lfm.p, 0, n_sb * sizeof(vp9hip_lfm));
  }

  /* ---- pass 1: records in decode order, masks --------------------------------------------------- */
  pk_parallel(pk->pool, threads, pk_pass1, j);
  for (int t = 0; t < threads; ++t)
    if (j->err[t]) PK_FAIL(pk, j->err[t], "%s", j->errmsg[t]);
  PK_MARK("pass 1 (records + masks)");

  /* ---- inter tasks by class, residual records by size: positions, then a parallel scatter ------- */
  const int ni = (int)n_inter, na = (int)n_intra_total;
  int nt = 0;
  uint32_t refs_used = 0;
  {
    int32_t run_i = 0, run_t = 0;
    for (int k = 0; k < VP9HIP_INTER_CLASSES; ++k) {
      out->inter_class_count[k] = 0;
      for (int t = 0; t < threads; ++t) {
        j->pos_inter[t][k] = run_i;
        run_i += j->hist_inter[t][k];
        out->inter_class_count[k] += j->hist_inter[t][k];
      }
    }
    for (int k = 0; k < 4; ++k) {
      out->txb_size_count[k] = 0;
      for (int t = 0; t < threads; ++t) {
        j->pos_txb[t][k] = run_t;
        run_t += j->hist_txb[t][k];
        out->txb_size_count[k] += j->hist_txb[t][k];
      }
    }
    nt = run_t;
    for (int t = 0; t < threads; ++t) refs_used |= j->refs_used[t];
  }
  pk_parallel(pk->pool, threads, pk_scatter, j);
  out->inter = (const vp9hip_inter_task *)pk->inter_sorted.p;
  out->n_inter = ni;
  out->txb = (const vp9hip_txb *)pk->txb_sorted.p;
  out->n_txb = nt;
  out->refs_used = refs_used;
  if (vec_reserve(&pk->count, sizeof(int32_t) * 16)) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
  int32_t *cnt = (int32_t *)pk->count.p;
  PK_MARK("sort inter + txb");
  /* ---- intra: dependency levels + connected components --------------------------------------- */
  out->intra_decode_order = ia;
  out->n_intra = na;
  if (vec_reserve(&pk->level, sizeof(int32_t) * (size_t)(na + 1)) || vec_reserve(&pk->parent, sizeof(int32_t) * (size_t)(na + 1)) ||
      vec_reserve(&pk->comp_id, sizeof(int32_t) * (size_t)(na + 1)) || vec_reserve(&pk->comp_size, sizeof(int32_t) * (size_t)(na + 2)) ||
      vec_reserve(&pk->order_b, sizeof(int32_t) * (size_t)(na + 1)) ||
      vec_reserve(&pk->intra_isl, sizeof(vp9hip_intra_task) * (size_t)(na + 1)) ||
      vec_reserve(&pk->intra_big, sizeof(vp9hip_intra_task) * (size_t)(na + 1)))
    PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
  int32_t *lv = (int32_t *)pk->level.p, *parent = (int32_t *)pk->parent.p, *comp = (int32_t *)pk->comp_id.p;
  int max_level = 0;
  if (na) {
    /* own_map[p][cell] = base + index of the last intra task that covered the 4x4 cell; values below
     * `base` are leftovers of earlier frames (the maps are not cleared per frame) */
    int32_t *omap[3];
    int reset = pk->own_base <= 0 || pk->own_base > INT32_MAX - (na + 2);
    for (int p = 0; p < 3; ++p) {
      const size_t cells = (size_t)(paw[p] >> 2) * (size_t)(pah[p] >> 2);
      if (cells != pk->own_cells[p]) reset = 1;
      if (vec_reserve(&pk->own_map[p], cells * sizeof(int32_t))) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
      omap[p] = (int32_t *)pk->own_map[p].p;
    }
    if (reset) {
      for (int p = 0; p < 3; ++p) {
        pk->own_cells[p] = (size_t)(paw[p] >> 2) * (size_t)(pah[p] >> 2);
        memset(omap[p], 0, pk->own_cells[p] * sizeof(int32_t));
      }
      pk->own_base = 1;
    }
    const int32_t base = pk->own_base;
    for (int i = 0; i < na; ++i) {
      const vp9hip_intra_task *t = &ia[i];
      const int p = t->plane, W = paw[p] >> 2, H = pah[p] >> 2;
      const int cx = t->x >> 2, cy = t->y >> 2, n = 1 << t->tx_size;
      int32_t *o = omap[p];
      int l = 0, last = -1;
      parent[i] = i;
#define DEP(yy, xx)                                  \
  do {                                               \
    const int32_t v_ = o[(size_t)(yy) * W + (xx)];   \
    if (v_ >= base && v_ - base != last) {           \
      last = v_ - base;                              \
      if (lv[last] > l) l = lv[last];                \
      const int ra_ = uf_find(parent, last), rb_ = uf_find(parent, i); \
      if (ra_ != rb_) parent[rb_] = ra_;             \
    }                                                \
  } while (0)
      if (t->flags & 2) /* left column */
        for (int y = cy; y < cy + n && y < H; ++y) DEP(y, cx - 1);
      if (t->flags & 1) { /* above row; 2*bs only for 4x4 with have_right (vp9_reconintra.c:349-393) */
        const int ext = (n == 1 && (t->flags & 4)) ? 2 : n;
        for (int x = cx; x < cx + ext && x < W; ++x) DEP(cy - 1, x);
        if (t->flags & 2) DEP(cy - 1, cx - 1);
      }
#undef DEP
      ++l;
      lv[i] = l;
      if (l > max_level) max_level = l;
      const int ye = cy + n < H ? cy + n : H, xe = cx + n < W ? cx + n : W;
      for (int y = cy; y < ye; ++y)
        for (int x = cx; x < xe; ++x) o[(size_t)y * W + x] = base + i;
    }
    pk->own_base = base + na + 1;
  }
  out->n_intra_waves = max_level;

  PK_MARK("intra levels + union-find");
  /* dense component ids in order of first appearance, sizes */
  int n_comp = 0;
  int32_t *csize = (int32_t *)pk->comp_size.p;
  {
    int32_t *root_to_id = (int32_t *)pk->order_b.p;
    for (int i = 0; i < na; ++i) root_to_id[i] = -1;
    for (int i = 0; i < na; ++i) {
      const int r = uf_find(parent, i);
      if (root_to_id[r] < 0) {
        root_to_id[r] = n_comp;
        csize[n_comp++] = 0;
      }
      comp[i] = root_to_id[r];
      ++csize[comp[i]];
    }
  }
  /* Small components are walked TOGETHER: an island (= one workgroup of the island kernel, eight block
   * slots wide) is the set of components whose first block lies in the same luma superblock, as long as the
   * set stays under ISLAND_GROUP_TASKS.  A frame of a real stream has thousands of one- or two-block
   * components; a workgroup each filled the GPU with mostly idle lanes (island walk of the bench frame:
   * 5853 workgroups, 190 us of the whole GPU).  Levels are per block, so blocks of different components
   * simply share waves; the per-superblock completion marks work on islands, whatever they contain. */
  comp_box *cbox = NULL;
  if (n_comp) {
    const size_t n_sb_all = (size_t)sb_rows * sb_cols;
    if (vec_reserve(&pk->lvl_map[0], sizeof(int32_t) * (size_t)(n_comp + 1)) || vec_reserve(&pk->lvl_map[1], sizeof(int32_t) * (n_sb_all + 1)) ||
        vec_reserve(&pk->comp_box, sizeof(comp_box) * (size_t)(n_comp + 1)))
      PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
    int32_t *grp = (int32_t *)pk->lvl_map[0].p, *sb_group = (int32_t *)pk->lvl_map[1].p;
    cbox = (comp_box *)pk->comp_box.p;
    for (int c = 0; c < n_comp; ++c) {
      grp[c] = -1;
      comp_box_init(&cbox[c]);
    }
    for (int i = 0; i < na; ++i) {
      const vp9hip_intra_task *t = &ia[i];
      int32_t *b = cbox[comp[i]].box[t->plane];
      const int bsz = 4 << t->tx_size;
      if (t->x < b[0]) b[0] = t->x;
      if (t->y < b[1]) b[1] = t->y;
      if (t->x + bsz > b[2]) b[2] = t->x + bsz;
      if (t->y + bsz > b[3]) b[3] = t->y + bsz;
      if (t->tx_size == 3 && t->eob > 1) ++cbox[comp[i]].n_tx32;
    }
    for (size_t s = 0; s < n_sb_all; ++s) sb_group[s] = -1;
    for (int i = 0; i < na; ++i) {
      const int c = comp[i];
      if (grp[c] >= 0) continue;
      grp[c] = c;
      if (csize[c] > ISLAND_GROUP_TASKS) continue; /* large enough to fill a workgroup on its own */
      const vp9hip_intra_task *t0 = &ia[i];
      const int sc = t0->plane ? ss : 0;
      int q = (t0->y << sc) >> 6, c2 = (t0->x << sc) >> 6;
      if (q > sb_rows - 1) q = sb_rows - 1;
      if (c2 > sb_cols - 1) c2 = sb_cols - 1;
      const int sb = q * sb_cols + c2, g = sb_group[sb];
      if (g < 0) {
        sb_group[sb] = c;
      } else if (csize[g] + csize[c] <= ISLAND_GROUP_TASKS) {
        /* together only while the group's window still fits the LDS of a workgroup: a component that reaches far
         * into the neighbours stays on its own instead of widening the window of everything around it */
        comp_box u;
        comp_box_union(&u, &cbox[g], &cbox[c]);
        if (VP9HIP_ISLAND_FITS(u.box, csize[g] + csize[c], u.n_tx32)) {
          grp[c] = g;
          csize[g] += csize[c];
          csize[c] = 0;
          cbox[g] = u;
        } else if (!VP9HIP_ISLAND_FITS(cbox[g].box, csize[g], cbox[g].n_tx32)) {
          sb_group[sb] = c; /* the superblock's first component is too wide itself: gather the rest around this one */
        }
      } else {
        sb_group[sb] = c; /* as before: a full group is closed, the next one starts here */
      }
    }
    for (int i = 0; i < na; ++i) comp[i] = grp[comp[i]];
  }
  /* split: islands (one workgroup walks them in its LDS: VP9HIP_ISLAND_FITS — groups were only formed while they
   * fit, so this is about single components) / big components (global waves: key frames, large intra areas) */
#define ISLAND_OK(c) (csize[c] <= MAX_ISLAND_TASKS && VP9HIP_ISLAND_FITS(cbox[c].box, csize[c], cbox[c].n_tx32))
  {
    vp9hip_intra_task *isl = (vp9hip_intra_task *)pk->intra_isl.p, *big = (vp9hip_intra_task *)pk->intra_big.p;
    const size_t kmax = (size_t)(max_level > n_comp ? max_level : n_comp) + 2;
    if (vec_reserve(&pk->count, sizeof(int32_t) * kmax) || vec_reserve(&pk->order_a, sizeof(int32_t) * (size_t)(2 * na + 2)) ||
        vec_reserve(&pk->inter, sizeof(vp9hip_intra_task) * (size_t)(na + 1)))
      PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
    cnt = (int32_t *)pk->count.p;
    int32_t *k1 = (int32_t *)pk->order_a.p, *k2 = k1 + na + 1;
    vp9hip_intra_task *tmp = (vp9hip_intra_task *)pk->inter.p; /* scratch: the unsorted inter list is dead */
    int n_isl = 0, n_big = 0;
    /* [0, n_sb]: expected counts; (n_sb, 2 n_sb]: last island seen per superblock (stamp) */
    if (vec_reserve(&pk->rows_expected, sizeof(int32_t) * 2 * ((size_t)sb_rows * sb_cols + 1))) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
    int32_t *rexp = (int32_t *)pk->rows_expected.p;
    int32_t *sb_stamp = rexp + (size_t)sb_rows * sb_cols + 1;
    memset(rexp, 0, sizeof(int32_t) * 2 * ((size_t)sb_rows * sb_cols + 1));
    out->island_sb_expected = rexp;
    /* gather, keeping decode order */
    for (int i = 0; i < na; ++i) {
      if (!ISLAND_OK(comp[i])) {
        big[n_big] = ia[i];
        k2[n_big++] = lv[i] - 1;
      }
    }
    /* big: stable sort by level */
    if (n_big) {
      memcpy(tmp, big, sizeof(vp9hip_intra_task) * (size_t)n_big);
      counting_sort(tmp, big, sizeof(vp9hip_intra_task), k2, n_big, max_level, cnt);
    }
    if (vec_reserve(&pk->big_wave_start, sizeof(int32_t) * (size_t)(max_level + 2)))
      PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
    int32_t *bws = (int32_t *)pk->big_wave_start.p;
    int n_big_waves = 0;
    if (n_big) {
      /* cnt[] after counting_sort holds the END offset of each key; rebuild starts */
      memset(bws, 0, sizeof(int32_t) * (size_t)(max_level + 2));
      for (int i = 0; i < n_big; ++i) ++bws[k2[i] + 1];
      for (int k = 0; k < max_level; ++k) bws[k + 1] += bws[k];
      n_big_waves = max_level;
      while (n_big_waves > 0 && bws[n_big_waves] == bws[n_big_waves - 1]) --n_big_waves;
    } else {
      bws[0] = 0;
    }
    out->intra_big_tasks = big;
    out->n_intra_big_tasks = n_big;
    out->big_wave_start = bws;
    out->n_big_waves = n_big_waves;

    /* islands: sort by (component, level) = stable by level, then stable by component */
    for (int i = 0; i < na; ++i) {
      if (ISLAND_OK(comp[i])) {
        tmp[n_isl] = ia[i];
        k1[n_isl] = lv[i] - 1;
        k2[n_isl] = comp[i];
        ++n_isl;
      }
    }
    if (n_isl) {
      /* sort (task, comp key) pairs by level: carry the comp key along by sorting an index */
      if (vec_reserve(&pk->order_b, sizeof(int32_t) * (size_t)(3 * na + 3))) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
      int32_t *idx = (int32_t *)pk->order_b.p, *idx2 = idx + na + 1, *kk = idx2 + na + 1;
      for (int i = 0; i < n_isl; ++i) idx[i] = i;
      counting_sort(idx, idx2, sizeof(int32_t), k1, n_isl, max_level, cnt);
      for (int i = 0; i < n_isl; ++i) kk[i] = k2[idx2[i]];
      counting_sort(idx2, idx, sizeof(int32_t), kk, n_isl, n_comp, cnt);
      for (int i = 0; i < n_isl; ++i) isl[i] = tmp[idx[i]];
      /* island records + wave offsets */
      if (vec_reserve(&pk->islands, sizeof(vp9hip_intra_island) * (size_t)(n_comp + 1)) ||
          vec_reserve(&pk->wave_off, sizeof(int32_t) * (size_t)(n_isl + n_comp + 2)))
        PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
      vp9hip_intra_island *is = (vp9hip_intra_island *)pk->islands.p;
      int32_t *wo = (int32_t *)pk->wave_off.p;
      int n_is = 0, n_wo = 0, a = 0, n_lds = 0;
      while (a < n_isl) {
        const int c = k2[idx[a]];
        int e = a;
        vp9hip_intra_island *r = &is[n_is++];
        r->task_start = (uint32_t)a;
        r->wave_off_start = (uint32_t)n_wo;
        r->n_waves = 0;
        r->reserved = 0;
        int prev = -1;
        while (e < n_isl && k2[idx[e]] == c) {
          const int l = k1[idx[e]];
          if (l != prev) {
            wo[n_wo++] = e - a;
            ++r->n_waves;
            prev = l;
          }
          ++e;
        }
        wo[n_wo++] = e - a;
        /* luma superblocks the island's samples lie in (all its tasks are in one plane) */
        {
          int ylo = INT_MAX, yhi = 0, xlo = INT_MAX, xhi = 0;
          for (int k = a; k < e; ++k) {
            const vp9hip_intra_task *t = &isl[k];
            const int sc = t->plane ? ss : 0, bsz = 4 << t->tx_size;
            const int y0 = t->y << sc, y1 = (t->y + bsz) << sc, x0 = t->x << sc, x1 = (t->x + bsz) << sc;
            if (y0 < ylo) ylo = y0;
            if (y1 > yhi) yhi = y1;
            if (x0 < xlo) xlo = x0;
            if (x1 > xhi) xhi = x1;
          }
          int rlo = ylo >> 6, rhi = (yhi - 1) >> 6, clo = xlo >> 6, chi = (xhi - 1) >> 6;
          if (rhi > sb_rows - 1) rhi = sb_rows - 1;
          if (chi > sb_cols - 1) chi = sb_cols - 1;
          r->reserved = (uint32_t)rlo | ((uint32_t)rhi << 8) | ((uint32_t)clo << 16) | ((uint32_t)chi << 24);
        }
        /* the LAST task (highest wave, then list order) of the island inside each luma superblock gets
         * bit 0 of `reserved`: when its wave is done the island is done with that superblock, and the
         * island kernel says so to the loop filter (vp9hip_intra_islands_lf); island_sb_expected counts the
         * marks per superblock.  A transform block never straddles superblocks. */
        ++n_lds; /* every island fits: the others went to the global waves */
        for (int k = e - 1; k >= a; --k) {
          vp9hip_intra_task *t = &isl[k];
          const int sc = t->plane ? ss : 0;
          int q = (t->y << sc) >> 6, c2 = (t->x << sc) >> 6;
          if (q > sb_rows - 1) q = sb_rows - 1;
          if (c2 > sb_cols - 1) c2 = sb_cols - 1;
          const int sb = q * sb_cols + c2;
          t->reserved = 0;
          if (sb_stamp[sb] != n_is) { /* n_is = 1-based id of this island */
            sb_stamp[sb] = n_is;
            t->reserved = 1;
            ++rexp[sb];
          }
        }
        a = e;
      }
      qsort(is, (size_t)n_is, sizeof(*is), island_deadline_first);
      if (vec_reserve(&pk->row_pos, sizeof(int32_t) * (size_t)(sb_rows + 1))) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
      int32_t *rpos = (int32_t *)pk->row_pos.p;
      memset(rpos, 0, sizeof(int32_t) * (size_t)(sb_rows + 1));
      for (int k = 0; k < n_is; ++k) {
        const int g = island_group(&is[k]);
        ++rpos[g < sb_rows ? g : sb_rows - 1];
      }
      for (int r = 1; r < sb_rows; ++r) rpos[r] += rpos[r - 1];
      out->island_row_pos = rpos;
      out->n_islands_lds = n_lds;
      out->islands = is;
      out->n_islands = n_is;
      out->island_wave_off = wo;
      out->n_island_wave_off = n_wo;
    } else {
      if (vec_reserve(&pk->wave_off, sizeof(int32_t) * 4)) PK_FAIL(pk, VP9HIP_ENOMEM, "vp9hip_pack_frame: out of memory");
      ((int32_t *)pk->wave_off.p)[0] = 0;
      out->island_wave_off = (const int32_t *)pk->wave_off.p;
      out->n_island_wave_off = 1;
    }
    out->intra_island_tasks = isl;
    out->n_intra_island_tasks = n_isl;
  }

  PK_MARK("islands");
  /* ---- loop-filter masks ---------------------------------------------------------------------- */
  out->sb_rows = sb_rows;
  out->sb_cols = sb_cols;
  if (P->build_lf_masks) out->lfm = (const vp9hip_lfm *)pk->lfm.p; /* built in pass 1 */
  PK_MARK("loop-filter masks");
  return VP9HIP_OK;
}
