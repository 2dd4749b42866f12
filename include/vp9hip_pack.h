/*
 * vp9hip_pack.h — host-side packers of libvp9hip.so (plain C, no GPU calls): from the decoded
 * mode information of one frame to the work lists of include/vp9hip.h.
 *
 * This is the reference's host logic on the reconstruction path,
 * restated without libvpx types so that it sits behind a C-ABI:
 *
 *   inter tasks     dec_build_inter_predictors_sb / dec_build_inter_predictors
 *                   (libvpx/vp9/decoder/vp9_decodeframe.c:556-800): one task per plane per
 *                   prediction block, per 4x4 for sub-8x8 blocks with average_split_mvs
 *                   (libvpx/vp9/common/vp9_reconinter.c:65-124); scaled references through
 *                   clamp_mv_to_umv_border_sb + vp9_scale_mv (vp9_reconinter.c:90-110,
 *                   vp9_scale.c:17-44); replaces createBuffers of the reference's GPU path
 *                   (vpx-master/inter_cuda_kernel.cu:897-1028)
 *   transform blocks   visit order and frame-edge clipping of detoken_block / inter_decode /
 *                   intra_decode (vp9_decodeframe.c:919-1024, 1026-1071, 1150-1196 ==
 *                   vp9_foreach_transformed_block_in_plane, vp9/common/vp9_blockd.c:37-75),
 *                   uv transform size uv_txsize_lookup (vp9/common/vp9_common_data.c), tx_type
 *                   intra_mode_to_tx_type_lookup (vp9/common/vp9_reconintra.c:24-35)
 *   intra tasks     availability flags of vp9_predict_intra_block (vp9_reconintra.c:404-424)
 *                   incl. the tile-column rule of set_mi_row_col (vp9_onyxc_int.h: left_mi is
 *                   NULL at a tile's first column), dependency waves + islands; replaces
 *                   createBuffersTr / globalCount / frameAnalyz / canDecodeHost
 *                   (vpx-master/intra_cuda_kernel.cu:931-1304)
 *   loop-filter masks   vp9_build_mask + vp9_adjust_mask (vp9/common/vp9_loopfilter.c:1528-1608,
 *                   766-880), level tables vp9_loop_filter_frame_init (:252-295), thresholds
 *                   update_sharpness / vp9_loop_filter_init (:212-250)
 *
 * The libvpx-side shim (shim/vp9hip_libvpx_shim.c) copies MODE_INFO fields into vp9hip_block
 * records and calls these.  Only 4:2:0 and 4:4:4 with equal subsampling are accepted.
 */
#ifndef VP9HIP_PACK_H_
#define VP9HIP_PACK_H_

#include <stddef.h>

#include "vp9hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The subset of MODE_INFO (libvpx/vp9/common/vp9_blockd.h:74-97) the path reads, plus the
 * block's position (ModeInfoBuf.mi_row / mi_col, vpx-master/buffers_struct.h:51-57).
 * Records are in decode order. */
typedef struct vp9hip_block {
  int16_t mi_row, mi_col;  /* 8-pixel units */
  uint8_t sb_type;         /* BLOCK_SIZE: 0 4X4, 1 4X8, 2 8X4, 3 8X8, 4 8X16, 5 16X8, 6 16X16,
                              7 16X32, 8 32X16, 9 32X32, 10 32X64, 11 64X32, 12 64X64 */
  uint8_t tx_size;         /* luma TX_SIZE 0..3 */
  uint8_t skip;            /* mi->skip as parsed: no coefficients were read for the block */
  uint8_t interp_filter;   /* 0 EIGHTTAP, 1 EIGHTTAP_SMOOTH, 2 EIGHTTAP_SHARP, 3 BILINEAR */
  int8_t ref_frame[2];     /* [0]: 0 INTRA_FRAME, 1..3 LAST/GOLDEN/ALTREF; [1] <= 0: no second ref */
  uint8_t mode;            /* intra y mode (blocks >= 8X8), PREDICTION_MODE 0..9 */
  uint8_t uv_mode;
  uint8_t sub_mode[4];     /* intra, sb_type < 8X8: bmi[i].as_mode */
  uint8_t filter_level;    /* get_filter_level(): lvl[segment_id][ref_frame[0]][mode_lf_lut[mode]] */
  uint8_t reserved[3];
  int16_t mv[2][2];        /* [ref][0 row, 1 col], 1/8 luma sample (blocks >= 8X8) */
  int16_t sub_mv[4][2][2]; /* sb_type < 8X8: bmi[i].as_mv[ref] */
  uint8_t reserved2[4];
} vp9hip_block; /* 64 bytes */

typedef struct vp9hip_frame_params {
  int32_t width, height;    /* luma crop size (cm->width / cm->height) */
  int32_t ss_x, ss_y;       /* chroma subsampling: 1,1 or 0,0 */
  int32_t bit_depth, hbd;   /* hbd: 16-bit sample storage (YV12_FLAG_HIGHBITDEPTH) */
  int32_t lossless;         /* xd->lossless: WHT instead of DCT/ADST */
  int32_t log2_tile_cols;   /* cm->log2_tile_cols (intra left availability stops at tile columns) */
  /* reference i = LAST_FRAME + i: luma crop size of the reference frame (scaling when it differs
   * from width/height; vp9_setup_scale_factors_for_frame, vp9_scale.c:46-77).  0 = unused. */
  int32_t ref_width[3], ref_height[3];
  int32_t build_lf_masks;   /* != 0: also build the loop-filter masks from the blocks */
  int32_t assume_coded;     /* != 0 and no eob planes given: every transform block of a non-skip block
                               counts as coded (eob 1) — for the residual-plane mode, where adding the
                               (zero) residual of an uncoded block changes nothing */
  int32_t reserved[2];
} vp9hip_frame_params;

/* Where the coefficients of the frame are, in the reference's layout
 * (frameBuf, vpx-master/buffers_struct.h:9-15, filled by detoken_block):
 *   - per plane one array of concatenated N*N blocks in decode order, one slot for EVERY visited
 *     transform block of every non-skip block (also those with eob 0);
 *   - eob[plane][y * eob_stride[plane] + x] at each transform block's top-left sample. */
/* A used stretch of plane `plane`'s host coefficient array: [start, start + count) coefficients. */
typedef struct vp9hip_coeff_region {
  int32_t plane, reserved;
  int64_t start, count;
} vp9hip_coeff_region;

typedef struct vp9hip_coeff_layout {
  const int32_t *eob[3];
  int32_t eob_stride[3];
  /* 0: eob[plane][y * eob_stride + x] (the reference's frame-strided plane, one int per SAMPLE position);
   * 2: eob[plane][(y >> 2) * eob_stride + (x >> 2)] — one int per 4x4 position, 16 times denser (what the
   * patched detoken_block writes, oracle/patch_decodeframe.py E11) */
  int32_t eob_shift;
  /* Optional (NULL = the sequential layout above).  block_off[3 * i + p]: where block i's coefficient slots of
   * plane p start inside plane p's host array, in coefficients — for callers whose entropy stage fills one
   * region per TILE COLUMN from several threads, so that slots are consecutive per tile, not
   * per frame.  plane_base[p] is then where plane p's host array is mirrored in the device coefficient
   * buffer and `total` the size of that buffer (both in coefficients). */
  const uint32_t *block_off;
  int64_t plane_base[3];
  int64_t total;
  /* with block_off: the stretches of the host arrays that hold this frame's coefficients (what the frame
   * driver copies to the device, each to plane_base[plane] + start) */
  const vp9hip_coeff_region *regions;
  int64_t n_regions;
  /* != 0 (needs block_off): a transform block's slot holds only vp9hip_coeff_extent() coefficients — the rows
   * the reference's own clearing rule says can be non-zero (detoken_block, vp9_decodeframe.c:960-967) — and
   * nothing at eob 0; slots follow each other without gaps (oracle/patch_decodeframe.py E12).  0: every slot
   * has the full N*N coefficients, as the reference writes them. */
  int32_t compact;
  /* != 0 (needs block_off): the slots hold int16, not int32 — same offsets (in coefficients), half the bytes; the
   * arrays are still passed as int32_t pointers.  For callers that have checked every coefficient of the frame
   * (vp9hip_fe does; the reference's build keeps 32-bit coefficients, vpx_dsp/vpx_dsp_common.h:36-37, and its C
   * transforms take them as such, so a frame with a single coefficient outside int16 keeps the wide slots). */
  int32_t narrow;
} vp9hip_coeff_layout;

/* Rows of an N x N coefficient block (N = 4 << tx_size, row-major) that can hold non-zero values, by the rule
 * with which the reference clears its scratch block after use (vp9_decodeframe.c:960-967 and :1009-1016, the
 * same as libvpx's inverse_transform_block_*): eob 1 -> only [0]; DCT_DCT up to 16x16 with eob <= 10 -> the
 * first 4 rows; 32x32 with eob <= 34 -> the first 256 coefficients = 8 rows; otherwise all.  The kernels read
 * exactly these rows (the rest is zero by that rule), the compact layout stores exactly these rows.
 * tx_type: 0 = DCT_DCT.  Kept in step with txfm::coeff_rows() in csrc/txfm_device.h. */
int vp9hip_coeff_rows(int eob, int tx_type, int tx_size);
int vp9hip_coeff_extent(int eob, int tx_type, int tx_size);

typedef struct vp9hip_packed {
  /* inter prediction, sorted into the classes of vp9hip_inter_pred_batch */
  const vp9hip_inter_task *inter;
  int32_t n_inter, inter_class_count[VP9HIP_INTER_CLASSES];
  /* residual of inter blocks, sorted by transform size (vp9hip_idct_add_batch) */
  const vp9hip_txb *txb;
  int32_t n_txb, txb_size_count[4];
  /* intra: islands (vp9hip_intra_pred_islands) + the remainder of very large components as
   * global waves (vp9hip_intra_pred_waves); intra_decode_order is the same set in decode order */
  const vp9hip_intra_task *intra_island_tasks;
  int32_t n_intra_island_tasks;
  /* the islands in the order vp9hip_intra_islands_lf wants: by group g = max(first superblock row - 1, 0),
   * earliest deadline first inside a group.  Every island fits the LDS window of a workgroup
   * (VP9HIP_ISLAND_FITS: small components are gathered into an island only while it does, and a component that
   * does not fit by itself goes to the global waves below); n_islands_lds == n_islands */
  const vp9hip_intra_island *islands;
  int32_t n_islands, n_islands_lds;
  const int32_t *island_wave_off;
  int32_t n_island_wave_off;
  /* islands[i].reserved = the LUMA superblocks the island's samples lie in: first row | last row << 8 |
   * first column << 16 | last column << 24; island_sb_expected[r * sb_cols + c] = number of islands
   * that touch superblock (r, c) — what the loop filter waits for, superblock by superblock, when it
   * runs beside the island walk (vp9hip_intra_islands_lf); island_row_pos[r] (sb_rows entries) = number of
   * islands with group <= r = the islands that go in front of filter row r in that launch's grid */
  const int32_t *island_sb_expected;
  const int32_t *island_row_pos;
  const vp9hip_intra_task *intra_big_tasks;
  int32_t n_intra_big_tasks;
  const int32_t *big_wave_start; /* n_big_waves + 1 entries */
  int32_t n_big_waves;
  const vp9hip_intra_task *intra_decode_order;
  int32_t n_intra;
  int32_t n_intra_waves; /* depth of the whole dependency graph */
  /* coefficient slots: plane p's array starts at coeff_base[p] (in coefficients) of one
   * concatenated buffer of coeff_total entries; records' coeff_off already include it */
  int64_t coeff_base[3], coeff_count[3], coeff_total;
  /* loop filter (only when params.build_lf_masks) */
  const vp9hip_lfm *lfm;
  int32_t sb_rows, sb_cols;
  /* which references (bit i = LAST_FRAME + i) inter tasks read */
  uint32_t refs_used;
} vp9hip_packed;

typedef struct vp9hip_packer vp9hip_packer;

/* A packer owns growable host arrays that are reused from frame to frame (the reference mallocs
 * and frees its lists for every frame, vp9_decodeframe.c:2316-2330, 2625-2634). */
int vp9hip_packer_create(vp9hip_packer **out);
/* Same, with the packer's arrays taken from the caller's allocator (both functions or neither): the
 * frame driver passes page-locked host memory so that the work lists go to the device with
 * asynchronous copies straight from where the packer wrote them. */
typedef void *(*vp9hip_alloc_fn)(void *user, size_t bytes);
typedef void (*vp9hip_free_fn)(void *user, void *ptr);
int vp9hip_packer_create_ex(vp9hip_packer **out, vp9hip_alloc_fn alloc, vp9hip_free_fn release, void *user);
void vp9hip_packer_destroy(vp9hip_packer *pk);
const char *vp9hip_packer_error(const vp9hip_packer *pk);

/* Packs one frame.  `out` points into the packer's arrays and stays valid until the next call.
 * coeffs may be NULL: every eob then reads as 0 (prediction only) unless params.assume_coded.
 * Returns VP9HIP_OK or VP9HIP_EINVAL / VP9HIP_ENOMEM. */
int vp9hip_pack_frame(vp9hip_packer *pk, const vp9hip_frame_params *params, const vp9hip_block *blocks,
                      int n_blocks, const vp9hip_coeff_layout *coeffs, vp9hip_packed *out);

/* vp9_adjust_mask (libvpx/vp9/common/vp9_loopfilter.c:766-880) over a frame's masks: `raw` is what
 * vp9_build_mask accumulated during parsing (cm->lf.lfm as libvpx holds it before loop_filter_rows
 * adjusts each record in place, :1440-1468), `out` what vp9hip_loop_filter_frame takes.  raw == out is
 * allowed. */
int vp9hip_lf_adjust_masks(const vp9hip_lfm *raw, int sb_rows, int sb_cols, int mi_rows, int mi_cols, vp9hip_lfm *out);

/* Loop-filter level table and thresholds of a frame (vp9_loop_filter_frame_init +
 * update_sharpness, vp9_loopfilter.c:212-295).
 *   seg_lvl[s]: INT32_MIN-free encoding: seg_enabled[s] != 0 -> seg_data[s] is the SEG_LVL_ALT_LF
 *   value of segment s; abs_delta != 0 -> absolute, else added to default_lvl.
 *   ref_deltas[4], mode_deltas[2] used when mode_ref_delta_enabled.
 * out_lvl[8][4][2] as loop_filter_info_n.lvl. */
void vp9hip_lf_frame_init(int default_lvl, int sharpness, const int32_t seg_enabled[8], const int32_t seg_data[8],
                          int abs_delta, int mode_ref_delta_enabled, const int8_t ref_deltas[4],
                          const int8_t mode_deltas[2], uint8_t out_lvl[8][4][2], vp9hip_lf_thresh *out_thresh);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_PACK_H_ */
