/*
 * vp9hip_fe.h — VP9 bitstream front-end of libvp9hip.so (plain C, no GPU calls): from the bytes
 * of one coded frame to what include/vp9hip_decoder.h takes — the decode-order block list, the coefficient
 * slots (compact layout, include/vp9hip_pack.h) with their eob plane, and the frame parameters.
 *
 * It replaces, on the caller's side of the reconstruction path, what the reference runs on the CPU before
 * wrap_cuda_* (SURVEY §3.1-3.2): vp9_receive_compressed_data (libvpx/vp9/decoder/vp9_decoder.c:380-490),
 * vp9_decode_frame / read_uncompressed_header / read_compressed_header / setup_* (vp9_decodeframe.c:3114-3587,
 * 1560-1875), the entropy phase of decode_tiles (:2388-2430: decode_partition :1386, decode_block :1198,
 * detoken_block :919), vp9_read_mode_info (vp9_decodemv.c:809), vp9_decode_block_tokens
 * (vp9_detokenize.c:268), the context derivations of vp9_pred_common.c/.h and vp9_mvref_common.h, and the
 * backward adaptation (vp9_entropy.c:1041-1100, vp9_entropymode.c:340-412, vp9_entropymv.c:162-189) — restated
 * from the format's definition with own structures; the constant tables (csrc/fe/vp9fe_tables.inc) are the
 * format's normative ones.
 *
 * Two deliberate differences from the reference's fork, both the behaviour of stock libvpx / the VP9
 * specification (identical results on every stream either decoder accepts in sync with its encoder):
 *   - an inter block of 8x8 or more whose transform blocks all came out empty counts as skipped for the
 *     contexts of the blocks after it (the fork sets the flag only after the whole frame was parsed,
 *     vp9_decodeframe.c:1195);
 *   - the entropy contexts of 4x4 columns / rows beyond the frame edge read as zero (the fork never sets
 *     xd->max_blocks_wide in detoken_block, so vp9_detokenize.c:255-265 does not clear them).
 */
#ifndef VP9HIP_FE_H_
#define VP9HIP_FE_H_

#include <stddef.h>
#include <stdint.h>

#include "vp9hip_pack.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp9hip_fe vp9hip_fe;

#define VP9HIP_FE_SLOTS 12 /* frame buffers the front-end hands out (8 reference map entries + the new frame + spare) */

typedef struct vp9hip_fe_frame {
  /* show_existing_frame: nothing was decoded, buffer `show_slot` is to be output */
  int32_t show_existing;
  int32_t show_slot;
  /* a decoded frame */
  int32_t show_frame;        /* output it after reconstruction */
  int32_t key_frame, intra_only, error_resilient;
  int32_t new_slot;          /* frame buffer the frame is reconstructed into */
  int32_t ref_slot[3];       /* frame buffers of LAST / GOLDEN / ALTREF (-1: none; intra frames) */
  int32_t refresh_flags;     /* bit i: reference map entry i is replaced by new_slot after the frame */
  int32_t filter_level, sharpness;
  vp9hip_lf_thresh lf_thresh; /* valid when filter_level != 0 */
  vp9hip_frame_params params; /* width, height, subsampling, bit depth, lossless, tile columns, reference sizes */
  const vp9hip_block *blocks; /* decode order (superblock raster order within a tile row) */
  int32_t n_blocks;
  vp9hip_coeff_layout layout; /* eob plane (one int per 4x4 position), compact caller-placed coefficient slots */
  const int32_t *dqcoeff[3];
  /* statistics */
  int64_t coeff_count;        /* coefficients stored (all planes) */
  int32_t tile_cols, tile_rows;
} vp9hip_fe_frame;

/* Environment (read by vp9hip_fe_create): VP9HIP_FE_TRACE — print where a frame's parse time went when the front-end
 * is destroyed; VP9HIP_FE_CHECKSUMS — a checksum over every block's eobs and coefficients in vp9hip_block.reserved2
 * (reserved[0] always holds the block's segment id, reserved[2] its skip flag as parsed): tests/test_fe_blocks.py. */

/* alloc / release (both or neither): where the coefficient arrays come from — the frame driver's page-locked
 * memory (vp9hip_decoder_host_alloc) lets them travel asynchronously.  threads: entropy threads (one tile column
 * each at most); <= 0: number of tile columns, capped at 16. */
int vp9hip_fe_create(vp9hip_fe **out, vp9hip_alloc_fn alloc, vp9hip_free_fn release, void *user, int threads);
/* Coefficient slots as int16 where a frame allows it (default: off, int32 slots as the reference's build keeps them).
 * With on != 0 every frame is parsed into int16 slots (vp9hip_fe_frame.layout.narrow = 1: same offsets, half the bytes
 * for the frame driver to move) while every coefficient is checked; a frame in which one does not fit is parsed again
 * into int32 slots (layout.narrow = 0) — bit-exact either way.  vp9hip_fe_wide_frames: how often that happened.
 * (on > 1, for tests: a coefficient of that magnitude or more already counts as not fitting.) */
void vp9hip_fe_set_narrow_slots(vp9hip_fe *fe, int on);
int vp9hip_fe_wide_frames(const vp9hip_fe *fe);
void vp9hip_fe_destroy(vp9hip_fe *fe);
const char *vp9hip_fe_error(const vp9hip_fe *fe);

/* Parses one frame (one entry of a superframe; see vp9hip_fe_split_superframe).  `out` points into the
 * front-end's arrays and stays valid for this call and the next two (three sets of output arrays in rotation).
 * The reference map and the probability contexts advance as libvpx's do at the end of
 * vp9_receive_compressed_data: the caller reconstructs every frame it is handed, in order.  VP9HIP_OK, VP9HIP_EINVAL (corrupt / unsupported stream) or VP9HIP_ENOMEM. */
int vp9hip_fe_parse(vp9hip_fe *fe, const uint8_t *data, size_t size, vp9hip_fe_frame *out);

/* The superframe index at the end of a packet (vp9_parse_superframe_index, libvpx/vp9/vp9_dx_iface.c): sizes of
 * the frames the packet holds.  Returns the number of frames (1 with sizes[0] = size when there is no index). */
int vp9hip_fe_split_superframe(const uint8_t *data, size_t size, uint32_t sizes[8]);

/* Test hook: the neighbour contexts the parse derives from the blocks above and left of a block (NULL: none) —
 * skip, intra/inter, interpolation filter, transform size (for max_tx), reference mode, compound reference, single
 * reference p1 / p2 (vp9_pred_common.h / .c) — under the given sign biases of LAST / GOLDEN / ALTREF.
 * tests/test_fe_contexts.py holds them against a table from the reference's own functions. */
void vp9hip_fe_debug_contexts(const vp9hip_block *above, const vp9hip_block *left, const int32_t sign_bias[3], int max_tx,
                              int32_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_FE_H_ */
