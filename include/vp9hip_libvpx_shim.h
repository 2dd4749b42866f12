/*
 * vp9hip_libvpx_shim.h — what shim/vp9hip_libvpx_shim.c exports besides the reference's own two
 * symbols.  Include after the libvpx headers (it only forward-declares their structs).
 *
 * The two reference symbols keep the reference's prototypes and are declared by its caller
 * (/root/reference/libvpx/vp9/decoder/vp9_decodeframe.c:2299-2302):
 *   int wrap_cuda_inter_prediction(int n, double *gpu_copy, double *gpu_run, int *size_for_mb,
 *                                  ModeInfoBuf *MiBuf, VP9_COMMON *cm, VP9Decoder *pbi,
 *                                  int tile_rows, int tile_cols, tran_high_t *residuals);
 *   int wrap_cuda_intra_prediction(double *gpu_copy, double *gpu_run, int *size_for_mb,
 *                                  ModeInfoBuf *MiBuf, VP9_COMMON *cm, VP9Decoder *pbi,
 *                                  int tile_rows, int tile_cols, frameBuf *frameBuffer);
 * (/root/reference/vpx-master/cuda_extern_wrap.cpp:5-17).
 */
#ifndef VP9HIP_LIBVPX_SHIM_H_
#define VP9HIP_LIBVPX_SHIM_H_

#ifdef __cplusplus
extern "C" {
#endif

#include <stddef.h>

struct VP9Decoder;
struct VP9Common;
struct frame_buffer; /* frameBuf, vpx-master/buffers_struct.h:9-15 */

/* The eob plane's granularity: 0 (default) = the reference's layout, one int per SAMPLE position
 * (eob_buf[4 * row * stride + 4 * col], vp9_decodeframe.c:969, 1018); 2 = one int per 4x4 position
 * (plane_eob[plane][(y >> 2) * (stride >> 2) + (x >> 2)]: 16 times denser, what patch E11 makes detoken_block
 * write).  Call before the wrappers of the first frame. */
void vp9hip_shim_set_eob_layout(struct VP9Decoder *pbi, int log2_granularity);

/* Memory for what initBuf() (vp9_decodeframe.c:2242-2270) mallocs and frees for every frame: which = 0..2
 * the coefficient array of that plane (frameBuf.dqcoeff[plane]), 3 the eob plane (frameBuf.eob).  Page-
 * locked, owned by the shim, kept (and grown) from frame to frame; valid until the next call with the
 * same `which` for the same decoder.  The caller must not free it.  NULL on failure. */
void *vp9hip_shim_frame_memory(struct VP9Common *cm, int which, size_t bytes);

/* Coefficient mode: call once per frame right after initBuf() (vp9_decodeframe.c:2316), before the
 * entropy loop advances frameBuffer->dqcoeff[].  From then on the wrappers run the inverse
 * transforms on the GPU from frameBuffer->dqcoeff / plane_eob and ignore the residual planes, so the
 * CPU transforms of phase B (vp9_decodeframe.c:2443-2534) can be removed.  NULL returns to the
 * residual-plane mode. */
void vp9hip_shim_attach_frame_buffer(struct VP9Decoder *pbi, const struct frame_buffer *frameBuffer);

/* Phase E on the GPU and references resident in HBM.  When enabled, the intra wrapper
 * also runs the loop filter before it delivers the frame — masks built from the frame's blocks with
 * vp9_build_mask / vp9_adjust_mask semantics and the skip flag stock libvpx filters with (an inter
 * block >= 8x8 without coded coefficients counts as skipped, vp9_decodeframe.c:1195), libvpx's
 * threshold table (cm->lf_info.lfthr) — so the caller must NOT run its CPU loop filter
 * (vp9_decodeframe.c:2585-2620) afterwards; and the frame stays in the device pool under its
 * frame-buffer index, so later frames that reference it need no upload.  Repeating the call with the
 * same value is free (a caller may issue it for every frame). */
void vp9hip_shim_set_gpu_loop_filter(struct VP9Decoder *pbi, int enable);

/* Tile-parallel entropy stage (oracle/patch_decodeframe.py --mt, E10).
 *   vp9hip_shim_run_parallel: fn(arg, 0) .. fn(arg, n - 1) on the shim's persistent thread pool (index 0 on
 *     the calling thread); returns when all have returned.  VP9HIP_SHIM_THREADS caps the pool (default: n).
 *   vp9hip_shim_block_off_buffer: a buffer of 3 * n_blocks uint32 the caller fills with, for every block of the
 *     canonical (decode-order) list, the offset of its coefficient slots inside each plane's array.
 *   vp9hip_shim_set_tile_layout: the stretches of the coefficient arrays the threads filled — region t, plane p:
 *     [start[3 * t + p], start[3 * t + p] + count[3 * t + p]) coefficients.  Valid for the frame being decoded;
 *     the next wrap_cuda_* call consumes it.  flags: VP9HIP_SHIM_COEFF_COMPACT — every transform block's slot
 *     holds only vp9hip_coeff_extent(eob, tx_type, tx_size) coefficients (vp9hip_pack.h; the rows the reference's
 *     own clearing rule leaves non-zero, nothing at eob 0), slots without gaps (E12): the entropy threads copy a
 *     fraction of the bytes and so does the upload. */
#define VP9HIP_SHIM_COEFF_COMPACT 1
void vp9hip_shim_run_parallel(struct VP9Decoder *pbi, int n, void (*fn)(void *arg, int index), void *arg);
#include <stdint.h>
#include "vp9hip_pack.h" /* vp9hip_coeff_extent */
uint32_t *vp9hip_shim_block_off_buffer(struct VP9Decoder *pbi, int n_blocks);
void vp9hip_shim_set_tile_layout(struct VP9Decoder *pbi, int n_blocks, int n_regions, const int64_t *start, const int64_t *count,
                                 int flags);

/* Measurement aid (VP9HIP_SHIM_TRACE=1; a no-op otherwise): marks 0..4 placed in decode_tiles — entry, before
 * the entropy loop, after it, after the two entry points, before return — split a frame's host time into
 * set-up / entropy decode / reconstruction entry points / tear-down in the trace printed at exit. */
void vp9hip_shim_mark(struct VP9Decoder *pbi, int mark);

/* Frees the GPU state kept for a decoder instance (call from vp9_decoder_remove). */
void vp9hip_shim_release(struct VP9Decoder *pbi);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_LIBVPX_SHIM_H_ */
