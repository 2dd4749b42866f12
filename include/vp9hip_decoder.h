/*
 * vp9hip_decoder.h — frame-level driver of libvp9hip.so: what the reference's two wrappers do
 * around their kernels (vpx-master/inter_cuda_kernel.cu:1041-1123 cuda_inter_prediction,
 * vpx-master/intra_cuda_kernel.cu:1306-1385 cuda_intra_prediction), without libvpx types:
 * pack the frame's blocks (vp9hip_pack.h), move lists / coefficients / frames between host and
 * HBM, enqueue the phases in the reference's order (libvpx/vp9/decoder/vp9_decodeframe.c:
 * 2536-2620: inter, intra, loop filter).
 *
 * Unlike the reference nothing is allocated per frame: the decoder object owns a device frame
 * pool, growable device work-list buffers and the packer's host arrays.  Frames can stay in the
 * pool between calls: a caller that keeps its references resident only uploads
 * coefficients and work lists and downloads what it displays.
 *
 * The libvpx-side shim (shim/vp9hip_libvpx_shim.c) maps wrap_cuda_inter_prediction /
 * wrap_cuda_intra_prediction onto these calls.
 */
#ifndef VP9HIP_DECODER_H_
#define VP9HIP_DECODER_H_

#include "vp9hip.h"
#include "vp9hip_pack.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp9hip_decoder vp9hip_decoder;

#define VP9HIP_POOL_SLOTS 16
/* Work-list sets in the decoder's ring: begin_frame packs frame N+1 into the next set
 * (page-locked host arrays) and copies it to the device on a copy stream while the kernels of frame N,
 * which read another set, run on the launch stream. */
#define VP9HIP_RING_SETS 4

int vp9hip_decoder_create(int device, vp9hip_decoder **out);
void vp9hip_decoder_destroy(vp9hip_decoder *dec);
const char *vp9hip_decoder_error(const vp9hip_decoder *dec);
vp9hip_ctx *vp9hip_decoder_ctx(vp9hip_decoder *dec);

/* A frame in host memory: plane[i] points at sample (0,0) (libvpx: y_buffer / u_buffer /
 * v_buffer, through CONVERT_TO_SHORTPTR for high-bitdepth buffers), stride in SAMPLES.  Rows
 * 0..aligned_height-1 and columns 0..aligned_width-1 must be addressable (they are inside a
 * libvpx buffer: aligned size = y_width x y_height, libvpx/vpx_scale/generic/yv12config.c:170-188). */
typedef struct vp9hip_host_frame {
  void *plane[3];
  int32_t stride[3];
  int32_t width, height; /* luma crop size */
  int32_t ss_x, ss_y;
  int32_t bit_depth, hbd;
} vp9hip_host_frame;

/* Make pool slot `slot` a frame of this geometry (contents undefined unless `clear`). */
int vp9hip_decoder_alloc_slot(vp9hip_decoder *dec, int slot, int width, int height, int ss, int bit_depth, int hbd,
                              int clear);
/* Host -> pool slot (allocates / re-shapes the slot as needed) and pool slot -> host.  Synchronous. */
int vp9hip_decoder_upload(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *src);
int vp9hip_decoder_download(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *dst);
/* Pool slot -> host as soon as the run of ring set `ring_set` (vp9hip_decoder_current_set after the frame's
 * begin_frame) is through — not behind whatever was enqueued after it: for callers that keep the next frame's
 * kernels queued while they fetch this one.  Waits for the copy.  Errors of the kernels themselves (a loop-filter
 * row that gave up waiting) are still reported by vp9hip_decoder_sync. */
int vp9hip_decoder_download_after(vp9hip_decoder *dec, int slot, const vp9hip_host_frame *dst, int ring_set);
/* The slot's device descriptor (for callers that use the batched entry points directly). */
int vp9hip_decoder_slot_frame(vp9hip_decoder *dec, int slot, vp9hip_frame *out);

/* Pack the frame's blocks and move the work lists and coefficients to the device.
 *   layout / dqcoeff: the reference's frameBuf (eob planes + per-plane coefficient arrays,
 *   vpx-master/buffers_struct.h:9-15).  dqcoeff NULL: the residual comes from
 *   vp9hip_decoder_set_residual_planes (layout may then be NULL too, with params.assume_coded).
 *   dqcoeff[p] must hold packed.coeff_count[p] entries. */
int vp9hip_decoder_begin_frame(vp9hip_decoder *dec, const vp9hip_frame_params *params, const vp9hip_block *blocks,
                               int n_blocks, const vp9hip_coeff_layout *layout, const int32_t *const dqcoeff[3]);

/* Same, with flags.  VP9HIP_BEGIN_HOST_PERSISTENT: dqcoeff[] point into page-locked memory (e.g.
 * vp9hip_decoder_host_alloc) that stays untouched until the frame has been run and synchronised — the
 * coefficient copy is then asynchronous as well and the call returns as soon as the frame is packed; the device
 * reads such arrays in place (one gather launch on the copy stream for all regions and lists of the frame instead of a
 * copy call per region). */
#define VP9HIP_BEGIN_HOST_PERSISTENT 1
int vp9hip_decoder_begin_frame_ex(vp9hip_decoder *dec, const vp9hip_frame_params *params, const vp9hip_block *blocks,
                                  int n_blocks, const vp9hip_coeff_layout *layout, const int32_t *const dqcoeff[3], int flags);
/* The ring set begin_frame used last (0..VP9HIP_RING_SETS-1), and making an earlier, still-begun set
 * current again (replaying resident work lists: benchmarks, tests). */
int vp9hip_decoder_current_set(const vp9hip_decoder *dec);
int vp9hip_decoder_select_set(vp9hip_decoder *dec, int set);
/* Page-locked host memory for buffers the caller fills between frames (coefficients). */
void *vp9hip_decoder_host_alloc(vp9hip_decoder *dec, size_t bytes);
void vp9hip_decoder_host_free(vp9hip_decoder *dec, void *p);

/* Residual-plane mode — the reference's contract when its CPU phase B is kept
 * (libvpx/vp9/decoder/vp9_decodeframe.c:2443-2486): the inverse transforms were already run on
 * the CPU into int64 planes laid out like the frame (frameBuf.plane_residuals; stride in
 * samples, plane[i] at sample (0,0)).  High-bitdepth frames only (the fork's CPU transforms
 * write int64 residuals only on its highbd path).  Call after begin_frame, before run. */
int vp9hip_decoder_set_residual_planes(vp9hip_decoder *dec, const int64_t *const res[3], const int32_t stride[3]);

#define VP9HIP_PHASE_INTER 1 /* inter prediction + residual of inter blocks */
#define VP9HIP_PHASE_INTRA 2 /* wave-ordered intra prediction + residual */
#define VP9HIP_PHASE_LF 4    /* loop filter (needs params.build_lf_masks or an explicit lfm) */
/* halves of VP9HIP_PHASE_INTER on their own (per-kernel timing): prediction / residual of inter blocks */
#define VP9HIP_PHASE_INTER_PRED 8
#define VP9HIP_PHASE_INTER_RESID 16

/* Enqueue phases for the frame begun last, reading references from pool slots ref_slot[0..2]
 * (LAST, GOLDEN, ALTREF; -1 = unused) and reconstructing into dst_slot.  h_lfm (HOST, sb_rows *
 * sb_cols records, e.g. cm->lf.lfm) overrides the packer's masks; thresh is needed with
 * VP9HIP_PHASE_LF.  Asynchronous; vp9hip_decoder_sync waits. */
int vp9hip_decoder_run(vp9hip_decoder *dec, int phases, const int ref_slot[3], int dst_slot,
                       const vp9hip_lfm *h_lfm, const vp9hip_lf_thresh *thresh);
int vp9hip_decoder_sync(vp9hip_decoder *dec);
/* The event pair that times a run is on by default; a caller that never reads vp9hip_decoder_last_run_ms can
 * switch it off (two packets less in the queue per frame). */
int vp9hip_decoder_set_timing(vp9hip_decoder *dec, int on);
/* GPU milliseconds of the kernels enqueued by the last vp9hip_decoder_run (after a sync). */
int vp9hip_decoder_last_run_ms(vp9hip_decoder *dec, float *ms);
/* Work lists of the frame begun last (host copies owned by the decoder; for tests). */
const vp9hip_packed *vp9hip_decoder_packed(const vp9hip_decoder *dec);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_DECODER_H_ */
