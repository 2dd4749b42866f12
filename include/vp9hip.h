/*
 * vp9hip.h — C-ABI of libvp9hip.so: the MI355X (gfx950) VP9 block-reconstruction
 * path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Three layers, outermost first (see INTEGRATION.md for the libvpx-side stubs):
 *
 *  (1) Frame-level reference ABI — the two symbols the reference's decoder calls
 *      (/root/reference/vpx-master/cuda_extern_wrap.cpp:5-17, declared by the caller at
 *      libvpx/vp9/decoder/vp9_decodeframe.c:2299-2302, called at :2546 and :2564):
 *        wrap_cuda_inter_prediction(), wrap_cuda_intra_prediction()
 *      They are thin adapters over layer (2) and need libvpx's struct layouts, so they
 *      live in include/vp9hip_libvpx_shim.h / shim/ and are compiled inside the libvpx tree.
 *
 *  (2) Batched entry points — what a frame-level caller uses: packed work lists in, pixels
 *      out, everything device-resident (vp9hip_*_batch, vp9hip_frame_*).  These are the
 *      hot path and what bench.py measures.
 *
 *  (3) Block-level `_hip` twins of the vpx_dsp_rtcd / vp9_rtcd prototypes (host pointers,
 *      one block per call) — include/vp9hip_rtcd.h.  Same signatures as the reference's
 *      `_c` functions so they can be assigned to the rtcd function pointers
 *      (vpx-master/vpx_dsp_rtcd.h:40, setup_rtcd_internal :2074).  Meant for parity tests
 *      and bring-up, not for speed.
 *
 * Error model: every int-returning function returns 0 on success and a negative
 * VP9HIP_E* code otherwise; vp9hip_last_error() gives the text.  Nothing falls back to a
 * CPU path: without a usable HIP device the calls fail.
 */
#ifndef VP9HIP_H_
#define VP9HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP9HIP_ABI_VERSION 2

enum {
  VP9HIP_OK = 0,
  VP9HIP_EINVAL = -1,  /* bad argument (shape/size the kernels do not accept) */
  VP9HIP_EDEVICE = -2, /* HIP runtime error; see vp9hip_last_error() */
  VP9HIP_ENOMEM = -3
};

typedef struct vp9hip_ctx vp9hip_ctx;

/* One context per decoder instance / per GPU.  Owns a HIP stream and scratch memory. */
int vp9hip_create(int device, vp9hip_ctx **out);
void vp9hip_destroy(vp9hip_ctx *ctx);
const char *vp9hip_last_error(const vp9hip_ctx *ctx);
int vp9hip_abi_version(void);
/* Width of the slots of the d_coeffs arrays handed to the calls that follow (vp9hip_idct_add_batch, the intra calls):
 * 32 (default) = int32, the reference build's tran_low_t; 16 = int16 — same offsets, half the bytes, for frames whose
 * dequantised coefficients all fit (the bitstream front-end checks every one and falls back to 32, vp9hip_fe.h).
 * The arithmetic is the same either way: a slot is sign-extended to the int the transforms start from. */
int vp9hip_set_coeff_bits(vp9hip_ctx *ctx, int bits);
/* hipStream_t of the context, as an opaque pointer (for event timing by the caller). */
void *vp9hip_stream(vp9hip_ctx *ctx);
int vp9hip_sync(vp9hip_ctx *ctx);

/* GPU timing on the context's stream (hipEvent pairs).  slot in [0, VP9HIP_TIMER_SLOTS):
 * vp9hip_timer_begin/end record events around whatever the caller enqueues in between;
 * vp9hip_timer_read waits for the end event and ACCUMULATES nothing — it returns the elapsed
 * milliseconds of that one begin/end pair.  Replaces the reference's clock()/cudaEvent pairs
 * reported through *gpu_copy / *gpu_run (vpx-master/inter_cuda_kernel.cu:1069-1101). */
#define VP9HIP_TIMER_SLOTS 4096
int vp9hip_timer_begin(vp9hip_ctx *ctx, int slot);
int vp9hip_timer_end(vp9hip_ctx *ctx, int slot);
int vp9hip_timer_read(vp9hip_ctx *ctx, int slot, float *ms);

/* Device memory plumbing for callers that do not bring their own allocator. */
void *vp9hip_malloc(vp9hip_ctx *ctx, size_t bytes);
void vp9hip_free(vp9hip_ctx *ctx, void *dptr);
int vp9hip_memcpy_h2d(vp9hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int vp9hip_memcpy_d2h(vp9hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int vp9hip_memset(vp9hip_ctx *ctx, void *dst, int value, size_t bytes);

/* ------------------------------------------------------------------------------------------
 * Frame descriptor: three planes resident in HBM.  No border is required: every kernel that
 * reads outside a plane clamps coordinates, which is what libvpx's decoder border emulation
 * (vp9_decodeframe.c:432-690, build_mc_border/extend_and_predict) computes.
 *   width/height      crop size of the plane in pixels (y_crop_width / uv_crop_width)
 *   awidth/aheight    8-aligned coded size (y_width / uv_width, yv12config.c:170-188): pixels
 *                     are stored for the whole aligned area
 *   stride            in SAMPLES (not bytes)
 *   bit_depth 8 -> uint8 samples unless hbd != 0; hbd -> uint16 samples (bd 8, 10 or 12)
 * ---------------------------------------------------------------------------------------- */
typedef struct vp9hip_frame {
  void *plane[3];
  int32_t stride[3];
  int32_t width[3], height[3];
  int32_t awidth[3], aheight[3];
  int32_t bit_depth;
  int32_t hbd;
} vp9hip_frame;

/* ------------------------------------------------------------------------------------------
 * (a1–a3) inverse transform + add.  One record per coded transform block.
 * Replaces the per-block calls of inverse_transform_block_{inter,intra}
 * (vp9_decodeframe.c:173-288) -> vp9_idct*_add / vp9_iht*_add / vp9_iwht4x4_add
 * (vp9/common/vp9_idct.c:119-204, highbd :308-396).
 * ---------------------------------------------------------------------------------------- */
typedef struct vp9hip_txb {
  uint32_t coeff_off; /* index of this block's first coefficient in the coefficient buffer (a multiple of 4: slots hold whole rows) */
  uint16_t x, y;      /* top-left pixel of the block inside its plane */
  uint8_t plane;      /* 0..2 */
  uint8_t tx_size;    /* 0: 4x4, 1: 8x8, 2: 16x16, 3: 32x32 */
  uint8_t tx_type;    /* 0 DCT_DCT, 1 ADST_DCT, 2 DCT_ADST, 3 ADST_ADST; bit 7: lossless (WHT) */
  uint8_t reserved;
  uint16_t eob;       /* as passed to vp9_idctNxN_add(): selects the DC-only shortcut */
  uint16_t reserved2;
} vp9hip_txb; /* 16 bytes */

/* coeffs: dequantised tran_low_t (int32), N*N per block, raster order, as written by
 * vp9_decode_block_tokens into dqcoeff (vp9_decodeframe.c:958).  d_blocks / d_coeffs are
 * DEVICE pointers.  The records are grouped by transform size: size_count[0] 4x4 records
 * first, then size_count[1] 8x8, size_count[2] 16x16, size_count[3] 32x32 (HOST array).
 * Blocks must not overlap.  Asynchronous on the context's stream. */
int vp9hip_idct_add_batch(vp9hip_ctx *ctx, const vp9hip_txb *d_blocks, const int32_t size_count[4],
                          const int32_t *d_coeffs, const vp9hip_frame *frame);

/* ------------------------------------------------------------------------------------------
 * (a5–a7) inter prediction: 8-tap sub-pel convolve with border clamping, single or compound,
 * unscaled or scaled references.  One record per prediction block per plane (sub-8x8 luma:
 * one per 4x4).  The host packer has already applied average_split_mvs / clamp_mv_to_umv_
 * border_sb / vp9_scale_mv (vp9_reconinter.c:90-124, vp9_scale.c:37-44):
 *   pos_x/pos_y = 16 * integer sample position + sub-pel phase of the block's top-left sample
 *                 in reference plane coordinates (may be negative / beyond the plane)
 *   step_x/step_y = 16 for unscaled references (sf->x_step_q4)
 * Replaces dec_build_inter_predictors (vp9_decodeframe.c:563-690) and the reference's
 * cuda_inter_4x4_both (vpx-master/inter_cuda_kernel.cu:869).
 * ---------------------------------------------------------------------------------------- */
typedef struct vp9hip_inter_task {
  int16_t dst_x, dst_y; /* top-left in the destination plane */
  uint8_t w, h;         /* 4..64 */
  uint8_t plane;
  uint8_t flags;        /* bit0: compound (second reference averaged in); bits1-3: INTERP_FILTER
                           0 EIGHTTAP 1 SMOOTH 2 SHARP 3 BILINEAR */
  int32_t pos_x[2], pos_y[2];
  uint8_t ref[2];       /* index into the refs[] array passed to the call */
  uint8_t step_x[2], step_y[2];
  uint8_t reserved[2];
} vp9hip_inter_task; /* 32 bytes */

#define VP9HIP_MAX_REFS 8
/* d_tasks (DEVICE) is grouped into VP9HIP_INTER_CLASSES classes, class_count[] (HOST) giving their sizes in order:
 *   0..12  unscaled (step 16) tasks of exactly one of VP9's block shapes, in this order (vp9hip_inter_class):
 *          4x4 4x8 | 8x4 8x8 8x16 | 16x8 16x16 16x32 | 32x16 32x32 32x64 | 64x32 64x64   (width x height)
 *          -> the fast kernels: every lane owns four output columns of a strip of 4 (the two ..x4 shapes) or 8 rows
 *          and a task takes exactly the lanes its shape needs (8-bit samples: registers only, dot4; 16-bit: LDS
 *          tiles, dot2)
 *   13     everything else (scaled references, other shapes)      -> generic kernel
 * Tasks must not overlap in the destination.  Asynchronous on the context's stream. */
#define VP9HIP_INTER_CLASSES 14
int vp9hip_inter_class(int w, int h, int unscaled); /* the class of a task: host code, no GPU */
int vp9hip_inter_pred_batch(vp9hip_ctx *ctx, const vp9hip_inter_task *d_tasks,
                            const int32_t class_count[VP9HIP_INTER_CLASSES], const vp9hip_frame *refs, int n_refs,
                            const vp9hip_frame *dst);

/* ------------------------------------------------------------------------------------------
 * (a8–a10) intra prediction, dependency-wave ordered.  One record per transform block
 * (prediction happens per transform block: vp9_predict_intra_block, vp9_reconintra.c:404).
 * The host sorts records by wave (level = 1 + max level of the left / above / above-left /
 * above-right neighbours it reads) and passes wave_start[n_waves+1].
 * If d_coeffs != NULL the residual is added in the same pass (records carry coeff_off/eob/
 * tx_type exactly like vp9hip_txb; eob == 0 -> prediction only).
 * Replaces the reference's intra_traditional waves (vpx-master/intra_cuda_kernel.cu:901,
 * 1291-1358).
 * ---------------------------------------------------------------------------------------- */
typedef struct vp9hip_intra_task {
  uint32_t coeff_off;
  uint16_t x, y;
  uint8_t plane;
  uint8_t tx_size;
  uint8_t tx_type; /* bit 7: lossless; bit 6: the N*N entries at coeff_off are the residual itself
                      (raster, int32), no transform — residual-plane mode of vp9hip_decoder.h */
  uint8_t mode;    /* 0 DC 1 V 2 H 3 D45 4 D135 5 D117 6 D153 7 D207 8 D63 9 TM */
  uint16_t eob;
  uint8_t flags;   /* bit0 have_top, bit1 have_left, bit2 have_right, bit3 raw edges: read all 2*bs
                      above samples from the frame instead of replicating (rtcd twins only) */
  uint8_t reserved; /* bit0: last task of its island inside its luma superblock (vp9hip_intra_islands_lf) */
} vp9hip_intra_task; /* 16 bytes */

int vp9hip_intra_pred_waves(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                            const int32_t *wave_start /* HOST, n_waves+1 entries */, int n_waves,
                            const int32_t *d_coeffs, const vp9hip_frame *frame);

/* Island form of the same thing, ONE launch: an island is a connected component of the intra
 * dependency graph (in an inter frame: an isolated intra block or a cluster of them; inter
 * blocks are already reconstructed and cut the graph), or several small components walked together.
 * Records of an island are contiguous in d_tasks starting at task_start, sorted by wave;
 * d_wave_off[wave_off_start + w] is the offset (relative to task_start) of its wave w, with one extra
 * end entry.  Islands are independent, one workgroup walks one island.  Use vp9hip_intra_pred_waves
 * for very large islands (key frames).
 *
 * An island whose sample window fits the workgroup's LDS is walked there: the window (per plane the
 * bounding box of its blocks, one row above, one column to the left, four columns to the right) is read
 * once, the residual of every coded block goes into the window as int16 (saturated: clip(pred + res) is
 * the same), the waves then read edges from and write predictions to LDS only, and the island's blocks
 * are written to the frame once at the end.  VP9HIP_ISLAND_FITS says which islands qualify. */
typedef struct vp9hip_intra_island {
  uint32_t task_start;
  uint32_t wave_off_start;
  uint32_t n_waves;
  uint32_t reserved; /* the LUMA superblocks the island's samples lie in: first row | last row << 8 | first
                        column << 16 | last column << 24 (needed by vp9hip_intra_islands_lf only) */
} vp9hip_intra_island; /* 16 bytes */
/* LDS window of an island: int16 elements in all, tasks cached in LDS, full 32x32 transforms per island */
#define VP9HIP_ISLAND_TILE_ELEMS 20480
#define VP9HIP_ISLAND_MAX_TASKS 768
#define VP9HIP_ISLAND_MAX_TX32 16
/* row pitch (elements) of a plane window whose blocks span w samples: w + 1 (left) + 4 (above-right of a 4x4
 * block), rounded up to an odd number of dwords */
#define VP9HIP_ISLAND_PITCH(w) (((((w) + 6) & ~1) & 2) ? (((w) + 6) & ~1) : ((((w) + 6) & ~1) + 2))
/* box[p] = { x0, y0, x1, y1 } of the island's blocks in plane p (x1 <= x0: no block in that plane) */
#define VP9HIP_ISLAND_PLANE_ELEMS(b) ((b)[2] > (b)[0] ? VP9HIP_ISLAND_PITCH((b)[2] - (b)[0]) * ((b)[3] - (b)[1] + 1) : 0)
#define VP9HIP_ISLAND_FITS(box, n_tasks, n_tx32)                                                                        \
  ((n_tasks) <= VP9HIP_ISLAND_MAX_TASKS && (n_tx32) <= VP9HIP_ISLAND_MAX_TX32 &&                                        \
   VP9HIP_ISLAND_PLANE_ELEMS((box)[0]) + VP9HIP_ISLAND_PLANE_ELEMS((box)[1]) + VP9HIP_ISLAND_PLANE_ELEMS((box)[2]) <= \
       VP9HIP_ISLAND_TILE_ELEMS)
/* Any islands: those that fit are walked in LDS, the others through the frame in memory. */
int vp9hip_intra_pred_islands(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks,
                              const vp9hip_intra_island *d_islands, int n_islands,
                              const int32_t *d_wave_off /* DEVICE */, const int32_t *d_coeffs,
                              const vp9hip_frame *frame);

/* ------------------------------------------------------------------------------------------
 * (a11–a12) loop filter of a whole frame.  Per 64x64 superblock one vp9hip_lfm record (the
 * content of libvpx's LOOP_FILTER_MASK, vp9_loopfilter.h:60-68, built by vp9_build_mask /
 * vp9_adjust_mask) and the per-level threshold table (loop_filter_info_n.lfthr,
 * vp9_loopfilter.h:53-58).  The filter order is libvpx's: superblocks in raster order, per
 * superblock and plane all vertical edges then all horizontal ones (vp9_loopfilter.c:1424-
 * 1469, 1241-1422); on the GPU one workgroup walks each superblock row, rows pipelined behind
 * each other, which keeps exactly that order where pixels overlap.
 * ---------------------------------------------------------------------------------------- */
typedef struct vp9hip_lfm {
  uint64_t left_y[4];  /* per TX_SIZE */
  uint64_t above_y[4];
  uint64_t int_4x4_y;
  uint16_t left_uv[4];
  uint16_t above_uv[4];
  uint16_t int_4x4_uv;
  uint8_t lfl_y[64];
  uint8_t reserved[6];
} vp9hip_lfm; /* 160 bytes */

typedef struct vp9hip_lf_thresh {
  uint8_t mblim[64], lim[64], hev_thr[64]; /* indexed by filter level 0..63 */
} vp9hip_lf_thresh;

int vp9hip_loop_filter_frame(vp9hip_ctx *ctx, const vp9hip_lfm *d_lfm, int sb_rows, int sb_cols,
                             const vp9hip_lf_thresh *h_thresh, const vp9hip_frame *frame,
                             int planes /* 1: Y only, 3: Y,U,V */);

/* The island walk and the loop filter of the same frame as ONE launch (walk_lf_kernel): a workgroup per island
 * and a workgroup per (superblock row, plane) of the filter.  The filter takes superblock (r, c) once the islands
 * touching superblocks (r..r+1, c-1..c+1) are done (an unfinished island there would still read samples the
 * filter changes), not when the whole walk is.  The hand-over is per (island, superblock): bit 0 of
 * vp9hip_intra_task.reserved marks the LAST task of its island inside a luma superblock (list order = wave
 * order), and d_sb_expected (DEVICE, sb_rows * sb_cols entries) = number of marked tasks per superblock
 * (vp9hip_pack.h fills both).  Every island must fit the LDS window (VP9HIP_ISLAND_FITS; vp9hip_pack.h only builds
 * such islands and sends larger components to the global waves); one that does not is reported by vp9hip_sync.
 *
 * Forward progress by construction: a workgroup's place in the launch's order is a ticket it draws when it starts
 * running (an atomic counter), so whoever holds a lower ticket is running or done, whatever order the hardware
 * starts workgroups in; and a workgroup only ever waits for lower tickets — islands wait for nothing, a filter row
 * waits for the row above and for islands in front of it.  Whatever else shares the GPU (other contexts, other
 * processes), every wait ends: nobody waits for a workgroup that is not resident.  h_row_pos (HOST, sb_rows
 * entries, non-decreasing, or NULL) says where the rows sit in that order: h_row_pos[r] = number of islands in front
 * of the workgroups of filter row r, which must include every island that touches superblock rows <= r + 1
 * (vp9hip_pack.h sorts the islands accordingly); NULL puts all islands first.  Rows start as early as their islands
 * allow instead of behind the whole walk.
 *
 * Frames with very large components (key frames: the vp9hip_intra_pred_waves remainder) use the calls in
 * sequence instead.  Ordered after everything enqueued before on the context, and later work is ordered after it.
 * At most 128 superblocks in either direction (8192 x 8192 samples); larger frames are refused here — the frame
 * driver (vp9hip_decoder_run) then runs the phases in sequence. */
int vp9hip_intra_islands_lf(vp9hip_ctx *ctx, const vp9hip_intra_task *d_tasks, const vp9hip_intra_island *d_islands,
                            int n_islands, const int32_t *d_wave_off, const int32_t *d_coeffs,
                            const int32_t *d_sb_expected, const int32_t *h_row_pos, const vp9hip_lfm *d_lfm, int sb_rows,
                            int sb_cols, const vp9hip_lf_thresh *h_thresh, const vp9hip_frame *frame, int planes);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_H_ */
