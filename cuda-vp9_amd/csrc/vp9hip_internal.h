// vp9hip_internal.h — shared by the translation units of libvp9hip.so (not installed).
#ifndef VP9HIP_INTERNAL_H_
#define VP9HIP_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vp9hip.h"

struct vp9hip_ctx {
  int device;
  hipStream_t stream;
  char err[512];
  // scratch owned by the context (grown on demand, never shrunk)
  void *scratch;
  size_t scratch_bytes;
  int cu_count;
  int *lf_err_flag;  // device flag (own allocation): a loop-filter wait timed out; set until vp9hip_sync reports it
  bool lf_err_armed; // a loop filter was launched since the flag was last read
  void *lf_hand;     // hand-off granules of the row-walking loop filter (lf_kernels.hip)
  size_t lf_hand_bytes;
  unsigned lf_gen;   // generation number of the last launch
  void *d_taps;  // packed i8 convolve taps (inter fast path)
  hipEvent_t *ev_begin, *ev_end;  // VP9HIP_TIMER_SLOTS each, created lazily
  // overlap of the intra island walk with the loop filter (vp9hip_intra_islands_lf)
  hipStream_t stream2;
  hipEvent_t ev_fork, ev_join;
  // residual of the intra island tasks, computed ahead of the walk (intra_kernels.hip): int32 per sample
  void *resid;
  size_t resid_bytes;
  // vp9hip_intra_residual_begin: the pre-pass already runs on stream2 for these lists; the walk waits for it
  hipEvent_t ev_resid_start, ev_resid_done;
  const void *resid_tasks, *resid_coeffs;
  int lf_zeroed_rows, lf_zeroed_cols;  // vp9hip_intra_residual_begin also zero-filled the filter / island counters
};

#define VP9HIP_FAIL(ctx, code, ...)                          \
  do {                                                       \
    if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
    return (code);                                           \
  } while (0)

#define VP9HIP_CHECK(ctx, expr)                                                        \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      VP9HIP_FAIL(ctx, VP9HIP_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

// Device-side view of a frame, passed to kernels by value.
struct FrameDev {
  void *plane[3];
  int stride[3];
  int width[3], height[3];
  int awidth[3], aheight[3];
  int bit_depth;
};

static inline FrameDev to_dev(const vp9hip_frame *f) {
  FrameDev d;
  for (int i = 0; i < 3; ++i) {
    d.plane[i] = f->plane[i];
    d.stride[i] = f->stride[i];
    d.width[i] = f->width[i];
    d.height[i] = f->height[i];
    d.awidth[i] = f->awidth[i];
    d.aheight[i] = f->aheight[i];
  }
  d.bit_depth = f->bit_depth;
  return d;
}

static inline int frame_ok(const vp9hip_frame *f) {
  if (!f) return 0;
  if (f->bit_depth != 8 && f->bit_depth != 10 && f->bit_depth != 12) return 0;
  if (f->bit_depth != 8 && !f->hbd) return 0;
  for (int i = 0; i < 3; ++i) {
    if (!f->plane[i]) continue;
    if (f->stride[i] < f->awidth[i] || f->awidth[i] < f->width[i] || f->aheight[i] < f->height[i] ||
        f->width[i] <= 0 || f->height[i] <= 0)
      return 0;
  }
  return f->plane[0] != NULL;
}

int vp9hip_ensure_scratch(vp9hip_ctx *ctx, size_t bytes);
int vp9hip_live_contexts(void);  /* contexts alive in this process */
int vp9hip_ensure_resid(vp9hip_ctx *ctx, const vp9hip_frame *frame);
int vp9hip_lf_zero_counters(vp9hip_ctx *ctx, const vp9hip_frame *frame, hipStream_t st);
int vp9hip_islands_prepare(vp9hip_ctx *ctx, hipStream_t st, const vp9hip_intra_task *d_tasks,
                           const vp9hip_intra_island *d_islands, int n_islands, const int32_t *d_wave_off,
                           const int32_t *d_coeffs, const vp9hip_frame *frame);
int vp9hip_islands_launch(vp9hip_ctx *ctx, hipStream_t st, const vp9hip_intra_task *d_tasks,
                          const vp9hip_intra_island *d_islands, int n_islands, const int32_t *d_wave_off,
                          const int32_t *d_coeffs, const vp9hip_frame *frame, int *d_sb_done, int sb_cols);

#endif
