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
  int coeff16;  // the d_coeffs arrays of the launches hold int16 slots (vp9hip_set_coeff_bits)
  int *lf_err_flag;  // device flag (own allocation): a loop-filter wait timed out; set until vp9hip_sync reports it
  bool lf_err_armed; // a loop filter was launched since the flag was last read
  void *lf_hand;     // hand-off granules of the row-walking loop filter (lf_kernels.hip)
  size_t lf_hand_bytes;
  unsigned lf_gen;   // generation number of the last launch
  void *d_taps;  // packed i8 convolve taps (inter fast path)
  hipEvent_t *ev_begin, *ev_end;  // VP9HIP_TIMER_SLOTS each, created lazily
  // island counters of the fused walk + filter launch (lf_kernels.hip): two sets in `scratch`, used in turn
  int gate_n, gate_parity;
  int *lf_ticket;  // two per-launch ticket counters, used in turn (lf_kernels.hip: draw_ticket)
  int lf_ticket_parity;
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

#endif
