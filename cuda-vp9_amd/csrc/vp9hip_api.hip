// vp9hip_api.hip — context, memory plumbing (C-style host code, C-ABI exports).
#include "vp9hip_internal.h"

#include <stdlib.h>

static char g_err[512] = "no context";

extern "C" int vp9hip_abi_version(void) { return VP9HIP_ABI_VERSION; }

extern "C" int vp9hip_create(int device, vp9hip_ctx **out) {
  if (!out) return VP9HIP_EINVAL;
  *out = NULL;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    snprintf(g_err, sizeof(g_err), "no HIP device available (%s)", hipGetErrorString(e));
    return VP9HIP_EDEVICE;
  }
  if (device < 0 || device >= n) {
    snprintf(g_err, sizeof(g_err), "device %d out of range (0..%d)", device, n - 1);
    return VP9HIP_EINVAL;
  }
  vp9hip_ctx *ctx = (vp9hip_ctx *)calloc(1, sizeof(*ctx));
  if (!ctx) return VP9HIP_ENOMEM;
  ctx->device = device;
  if ((e = hipSetDevice(device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "context creation failed: %s", hipGetErrorString(e));
    free(ctx);
    return VP9HIP_EDEVICE;
  }
  hipDeviceProp_t prop;
  ctx->cu_count = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
  ctx->err[0] = 0;
  *out = ctx;
  return VP9HIP_OK;
}

extern "C" void vp9hip_destroy(vp9hip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->lf_err_flag) (void)hipFree(ctx->lf_err_flag);
  if (ctx->lf_hand) (void)hipFree(ctx->lf_hand);
  if (ctx->lf_ticket) (void)hipFree(ctx->lf_ticket);
  if (ctx->d_taps) (void)hipFree(ctx->d_taps);
  if (ctx->ev_begin) {
    for (int i = 0; i < VP9HIP_TIMER_SLOTS; ++i) {
      if (ctx->ev_begin[i]) (void)hipEventDestroy(ctx->ev_begin[i]);
      if (ctx->ev_end[i]) (void)hipEventDestroy(ctx->ev_end[i]);
    }
    free(ctx->ev_begin);
    free(ctx->ev_end);
  }
  (void)hipStreamDestroy(ctx->stream);
  free(ctx);
}

extern "C" const char *vp9hip_last_error(const vp9hip_ctx *ctx) { return ctx ? ctx->err : g_err; }

extern "C" int vp9hip_set_coeff_bits(vp9hip_ctx *ctx, int bits) {
  if (!ctx) return VP9HIP_EINVAL;
  if (bits != 16 && bits != 32) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_set_coeff_bits: %d (16 or 32)", bits);
  ctx->coeff16 = bits == 16;
  return VP9HIP_OK;
}

extern "C" void *vp9hip_stream(vp9hip_ctx *ctx) { return ctx ? (void *)ctx->stream : NULL; }

extern "C" int vp9hip_sync(vp9hip_ctx *ctx) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->lf_err_flag && ctx->lf_err_armed) {
    int flag[8] = { 0 };
    VP9HIP_CHECK(ctx, hipMemcpy(flag, ctx->lf_err_flag, sizeof(flag), hipMemcpyDeviceToHost));
    ctx->lf_err_armed = false;
    if (flag[0]) {
      VP9HIP_CHECK(ctx, hipMemsetAsync(ctx->lf_err_flag, 0, sizeof(flag), ctx->stream));  // reported once
      VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (flag[0] & 2)
        VP9HIP_FAIL(ctx, VP9HIP_EINVAL,
                    "vp9hip_intra_islands_lf: an island does not fit the LDS window (VP9HIP_ISLAND_FITS); the frames "
                    "enqueued since the last synchronisation are not valid");
      static const char *const what[4] = { "?", "the intra islands of a superblock", "the rows handed down by the row above",
                                           "its own filtering wave" };
      VP9HIP_FAIL(ctx, VP9HIP_EDEVICE,
                  "loop filter: a superblock row gave up waiting (for the row above or for the intra islands around it): plane %d row %d "
                  "at superblock column %d waited for %s (had %d of %d; %d of the launch's %d workgroups had started); the frames enqueued "
                  "since the last synchronisation are not valid",
                  (flag[1] >> 8) & 255, flag[1] >> 16, flag[2], what[flag[1] & 3], flag[3] & 0xffff, flag[3] >> 16, flag[4], flag[5]);
    }
  }
  return VP9HIP_OK;
}

extern "C" void *vp9hip_malloc(vp9hip_ctx *ctx, size_t bytes) {
  if (!ctx) return NULL;
  void *p = NULL;
  (void)hipSetDevice(ctx->device);
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    snprintf(ctx->err, sizeof(ctx->err), "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return NULL;
  }
  return p;
}

extern "C" void vp9hip_free(vp9hip_ctx *ctx, void *dptr) {
  if (!ctx || !dptr) return;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dptr);
}

extern "C" int vp9hip_memcpy_h2d(vp9hip_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return VP9HIP_OK;
}

extern "C" int vp9hip_memcpy_d2h(vp9hip_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return VP9HIP_OK;
}

extern "C" int vp9hip_memset(vp9hip_ctx *ctx, void *dst, int value, size_t bytes) {
  if (!ctx) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
  return VP9HIP_OK;
}

int vp9hip_ensure_scratch(vp9hip_ctx *ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return VP9HIP_OK;
  VP9HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  ctx->scratch = NULL;
  ctx->scratch_bytes = 0;
  size_t want = bytes + (bytes >> 2);
  VP9HIP_CHECK(ctx, hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return VP9HIP_OK;
}

static int timer_slot(vp9hip_ctx *ctx, int slot) {
  if (!ctx) return VP9HIP_EINVAL;
  if (slot < 0 || slot >= VP9HIP_TIMER_SLOTS) VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "timer slot %d out of range", slot);
  if (!ctx->ev_begin) {
    ctx->ev_begin = (hipEvent_t *)calloc(VP9HIP_TIMER_SLOTS, sizeof(hipEvent_t));
    ctx->ev_end = (hipEvent_t *)calloc(VP9HIP_TIMER_SLOTS, sizeof(hipEvent_t));
    if (!ctx->ev_begin || !ctx->ev_end) VP9HIP_FAIL(ctx, VP9HIP_ENOMEM, "out of host memory");
  }
  if (!ctx->ev_begin[slot]) {
    VP9HIP_CHECK(ctx, hipEventCreate(&ctx->ev_begin[slot]));
    VP9HIP_CHECK(ctx, hipEventCreate(&ctx->ev_end[slot]));
  }
  return VP9HIP_OK;
}

extern "C" int vp9hip_timer_begin(vp9hip_ctx *ctx, int slot) {
  int rc = timer_slot(ctx, slot);
  if (rc) return rc;
  VP9HIP_CHECK(ctx, hipEventRecord(ctx->ev_begin[slot], ctx->stream));
  return VP9HIP_OK;
}

extern "C" int vp9hip_timer_end(vp9hip_ctx *ctx, int slot) {
  int rc = timer_slot(ctx, slot);
  if (rc) return rc;
  VP9HIP_CHECK(ctx, hipEventRecord(ctx->ev_end[slot], ctx->stream));
  return VP9HIP_OK;
}

extern "C" int vp9hip_timer_read(vp9hip_ctx *ctx, int slot, float *ms) {
  int rc = timer_slot(ctx, slot);
  if (rc) return rc;
  if (!ms) return VP9HIP_EINVAL;
  VP9HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_end[slot]));
  VP9HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev_begin[slot], ctx->ev_end[slot]));
  return VP9HIP_OK;
}
