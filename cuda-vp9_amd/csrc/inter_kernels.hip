// inter_kernels.hip — batched inter prediction: 8-tap sub-pel convolve (SURVEY §8 a5–a7).
//
// One wavefront (= one 64-thread workgroup) per prediction task.  The wave stages the
// clamped (w+7)x(h+7) reference window in LDS (coordinate clamping == libvpx's decoder border
// emulation, vp9_decodeframe.c:432-505 build_mc_border), filters rows into a clipped LDS
// intermediate (vpx_convolve.c:156-188: the clip BETWEEN the passes is normative) and then
// filters columns.  Phase-0 kernels are the identity ({0,0,0,128,0,0,0,0}: (128*p+64)>>7 = p),
// so always running both passes equals libvpx's copy / horiz-only / vert-only dispatch
// (vp9_scale.c:79-130).  Compound prediction = second reference averaged into the first
// ((a+b+1)>>1, vpx_convolve_avg_c :226-240).
//
// Algorithmic bytes per task: w*h*bps read (reference) + w*h*bps written (+ w*h*bps more
// reference for compound) + 32 (descriptor).
#include "vp9hip_internal.h"

namespace {

// vp9/common/vp9_filter.c:14-82, in INTERP_FILTER order (vp9_filter.h:23-28):
// EIGHTTAP, EIGHTTAP_SMOOTH, EIGHTTAP_SHARP, BILINEAR, FOURTAP.
__device__ const int16_t kFilters[5][16][8] = {
  { { 0, 0, 0, 128, 0, 0, 0, 0 },        { 0, 1, -5, 126, 8, -3, 1, 0 },
    { -1, 3, -10, 122, 18, -6, 2, 0 },   { -1, 4, -13, 118, 27, -9, 3, -1 },
    { -1, 4, -16, 112, 37, -11, 4, -1 }, { -1, 5, -18, 105, 48, -14, 4, -1 },
    { -1, 5, -19, 97, 58, -16, 5, -1 },  { -1, 6, -19, 88, 68, -18, 5, -1 },
    { -1, 6, -19, 78, 78, -19, 6, -1 },  { -1, 5, -18, 68, 88, -19, 6, -1 },
    { -1, 5, -16, 58, 97, -19, 5, -1 },  { -1, 4, -14, 48, 105, -18, 5, -1 },
    { -1, 4, -11, 37, 112, -16, 4, -1 }, { -1, 3, -9, 27, 118, -13, 4, -1 },
    { 0, 2, -6, 18, 122, -10, 3, -1 },   { 0, 1, -3, 8, 126, -5, 1, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 },       { -3, -1, 32, 64, 38, 1, -3, 0 },
    { -2, -2, 29, 63, 41, 2, -3, 0 },   { -2, -2, 26, 63, 43, 4, -4, 0 },
    { -2, -3, 24, 62, 46, 5, -4, 0 },   { -2, -3, 21, 60, 49, 7, -4, 0 },
    { -1, -4, 18, 59, 51, 9, -4, 0 },   { -1, -4, 16, 57, 53, 12, -4, -1 },
    { -1, -4, 14, 55, 55, 14, -4, -1 }, { -1, -4, 12, 53, 57, 16, -4, -1 },
    { 0, -4, 9, 51, 59, 18, -4, -1 },   { 0, -4, 7, 49, 60, 21, -3, -2 },
    { 0, -4, 5, 46, 62, 24, -3, -2 },   { 0, -4, 4, 43, 63, 26, -2, -2 },
    { 0, -3, 2, 41, 63, 29, -2, -2 },   { 0, -3, 1, 38, 64, 32, -1, -3 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 },         { -1, 3, -7, 127, 8, -3, 1, 0 },
    { -2, 5, -13, 125, 17, -6, 3, -1 },   { -3, 7, -17, 121, 27, -10, 5, -2 },
    { -4, 9, -20, 115, 37, -13, 6, -2 },  { -4, 10, -23, 108, 48, -16, 8, -3 },
    { -4, 10, -24, 100, 59, -19, 9, -3 }, { -4, 11, -24, 90, 70, -21, 10, -4 },
    { -4, 11, -23, 80, 80, -23, 11, -4 }, { -4, 10, -21, 70, 90, -24, 11, -4 },
    { -3, 9, -19, 59, 100, -24, 10, -4 }, { -3, 8, -16, 48, 108, -23, 10, -4 },
    { -2, 6, -13, 37, 115, -20, 9, -4 },  { -2, 5, -10, 27, 121, -17, 7, -3 },
    { -1, 3, -6, 17, 125, -13, 5, -2 },   { 0, 1, -3, 8, 127, -7, 3, -1 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 },  { 0, 0, 0, 120, 8, 0, 0, 0 },
    { 0, 0, 0, 112, 16, 0, 0, 0 }, { 0, 0, 0, 104, 24, 0, 0, 0 },
    { 0, 0, 0, 96, 32, 0, 0, 0 },  { 0, 0, 0, 88, 40, 0, 0, 0 },
    { 0, 0, 0, 80, 48, 0, 0, 0 },  { 0, 0, 0, 72, 56, 0, 0, 0 },
    { 0, 0, 0, 64, 64, 0, 0, 0 },  { 0, 0, 0, 56, 72, 0, 0, 0 },
    { 0, 0, 0, 48, 80, 0, 0, 0 },  { 0, 0, 0, 40, 88, 0, 0, 0 },
    { 0, 0, 0, 32, 96, 0, 0, 0 },  { 0, 0, 0, 24, 104, 0, 0, 0 },
    { 0, 0, 0, 16, 112, 0, 0, 0 }, { 0, 0, 0, 8, 120, 0, 0, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 },     { 0, 0, -4, 126, 8, -2, 0, 0 },
    { 0, 0, -6, 120, 18, -4, 0, 0 },  { 0, 0, -8, 114, 28, -6, 0, 0 },
    { 0, 0, -10, 108, 36, -6, 0, 0 }, { 0, 0, -12, 102, 46, -8, 0, 0 },
    { 0, 0, -12, 94, 56, -10, 0, 0 }, { 0, 0, -12, 84, 66, -10, 0, 0 },
    { 0, 0, -12, 76, 76, -12, 0, 0 }, { 0, 0, -10, 66, 84, -12, 0, 0 },
    { 0, 0, -10, 56, 94, -12, 0, 0 }, { 0, 0, -8, 46, 102, -12, 0, 0 },
    { 0, 0, -6, 36, 108, -10, 0, 0 }, { 0, 0, -6, 28, 114, -8, 0, 0 },
    { 0, 0, -4, 18, 120, -6, 0, 0 },  { 0, 0, -2, 8, 126, -4, 0, 0 } }
};

struct RefSet {
  FrameDev f[VP9HIP_MAX_REFS];
};

constexpr int WIN = 72;  // max window extent (64 + 7, or 2*31 + 1 + 8 for 2:1 scaled 32x32 tiles)

template <typename Pix>
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// Predict one tile (tw x th at tile offset tx,ty inside the task) from one reference.
// Writes the result into dst (or averages into it when avg).
template <typename Pix>
__device__ void predict_tile(Pix *win, Pix *tmp, const FrameDev &rf, int plane, int px_q4, int py_q4,
                             int xs, int ys, int filt, int tw, int th, Pix *dst, int dstride, int vis_w,
                             int vis_h, bool avg, int maxv) {
  const int lane = threadIdx.x;
  const Pix *src = (const Pix *)rf.plane[plane];
  const int sstride = rf.stride[plane];
  const int fw = rf.width[plane], fh = rf.height[plane];
  const int x0 = px_q4 >> 4, y0 = py_q4 >> 4;
  const int subx = px_q4 & 15, suby = py_q4 & 15;
  const int ww = (((tw - 1) * xs + subx) >> 4) + 8;  // window columns  (x0-3 .. )
  const int wh = (((th - 1) * ys + suby) >> 4) + 8;  // window rows     (y0-3 .. ) == intermediate height
  // 1. stage the clamped window
  for (int i = lane; i < ww * wh; i += 64) {
    const int r = i / ww, c = i - r * ww;
    const int sx = clampi<Pix>(x0 - 3 + c, 0, fw - 1);
    const int sy = clampi<Pix>(y0 - 3 + r, 0, fh - 1);
    win[r * WIN + c] = src[(size_t)sy * sstride + sx];
  }
  __syncthreads();
  // 2. rows -> clipped intermediate tmp[wh][tw]
  for (int i = lane; i < tw * wh; i += 64) {
    const int r = i / tw, c = i - r * tw;
    const int pos = subx + c * xs;
    const int16_t *f = kFilters[filt][pos & 15];
    const Pix *s = win + r * WIN + (pos >> 4);
    int sum = 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)s[k] * f[k];
    tmp[r * 64 + c] = (Pix)clampi<Pix>(sum >> 7, 0, maxv);
  }
  __syncthreads();
  // 3. columns -> destination
  for (int i = lane; i < tw * th; i += 64) {
    const int r = i / tw, c = i - r * tw;
    const int pos = suby + r * ys;
    const int16_t *f = kFilters[filt][pos & 15];
    const Pix *s = tmp + (pos >> 4) * 64 + c;
    int sum = 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)s[k * 64] * f[k];
    int v = clampi<Pix>(sum >> 7, 0, maxv);
    if (r < vis_h && c < vis_w) {
      Pix *d = dst + (size_t)r * dstride + c;
      if (avg) v = ((int)*d + v + 1) >> 1;
      *d = (Pix)v;
    }
  }
  __syncthreads();
}

template <typename Pix>
__global__ __launch_bounds__(64) void inter_pred_kernel(const vp9hip_inter_task *__restrict__ tasks, int n_tasks,
                                                        RefSet refs, FrameDev dstf) {
  __shared__ Pix win[WIN * WIN];
  __shared__ Pix tmp[WIN * 64];
  const int ti = blockIdx.x;
  if (ti >= n_tasks) return;
  const vp9hip_inter_task t = tasks[ti];
  const int plane = t.plane;
  const int filt = (t.flags >> 1) & 7;
  const int nref = (t.flags & 1) ? 2 : 1;
  const int maxv = (1 << dstf.bit_depth) - 1;
  Pix *dplane = (Pix *)dstf.plane[plane];
  const int dstride = dstf.stride[plane];
  for (int r = 0; r < nref; ++r) {
    const FrameDev &rf = refs.f[t.ref[r]];
    const int xs = t.step_x[r], ys = t.step_y[r];
    // unscaled: one 64x64 tile; scaled: 32x32 tiles keep the window inside WIN
    const int tile = (xs == 16 && ys == 16) ? 64 : 32;
    for (int ty = 0; ty < t.h; ty += tile) {
      for (int tx = 0; tx < t.w; tx += tile) {
        const int tw = min(tile, t.w - tx), th = min(tile, t.h - ty);
        const int dx = t.dst_x + tx, dy = t.dst_y + ty;
        // visible part of the tile (blocks may overhang the aligned frame)
        const int vis_w = min(tw, dstf.awidth[plane] - dx), vis_h = min(th, dstf.aheight[plane] - dy);
        if (vis_w <= 0 || vis_h <= 0) continue;
        predict_tile<Pix>(win, tmp, rf, plane, t.pos_x[r] + tx * xs, t.pos_y[r] + ty * ys, xs, ys, filt, tw, th,
                          dplane + (size_t)dy * dstride + dx, dstride, vis_w, vis_h, r == 1, maxv);
      }
    }
  }
}

}  // namespace

extern "C" int vp9hip_inter_pred_batch(vp9hip_ctx *ctx, const vp9hip_inter_task *d_tasks, int n_tasks,
                                       const vp9hip_frame *refs, int n_refs, const vp9hip_frame *dst) {
  if (!ctx) return VP9HIP_EINVAL;
  if (!d_tasks || n_tasks < 0 || !refs || n_refs <= 0 || n_refs > VP9HIP_MAX_REFS || !frame_ok(dst))
    VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_inter_pred_batch: bad argument");
  RefSet rs;
  memset(&rs, 0, sizeof(rs));
  for (int i = 0; i < n_refs; ++i) {
    if (!frame_ok(&refs[i]) || refs[i].hbd != dst->hbd || refs[i].bit_depth != dst->bit_depth)
      VP9HIP_FAIL(ctx, VP9HIP_EINVAL, "vp9hip_inter_pred_batch: reference %d does not match the destination format", i);
    rs.f[i] = to_dev(&refs[i]);
  }
  if (n_tasks == 0) return VP9HIP_OK;
  const FrameDev d = to_dev(dst);
  if (dst->hbd)
    hipLaunchKernelGGL(inter_pred_kernel<uint16_t>, dim3(n_tasks), dim3(64), 0, ctx->stream, d_tasks, n_tasks, rs, d);
  else
    hipLaunchKernelGGL(inter_pred_kernel<uint8_t>, dim3(n_tasks), dim3(64), 0, ctx->stream, d_tasks, n_tasks, rs, d);
  VP9HIP_CHECK(ctx, hipGetLastError());
  return VP9HIP_OK;
}
