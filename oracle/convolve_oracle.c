/*
 * convolve_oracle.c — CPU restatement of the VP9 8-tap sub-pel convolve family.
 * TEST INFRASTRUCTURE ONLY (see vp9_oracle.h).
 *
 * Follows (relative to /root/reference/libvpx/):
 *   vpx_dsp/vpx_convolve.c:22-114   convolve_{horiz,avg_horiz,vert,avg_vert}
 *   vpx_dsp/vpx_convolve.c:116-240  vpx_convolve8*_c, vpx_convolve_{copy,avg}_c
 *   vpx_dsp/vpx_convolve.c:242-290  vpx_scaled_*_c (aliases of the above)
 *   vpx_dsp/vpx_convolve.c:292-535  highbd twins
 *   vp9/common/vp9_filter.c:14-82   kernel banks, order of vp9_filter_kernels[]
 *
 * One generic routine: a sample is src[q4>>4 + k - 3] weighted by kernel[q4&15][k],
 * sum rounded by 64 and shifted by 7, clipped to the pixel range.  The 2-D form
 * filters rows into a clipped intermediate (vpx_convolve.c:156-188) and then
 * filters that vertically.
 */
#include <string.h>

#include "vp9_oracle.h"

/* vp9/common/vp9_filter.c:14-82.  Index = INTERP_FILTER (vp9_filter.h:23-28):
 * 0 EIGHTTAP (Lagrangian "regular"), 1 EIGHTTAP_SMOOTH, 2 EIGHTTAP_SHARP,
 * 3 BILINEAR, 4 FOURTAP. */
static const int16_t kBank[5][16][8] = {
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

const int16_t (*vp9o_filter_kernels(int filter))[8] { return kBank[filter]; }

static inline int clipmax(int v, int mx) { return v < 0 ? 0 : v > mx ? mx : v; }

/* Generic 1-D pass over 16-bit samples (8-bit callers widen first).
 * dir_stride: element step between taps (1 = horizontal, stride = vertical). */
static void pass1d(const uint16_t *src, ptrdiff_t sstride, uint16_t *dst, ptrdiff_t dstride,
                   const int16_t (*kern)[8], int p0_q4, int step_q4, int w, int h, int vertical,
                   int avg, int mx) {
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const int pos = p0_q4 + (vertical ? y : x) * step_q4;
      const int16_t *f = kern[pos & 15];
      const uint16_t *s = vertical ? src + ((pos >> 4) - 3) * sstride + x
                                   : src + y * sstride + (pos >> 4) - 3;
      const ptrdiff_t ts = vertical ? sstride : 1;
      int sum = 0;
      for (int k = 0; k < 8; ++k) sum += s[k * ts] * f[k];
      int v = clipmax((sum + 64) >> 7, mx);
      uint16_t *d = dst + y * dstride + x;
      *d = avg ? (uint16_t)((*d + v + 1) >> 1) : (uint16_t)v;
    }
  }
}

/* Works on a widened copy so that one routine serves both depths. */
static void convolve_generic(int mode, const uint16_t *src, ptrdiff_t sstride, uint16_t *dst,
                             ptrdiff_t dstride, const int16_t (*kern)[8], int x0_q4, int xs,
                             int y0_q4, int ys, int w, int h, int mx) {
  const int do_h = mode & 1, do_v = (mode >> 1) & 1, avg = (mode >> 2) & 1;
  if (!do_h && !do_v) {
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        uint16_t *d = dst + y * dstride + x;
        uint16_t s = src[y * sstride + x];
        *d = avg ? (uint16_t)((*d + s + 1) >> 1) : s;
      }
    return;
  }
  if (do_h && !do_v) {
    pass1d(src, sstride, dst, dstride, kern, x0_q4, xs, w, h, 0, avg, mx);
    return;
  }
  if (!do_h && do_v) {
    pass1d(src, sstride, dst, dstride, kern, y0_q4, ys, w, h, 1, avg, mx);
    return;
  }
  /* 2-D: vpx_convolve8_c (vpx_convolve.c:156-188) */
  static const int TS = 64;
  uint16_t temp[64 * (64 * 4 + 8)];
  uint16_t out[64 * 64];
  const int ih = (((h - 1) * ys + y0_q4) >> 4) + 8;
  pass1d(src - 3 * sstride, sstride, temp, TS, kern, x0_q4, xs, w, ih, 0, 0, mx);
  pass1d(temp + 3 * TS, TS, out, TS, kern, y0_q4, ys, w, h, 1, 0, mx);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      uint16_t *d = dst + y * dstride + x;
      uint16_t s = out[y * TS + x];
      *d = avg ? (uint16_t)((*d + s + 1) >> 1) : s;
    }
}

/* widen a window of an 8-bit plane into 16 bit; returns pointer to (0,0) */
#define MAXWIN_W (64 * 4 + 16)
#define MAXWIN_H (64 * 4 + 16)

void vp9o_convolve(int mode, int scaled, const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst,
                   ptrdiff_t dst_stride, const int16_t (*kernel)[8], int x0_q4, int x_step_q4,
                   int y0_q4, int y_step_q4, int w, int h) {
  (void)scaled; /* vpx_scaled_* are aliases of the same arithmetic */
  static __thread uint16_t win[MAXWIN_W * MAXWIN_H];
  uint16_t d16[64 * 64];
  const int do_h = mode & 1, do_v = (mode >> 1) & 1;
  /* source extent actually touched */
  const int ww = do_h ? (((w - 1) * x_step_q4 + x0_q4) >> 4) + 8 : w;
  const int wh = do_v ? (((h - 1) * y_step_q4 + y0_q4) >> 4) + 8 : h;
  const int ox = do_h ? 3 : 0, oy = do_v ? 3 : 0;
  for (int y = 0; y < wh; ++y)
    for (int x = 0; x < ww; ++x) win[y * MAXWIN_W + x] = src[(y - oy) * src_stride + (x - ox)];
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) d16[y * 64 + x] = dst[y * dst_stride + x];
  convolve_generic(mode, win + oy * MAXWIN_W + ox, MAXWIN_W, d16, 64, kernel, x0_q4, x_step_q4,
                   y0_q4, y_step_q4, w, h, 255);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) dst[y * dst_stride + x] = (uint8_t)d16[y * 64 + x];
}

void vp9o_highbd_convolve(int mode, int scaled, const uint16_t *src, ptrdiff_t src_stride,
                          uint16_t *dst, ptrdiff_t dst_stride, const int16_t (*kernel)[8],
                          int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h,
                          int bd) {
  (void)scaled;
  convolve_generic(mode, src, src_stride, dst, dst_stride, kernel, x0_q4, x_step_q4, y0_q4,
                   y_step_q4, w, h, (1 << bd) - 1);
}

/* ---- block-level inter predictor with decoder border emulation --------------------------
 * Restates dec_build_inter_predictors + build_mc_border / high_build_mc_border +
 * extend_and_predict (vp9/decoder/vp9_decodeframe.c:432-690) for one block and one
 * reference: the (w+7)x(h+7) window is fetched with coordinates clamped to the reference
 * plane's crop size [0,fw-1]x[0,fh-1] (that is what the emulated border holds; blocks fully
 * inside read the same pixels directly), then handed to the predictor the scale-factor table
 * selects (vp9/common/vp9_scale.c:79-170): copy / horiz / vert / 2-D by which phase is
 * non-zero, the vpx_scaled_* forms when a step differs from 16.
 * pos_q4 = 16*integer position + phase of the block's first sample. */
static void inter_block(const uint8_t *ref8, const uint16_t *ref16, int rstride, int fw, int fh,
                        int px_q4, int py_q4, int xs, int ys, int filter, int w, int h,
                        uint8_t *dst8, uint16_t *dst16, int dstride, int avg, int bd) {
  static __thread uint16_t win[(64 * 2 + 16) * (64 * 2 + 16)];
  const int WS = 64 * 2 + 16;
  const int x0 = px_q4 >> 4, y0 = py_q4 >> 4, subx = px_q4 & 15, suby = py_q4 & 15;
  const int ww = (((w - 1) * xs + subx) >> 4) + 8, wh = (((h - 1) * ys + suby) >> 4) + 8;
  for (int r = 0; r < wh; ++r)
    for (int c = 0; c < ww; ++c) {
      int sx = x0 - 3 + c, sy = y0 - 3 + r;
      sx = sx < 0 ? 0 : sx > fw - 1 ? fw - 1 : sx;
      sy = sy < 0 ? 0 : sy > fh - 1 ? fh - 1 : sy;
      win[r * WS + c] = ref16 ? ref16[sy * rstride + sx] : ref8[sy * rstride + sx];
    }
  /* sf->predict[subpel_x != 0][subpel_y != 0][avg], vp9_scale.c:79-130 */
  int do_h, do_v;
  if (xs == 16 && ys == 16) {
    do_h = subx != 0;
    do_v = suby != 0;
  } else if (xs == 16) { /* y scaled: always vertical; 2-D when x has a phase */
    do_h = subx != 0;
    do_v = 1;
  } else if (ys == 16) {
    do_h = 1;
    do_v = suby != 0;
  } else {
    do_h = do_v = 1;
  }
  const int mode = do_h | (do_v << 1) | (avg ? 4 : 0);
  uint16_t d16[64 * 64];
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) d16[r * 64 + c] = dst16 ? dst16[r * dstride + c] : dst8[r * dstride + c];
  convolve_generic(mode, win + 3 * WS + 3, WS, d16, 64, kBank[filter], subx, xs, suby, ys, w, h,
                   (1 << bd) - 1);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      if (dst16)
        dst16[r * dstride + c] = d16[r * 64 + c];
      else
        dst8[r * dstride + c] = (uint8_t)d16[r * 64 + c];
    }
}

void vp9o_inter_predict_block(const uint8_t *ref, int ref_stride, int fw, int fh, int px_q4,
                              int py_q4, int xs, int ys, int filter, int w, int h, uint8_t *dst,
                              int dst_stride, int avg) {
  inter_block(ref, NULL, ref_stride, fw, fh, px_q4, py_q4, xs, ys, filter, w, h, dst, NULL,
              dst_stride, avg, 8);
}

void vp9o_highbd_inter_predict_block(const uint16_t *ref, int ref_stride, int fw, int fh,
                                     int px_q4, int py_q4, int xs, int ys, int filter, int w,
                                     int h, uint16_t *dst, int dst_stride, int avg, int bd) {
  inter_block(NULL, ref, ref_stride, fw, fh, px_q4, py_q4, xs, ys, filter, w, h, NULL, dst,
              dst_stride, avg, bd);
}
