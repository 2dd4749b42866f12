// rtcd_twins.hip — block-level `_hip` twins of the vpx_dsp_rtcd / vp9_rtcd prototypes
// (include/vp9hip_rtcd.h).  Host pointers in, host pointers out: each call moves one block to
// the GPU, runs the SAME kernels as the batched entry points and moves the result back.
#include <stdlib.h>

#include "../../include/vp9hip_rtcd.h"
#include "vp9hip_internal.h"

namespace {

vp9hip_ctx *g_ctx = NULL;
char g_rtcd_err[512] = "";

// vp9/common/vp9_filter.c:14-82 in vp9_filter_kernels[] order; used to recognise which bank a
// caller-supplied InterpKernel table is (the kernels index their own copy by INTERP_FILTER).
#include "filter_table.inc"
const int16_t kBanks[5][16][8] = VP9HIP_FILTER_TABLE;

vp9hip_ctx *ctx() {
  if (g_ctx) return g_ctx;
  const char *e = getenv("VP9HIP_DEVICE");
  if (vp9hip_create(e ? atoi(e) : 0, &g_ctx) != 0) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "default context: %s", vp9hip_last_error(NULL));
    g_ctx = NULL;
  }
  return g_ctx;
}

#define TW_CHECK(call)                                                         \
  do {                                                                         \
    if ((call) != 0) {                                                         \
      snprintf(g_rtcd_err, sizeof(g_rtcd_err), "%s", vp9hip_last_error(c));    \
      goto done;                                                               \
    }                                                                          \
  } while (0)

// one-plane frame descriptor over a device buffer
vp9hip_frame one_plane(void *d, int w, int h, int stride, int bd, int hbd) {
  vp9hip_frame f;
  memset(&f, 0, sizeof(f));
  f.plane[0] = d;
  f.stride[0] = stride;
  f.width[0] = f.awidth[0] = w;
  f.height[0] = f.aheight[0] = h;
  f.bit_depth = bd;
  f.hbd = hbd;
  return f;
}

template <typename Pix>
void gather(Pix *dst, int dstride, const Pix *src, ptrdiff_t sstride, int w, int h) {
  for (int y = 0; y < h; ++y) memcpy(dst + (size_t)y * dstride, src + (ptrdiff_t)y * sstride, (size_t)w * sizeof(Pix));
}

// ---- transforms -----------------------------------------------------------------------------
void twin_txfm(const int32_t *in, void *dest, int stride, int n, int tx_type, int lossless, int variant, int bd,
               int hbd) {
  vp9hip_ctx *c = ctx();
  if (!c) return;
  g_rtcd_err[0] = 0;
  const int bps = hbd ? 2 : 1;
  int32_t coeffs[32 * 32];
  // the eob-limited variants transform only their first rows (inv_txfm.c:345-352, 744-752,
  // 773-781, 1214-1222, 1241-1249): rows they ignore are zeroed here
  int rows = n;
  if (!lossless && tx_type == 0) {
    if (n == 8 && variant == 12) rows = 4;
    if (n == 16 && variant == 10) rows = 4;
    if (n == 16 && variant == 38) rows = 8;
    if (n == 32 && variant == 34) rows = 8;
    if (n == 32 && variant == 135) rows = 16;
  }
  memset(coeffs, 0, sizeof(coeffs));
  if (variant == 1)
    coeffs[0] = in[0];
  else
    memcpy(coeffs, in, (size_t)rows * n * sizeof(int32_t));
  const int dstride = 64;
  unsigned char host[32 * 64 * 2];
  memset(host, 0, sizeof(host));
  if (hbd)
    gather((uint16_t *)host, dstride, (const uint16_t *)dest, stride, n, n);
  else
    gather((uint8_t *)host, dstride, (const uint8_t *)dest, stride, n, n);
  void *d_c = vp9hip_malloc(c, sizeof(coeffs)), *d_p = vp9hip_malloc(c, sizeof(host)), *d_b = vp9hip_malloc(c, 64);
  vp9hip_txb b;
  memset(&b, 0, sizeof(b));
  b.tx_size = n == 4 ? 0 : n == 8 ? 1 : n == 16 ? 2 : 3;
  b.tx_type = (uint8_t)(tx_type | (lossless ? 0x80 : 0));
  b.eob = (uint16_t)variant;
  int32_t counts[4] = { 0, 0, 0, 0 };
  counts[b.tx_size] = 1;
  vp9hip_frame f = one_plane(d_p, n, n, dstride, bd, hbd);
  if (!d_c || !d_p || !d_b) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "%s", vp9hip_last_error(c));
    goto done;
  }
  TW_CHECK(vp9hip_memcpy_h2d(c, d_c, coeffs, sizeof(coeffs)));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_p, host, (size_t)n * dstride * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_b, &b, sizeof(b)));
  TW_CHECK(vp9hip_idct_add_batch(c, (const vp9hip_txb *)d_b, counts, (const int32_t *)d_c, &f));
  TW_CHECK(vp9hip_memcpy_d2h(c, host, d_p, (size_t)n * dstride * bps));
  if (hbd)
    gather((uint16_t *)dest, stride, (const uint16_t *)host, dstride, n, n);
  else
    gather((uint8_t *)dest, stride, (const uint8_t *)host, dstride, n, n);
done:
  vp9hip_free(c, d_c);
  vp9hip_free(c, d_p);
  vp9hip_free(c, d_b);
}

// ---- convolve -------------------------------------------------------------------------------
int bank_of(const vp9hip_interp_kernel *k) {
  if (!k) return 0;  // copy / avg are called with a NULL kernel (vpx_convolve.c:203)
  for (int f = 0; f < 5; ++f)
    if (memcmp(k, kBanks[f], sizeof(kBanks[f])) == 0) return f;
  return -1;
}

void twin_convolve(int mode, const void *src, ptrdiff_t sstride, void *dst, ptrdiff_t dstride,
                   const vp9hip_interp_kernel *kernel, int x0_q4, int xs, int y0_q4, int ys, int w, int h, int bd,
                   int hbd) {
  vp9hip_ctx *c = ctx();
  if (!c) return;
  g_rtcd_err[0] = 0;
  const int do_h = mode & 1, do_v = (mode >> 1) & 1, avg = (mode >> 2) & 1;
  const int bank = (do_h || do_v) ? bank_of(kernel) : 0;
  if (bank < 0) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "convolve twin: kernel table is not one of vp9_filter_kernels[]");
    return;
  }
  if (w < 1 || h < 1 || w > 64 || h > 64 || xs < 1 || xs > 64 || ys < 1 || ys > 64) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "convolve twin: unsupported size/step");
    return;
  }
  const int bps = hbd ? 2 : 1;
  // source window actually read by the libvpx function for this mode
  if (!do_h) { x0_q4 = 0; xs = 16; }
  if (!do_v) { y0_q4 = 0; ys = 16; }
  const int ox = do_h ? 3 : 0, oy = do_v ? 3 : 0;
  const int ww = do_h ? (((w - 1) * xs + x0_q4) >> 4) + 8 : w;
  const int wh = do_v ? (((h - 1) * ys + y0_q4) >> 4) + 8 : h;
  const int wstride = (ww + 63) & ~63, bstride = 64;
  unsigned char *hw = (unsigned char *)calloc((size_t)wstride * wh, bps), *hb = (unsigned char *)calloc(bstride * 64, bps);
  void *d_w = vp9hip_malloc(c, (size_t)wstride * wh * bps), *d_d = vp9hip_malloc(c, bstride * 64 * bps),
       *d_o = vp9hip_malloc(c, bstride * 64 * bps), *d_t = vp9hip_malloc(c, sizeof(vp9hip_inter_task));
  vp9hip_inter_task t;
  vp9hip_frame refs[2], df;
  int32_t counts[VP9HIP_INTER_CLASSES] = { 0 };
  if (!hw || !hb || !d_w || !d_d || !d_o || !d_t) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "convolve twin: out of memory");
    goto done;
  }
  if (hbd) {
    gather((uint16_t *)hw, wstride, (const uint16_t *)src - oy * sstride - ox, sstride, ww, wh);
    gather((uint16_t *)hb, bstride, (const uint16_t *)dst, dstride, w, h);
  } else {
    gather((uint8_t *)hw, wstride, (const uint8_t *)src - oy * sstride - ox, sstride, ww, wh);
    gather((uint8_t *)hb, bstride, (const uint8_t *)dst, dstride, w, h);
  }
  memset(&t, 0, sizeof(t));
  t.w = (uint8_t)w;
  t.h = (uint8_t)h;
  t.flags = (uint8_t)((bank << 1) | (avg ? 1 : 0));
  {
    // avg: "reference 0" is the old destination copied at phase 0 (identity), reference 1 the
    // filtered source averaged in — the (dst + pred + 1) >> 1 of the *_avg_* functions
    const int r = avg ? 1 : 0;
    t.pos_x[r] = 16 * ox + x0_q4;
    t.pos_y[r] = 16 * oy + y0_q4;
    t.step_x[r] = (uint8_t)xs;
    t.step_y[r] = (uint8_t)ys;
    t.ref[r] = 0;
    if (avg) {
      t.pos_x[0] = t.pos_y[0] = 0;
      t.step_x[0] = t.step_y[0] = 16;
      t.ref[0] = 1;
    }
  }
  refs[0] = one_plane(d_w, ww, wh, wstride, bd, hbd);
  refs[1] = one_plane(d_o, w, h, bstride, bd, hbd);
  df = one_plane(d_d, w, h, bstride, bd, hbd);
  {
    // (16-bit samples and scaled steps take the generic kernel, like the reference's dispatch table does not
    // distinguish them either: vp9_scale.c:79-170)
    const int cls = vp9hip_inter_class(w, h, !hbd && xs == 16 && ys == 16);
    counts[cls] = 1;
  }
  TW_CHECK(vp9hip_memcpy_h2d(c, d_w, hw, (size_t)wstride * wh * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_d, hb, (size_t)bstride * 64 * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_o, hb, (size_t)bstride * 64 * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_t, &t, sizeof(t)));
  TW_CHECK(vp9hip_inter_pred_batch(c, (const vp9hip_inter_task *)d_t, counts, refs, 2, &df));
  TW_CHECK(vp9hip_memcpy_d2h(c, hb, d_d, (size_t)bstride * 64 * bps));
  if (hbd)
    gather((uint16_t *)dst, dstride, (const uint16_t *)hb, bstride, w, h);
  else
    gather((uint8_t *)dst, dstride, (const uint8_t *)hb, bstride, w, h);
done:
  free(hw);
  free(hb);
  vp9hip_free(c, d_w);
  vp9hip_free(c, d_d);
  vp9hip_free(c, d_o);
  vp9hip_free(c, d_t);
}

// ---- intra predictors ------------------------------------------------------------------------
// mode ids: 0..9 PREDICTION_MODE, 10 DC_128, 11 DC_LEFT, 12 DC_TOP, 13 D45E, 14 D63E, 15 HE, 16 VE (4x4 only)
void twin_intra(int mode, int bs, void *dst, ptrdiff_t stride, const void *above, const void *left, int bd, int hbd) {
  vp9hip_ctx *c = ctx();
  if (!c) return;
  g_rtcd_err[0] = 0;
  const int bps = hbd ? 2 : 1;
  const int fw = 1 + 2 * bs, fh = 1 + bs, fs = 128;
  unsigned char host[33 * 128 * 2];
  memset(host, 0, sizeof(host));
  // row 0 = above[-1 .. 2bs-1], column 0 (rows 1..bs) = left; the block sits at (1,1)
  for (int i = -1; i < 2 * bs; ++i) {
    if (hbd)
      ((uint16_t *)host)[1 + i] = ((const uint16_t *)above)[i];
    else
      host[1 + i] = ((const uint8_t *)above)[i];
  }
  for (int r = 0; r < bs; ++r) {
    if (hbd)
      ((uint16_t *)host)[(1 + r) * fs] = ((const uint16_t *)left)[r];
    else
      host[(1 + r) * fs] = ((const uint8_t *)left)[r];
  }
  void *d_p = vp9hip_malloc(c, sizeof(host)), *d_t = vp9hip_malloc(c, 64);
  vp9hip_intra_task t;
  memset(&t, 0, sizeof(t));
  t.x = t.y = 1;
  t.tx_size = bs == 4 ? 0 : bs == 8 ? 1 : bs == 16 ? 2 : 3;
  int have_top = 1, have_left = 1;
  if (mode == 10) have_top = have_left = 0;
  if (mode == 11) have_top = 0;
  if (mode == 12) have_left = 0;
  t.mode = (uint8_t)(mode >= 10 && mode <= 12 ? 0 : mode);
  t.flags = (uint8_t)(have_top | (have_left << 1) | 4 | 8);
  int32_t wave_start[2] = { 0, 1 };
  vp9hip_frame f = one_plane(d_p, fw, fh, fs, bd, hbd);
  if (!d_p || !d_t) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "%s", vp9hip_last_error(c));
    goto done;
  }
  TW_CHECK(vp9hip_memcpy_h2d(c, d_p, host, (size_t)fh * fs * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_t, &t, sizeof(t)));
  TW_CHECK(vp9hip_intra_pred_waves(c, (const vp9hip_intra_task *)d_t, wave_start, 1, NULL, &f));
  TW_CHECK(vp9hip_memcpy_d2h(c, host, d_p, (size_t)fh * fs * bps));
  for (int r = 0; r < bs; ++r) {
    if (hbd)
      memcpy((uint16_t *)dst + r * stride, (uint16_t *)host + (1 + r) * fs + 1, (size_t)bs * 2);
    else
      memcpy((uint8_t *)dst + r * stride, host + (1 + r) * fs + 1, (size_t)bs);
  }
done:
  vp9hip_free(c, d_p);
  vp9hip_free(c, d_t);
}

// ---- loop filter -------------------------------------------------------------------------------
// The edge is placed inside a one-superblock luma frame at mask column 1: horizontal edges on
// mask row 1 (y = 8), vertical edges on mask rows 0(,1) (x = 8); the thresholds become levels
// 1 and 2 of a private table.
void twin_lpf(int vertical, int kind, int dual, void *s, int pitch, const uint8_t *b0, const uint8_t *l0,
              const uint8_t *t0, const uint8_t *b1, const uint8_t *l1, const uint8_t *t1, int bd, int hbd) {
  vp9hip_ctx *c = ctx();
  if (!c) return;
  g_rtcd_err[0] = 0;
  const int bps = hbd ? 2 : 1;
  const int fs = 64, fdim = 32;
  unsigned char host[32 * 64 * 2];
  memset(host, 0, sizeof(host));
  const int lines = (dual ? 16 : 8);
  // copy the 16 samples across the edge for every line
  const int cw = vertical ? 16 : lines, chh = vertical ? lines : 16;
  const int fx = vertical ? 0 : 8, fy = 0;  // frame position of the copied rectangle
  const ptrdiff_t so = vertical ? -8 : -8 * (ptrdiff_t)pitch;
  if (hbd)
    gather((uint16_t *)host + fy * fs + fx, fs, (const uint16_t *)s + so, pitch, cw, chh);
  else
    gather((uint8_t *)host + fy * fs + fx, fs, (const uint8_t *)s + so, pitch, cw, chh);
  vp9hip_lfm m;
  memset(&m, 0, sizeof(m));
  const int tx = kind == 4 ? 0 : (kind == 8 ? 1 : 2);
  if (vertical) {
    m.left_y[tx] = 1ull << 1;
    m.lfl_y[1] = 1;
    if (dual) {
      m.left_y[tx] |= 1ull << 9;
      m.lfl_y[9] = (kind == 16) ? 1 : 2;
    }
  } else {
    m.above_y[tx] = 1ull << 9;
    m.lfl_y[9] = 1;
    if (dual) {
      m.above_y[tx] |= 1ull << 10;
      m.lfl_y[10] = (kind == 16) ? 1 : 2;
    }
  }
  vp9hip_lf_thresh th;
  memset(&th, 0, sizeof(th));
  th.mblim[1] = *b0; th.lim[1] = *l0; th.hev_thr[1] = *t0;
  if (dual && kind != 16) { th.mblim[2] = *b1; th.lim[2] = *l1; th.hev_thr[2] = *t1; }
  void *d_p = vp9hip_malloc(c, sizeof(host)), *d_m = vp9hip_malloc(c, sizeof(m));
  vp9hip_frame f = one_plane(d_p, fdim, fdim, fs, bd, hbd);
  if (!d_p || !d_m) {
    snprintf(g_rtcd_err, sizeof(g_rtcd_err), "%s", vp9hip_last_error(c));
    goto done;
  }
  TW_CHECK(vp9hip_memcpy_h2d(c, d_p, host, (size_t)fdim * fs * bps));
  TW_CHECK(vp9hip_memcpy_h2d(c, d_m, &m, sizeof(m)));
  TW_CHECK(vp9hip_loop_filter_frame(c, (const vp9hip_lfm *)d_m, 1, 1, &th, &f, 1));
  TW_CHECK(vp9hip_memcpy_d2h(c, host, d_p, (size_t)fdim * fs * bps));
  if (hbd)
    gather((uint16_t *)s + so, pitch, (const uint16_t *)host + fy * fs + fx, fs, cw, chh);
  else
    gather((uint8_t *)s + so, pitch, (const uint8_t *)host + fy * fs + fx, fs, cw, chh);
done:
  vp9hip_free(c, d_p);
  vp9hip_free(c, d_m);
}

}  // namespace

extern "C" const char *vp9hip_rtcd_last_error(void) { return g_rtcd_err; }

#define TX_TWIN(lo, hi, n, variant, lossless)                                                          \
  extern "C" void lo(const vp9hip_tran_low_t *in, uint8_t *d, int s) { twin_txfm(in, d, s, n, 0, lossless, variant, 8, 0); } \
  extern "C" void hi(const vp9hip_tran_low_t *in, uint16_t *d, int s, int bd) { twin_txfm(in, d, s, n, 0, lossless, variant, bd, 1); }
#define IHT_TWIN(lo, hi, n)                                                                            \
  extern "C" void lo(const vp9hip_tran_low_t *in, uint8_t *d, int s, int tt) { twin_txfm(in, d, s, n, tt, 0, n * n, 8, 0); } \
  extern "C" void hi(const vp9hip_tran_low_t *in, uint16_t *d, int s, int tt, int bd) { twin_txfm(in, d, s, n, tt, 0, n * n, bd, 1); }
#define CONV_TWIN(name, mode)                                                                          \
  extern "C" void name(const uint8_t *src, ptrdiff_t ss, uint8_t *dst, ptrdiff_t ds, const vp9hip_interp_kernel *k, \
                       int x0, int xs, int y0, int ys, int w, int h) {                                 \
    twin_convolve(mode, src, ss, dst, ds, k, x0, xs, y0, ys, w, h, 8, 0);                              \
  }
#define HCONV_TWIN(name, mode)                                                                         \
  extern "C" void name(const uint16_t *src, ptrdiff_t ss, uint16_t *dst, ptrdiff_t ds, const vp9hip_interp_kernel *k, \
                       int x0, int xs, int y0, int ys, int w, int h, int bd) {                         \
    twin_convolve(mode, src, ss, dst, ds, k, x0, xs, y0, ys, w, h, bd, 1);                             \
  }
#define INTRA_TWIN(lo, hi, mode, n)                                                                    \
  extern "C" void lo(uint8_t *d, ptrdiff_t s, const uint8_t *a, const uint8_t *l) { twin_intra(mode, n, d, s, a, l, 8, 0); } \
  extern "C" void hi(uint16_t *d, ptrdiff_t s, const uint16_t *a, const uint16_t *l, int bd) { twin_intra(mode, n, d, s, a, l, bd, 1); }
#define LPF_TWIN3(lo, hi, vert, kind, dual)                                                            \
  extern "C" void lo(uint8_t *s, int p, const uint8_t *b, const uint8_t *l, const uint8_t *t) {        \
    twin_lpf(vert, kind, dual, s, p, b, l, t, b, l, t, 8, 0);                                          \
  }                                                                                                    \
  extern "C" void hi(uint16_t *s, int p, const uint8_t *b, const uint8_t *l, const uint8_t *t, int bd) { \
    twin_lpf(vert, kind, dual, s, p, b, l, t, b, l, t, bd, 1);                                         \
  }
#define LPF_TWIN6(lo, hi, vert, kind)                                                                  \
  extern "C" void lo(uint8_t *s, int p, const uint8_t *b0, const uint8_t *l0, const uint8_t *t0, const uint8_t *b1, \
                     const uint8_t *l1, const uint8_t *t1) {                                           \
    twin_lpf(vert, kind, 1, s, p, b0, l0, t0, b1, l1, t1, 8, 0);                                       \
  }                                                                                                    \
  extern "C" void hi(uint16_t *s, int p, const uint8_t *b0, const uint8_t *l0, const uint8_t *t0, const uint8_t *b1, \
                     const uint8_t *l1, const uint8_t *t1, int bd) {                                   \
    twin_lpf(vert, kind, 1, s, p, b0, l0, t0, b1, l1, t1, bd, 1);                                      \
  }

TX_TWIN(vpx_idct4x4_1_add_hip, vpx_highbd_idct4x4_1_add_hip, 4, 1, 0)
TX_TWIN(vpx_idct4x4_16_add_hip, vpx_highbd_idct4x4_16_add_hip, 4, 16, 0)
TX_TWIN(vpx_idct8x8_1_add_hip, vpx_highbd_idct8x8_1_add_hip, 8, 1, 0)
TX_TWIN(vpx_idct8x8_12_add_hip, vpx_highbd_idct8x8_12_add_hip, 8, 12, 0)
TX_TWIN(vpx_idct8x8_64_add_hip, vpx_highbd_idct8x8_64_add_hip, 8, 64, 0)
TX_TWIN(vpx_idct16x16_1_add_hip, vpx_highbd_idct16x16_1_add_hip, 16, 1, 0)
TX_TWIN(vpx_idct16x16_10_add_hip, vpx_highbd_idct16x16_10_add_hip, 16, 10, 0)
TX_TWIN(vpx_idct16x16_38_add_hip, vpx_highbd_idct16x16_38_add_hip, 16, 38, 0)
TX_TWIN(vpx_idct16x16_256_add_hip, vpx_highbd_idct16x16_256_add_hip, 16, 256, 0)
TX_TWIN(vpx_idct32x32_1_add_hip, vpx_highbd_idct32x32_1_add_hip, 32, 1, 0)
TX_TWIN(vpx_idct32x32_34_add_hip, vpx_highbd_idct32x32_34_add_hip, 32, 34, 0)
TX_TWIN(vpx_idct32x32_135_add_hip, vpx_highbd_idct32x32_135_add_hip, 32, 135, 0)
TX_TWIN(vpx_idct32x32_1024_add_hip, vpx_highbd_idct32x32_1024_add_hip, 32, 1024, 0)
TX_TWIN(vpx_iwht4x4_1_add_hip, vpx_highbd_iwht4x4_1_add_hip, 4, 1, 1)
TX_TWIN(vpx_iwht4x4_16_add_hip, vpx_highbd_iwht4x4_16_add_hip, 4, 16, 1)
IHT_TWIN(vp9_iht4x4_16_add_hip, vp9_highbd_iht4x4_16_add_hip, 4)
IHT_TWIN(vp9_iht8x8_64_add_hip, vp9_highbd_iht8x8_64_add_hip, 8)
IHT_TWIN(vp9_iht16x16_256_add_hip, vp9_highbd_iht16x16_256_add_hip, 16)
CONV_TWIN(vpx_convolve_copy_hip, 0)
HCONV_TWIN(vpx_highbd_convolve_copy_hip, 0)
CONV_TWIN(vpx_convolve_avg_hip, 4)
HCONV_TWIN(vpx_highbd_convolve_avg_hip, 4)
CONV_TWIN(vpx_convolve8_horiz_hip, 1)
HCONV_TWIN(vpx_highbd_convolve8_horiz_hip, 1)
CONV_TWIN(vpx_convolve8_vert_hip, 2)
HCONV_TWIN(vpx_highbd_convolve8_vert_hip, 2)
CONV_TWIN(vpx_convolve8_hip, 3)
HCONV_TWIN(vpx_highbd_convolve8_hip, 3)
CONV_TWIN(vpx_convolve8_avg_horiz_hip, 5)
HCONV_TWIN(vpx_highbd_convolve8_avg_horiz_hip, 5)
CONV_TWIN(vpx_convolve8_avg_vert_hip, 6)
HCONV_TWIN(vpx_highbd_convolve8_avg_vert_hip, 6)
CONV_TWIN(vpx_convolve8_avg_hip, 7)
HCONV_TWIN(vpx_highbd_convolve8_avg_hip, 7)
CONV_TWIN(vpx_scaled_horiz_hip, 1)
CONV_TWIN(vpx_scaled_vert_hip, 2)
CONV_TWIN(vpx_scaled_2d_hip, 3)
CONV_TWIN(vpx_scaled_avg_horiz_hip, 5)
CONV_TWIN(vpx_scaled_avg_vert_hip, 6)
CONV_TWIN(vpx_scaled_avg_2d_hip, 7)
INTRA_TWIN(vpx_dc_predictor_4x4_hip, vpx_highbd_dc_predictor_4x4_hip, 0, 4)
INTRA_TWIN(vpx_dc_predictor_8x8_hip, vpx_highbd_dc_predictor_8x8_hip, 0, 8)
INTRA_TWIN(vpx_dc_predictor_16x16_hip, vpx_highbd_dc_predictor_16x16_hip, 0, 16)
INTRA_TWIN(vpx_dc_predictor_32x32_hip, vpx_highbd_dc_predictor_32x32_hip, 0, 32)
INTRA_TWIN(vpx_dc_left_predictor_4x4_hip, vpx_highbd_dc_left_predictor_4x4_hip, 11, 4)
INTRA_TWIN(vpx_dc_left_predictor_8x8_hip, vpx_highbd_dc_left_predictor_8x8_hip, 11, 8)
INTRA_TWIN(vpx_dc_left_predictor_16x16_hip, vpx_highbd_dc_left_predictor_16x16_hip, 11, 16)
INTRA_TWIN(vpx_dc_left_predictor_32x32_hip, vpx_highbd_dc_left_predictor_32x32_hip, 11, 32)
INTRA_TWIN(vpx_dc_top_predictor_4x4_hip, vpx_highbd_dc_top_predictor_4x4_hip, 12, 4)
INTRA_TWIN(vpx_dc_top_predictor_8x8_hip, vpx_highbd_dc_top_predictor_8x8_hip, 12, 8)
INTRA_TWIN(vpx_dc_top_predictor_16x16_hip, vpx_highbd_dc_top_predictor_16x16_hip, 12, 16)
INTRA_TWIN(vpx_dc_top_predictor_32x32_hip, vpx_highbd_dc_top_predictor_32x32_hip, 12, 32)
INTRA_TWIN(vpx_dc_128_predictor_4x4_hip, vpx_highbd_dc_128_predictor_4x4_hip, 10, 4)
INTRA_TWIN(vpx_dc_128_predictor_8x8_hip, vpx_highbd_dc_128_predictor_8x8_hip, 10, 8)
INTRA_TWIN(vpx_dc_128_predictor_16x16_hip, vpx_highbd_dc_128_predictor_16x16_hip, 10, 16)
INTRA_TWIN(vpx_dc_128_predictor_32x32_hip, vpx_highbd_dc_128_predictor_32x32_hip, 10, 32)
INTRA_TWIN(vpx_v_predictor_4x4_hip, vpx_highbd_v_predictor_4x4_hip, 1, 4)
INTRA_TWIN(vpx_v_predictor_8x8_hip, vpx_highbd_v_predictor_8x8_hip, 1, 8)
INTRA_TWIN(vpx_v_predictor_16x16_hip, vpx_highbd_v_predictor_16x16_hip, 1, 16)
INTRA_TWIN(vpx_v_predictor_32x32_hip, vpx_highbd_v_predictor_32x32_hip, 1, 32)
INTRA_TWIN(vpx_h_predictor_4x4_hip, vpx_highbd_h_predictor_4x4_hip, 2, 4)
INTRA_TWIN(vpx_h_predictor_8x8_hip, vpx_highbd_h_predictor_8x8_hip, 2, 8)
INTRA_TWIN(vpx_h_predictor_16x16_hip, vpx_highbd_h_predictor_16x16_hip, 2, 16)
INTRA_TWIN(vpx_h_predictor_32x32_hip, vpx_highbd_h_predictor_32x32_hip, 2, 32)
INTRA_TWIN(vpx_d45_predictor_4x4_hip, vpx_highbd_d45_predictor_4x4_hip, 3, 4)
INTRA_TWIN(vpx_d45_predictor_8x8_hip, vpx_highbd_d45_predictor_8x8_hip, 3, 8)
INTRA_TWIN(vpx_d45_predictor_16x16_hip, vpx_highbd_d45_predictor_16x16_hip, 3, 16)
INTRA_TWIN(vpx_d45_predictor_32x32_hip, vpx_highbd_d45_predictor_32x32_hip, 3, 32)
INTRA_TWIN(vpx_d135_predictor_4x4_hip, vpx_highbd_d135_predictor_4x4_hip, 4, 4)
INTRA_TWIN(vpx_d135_predictor_8x8_hip, vpx_highbd_d135_predictor_8x8_hip, 4, 8)
INTRA_TWIN(vpx_d135_predictor_16x16_hip, vpx_highbd_d135_predictor_16x16_hip, 4, 16)
INTRA_TWIN(vpx_d135_predictor_32x32_hip, vpx_highbd_d135_predictor_32x32_hip, 4, 32)
INTRA_TWIN(vpx_d117_predictor_4x4_hip, vpx_highbd_d117_predictor_4x4_hip, 5, 4)
INTRA_TWIN(vpx_d117_predictor_8x8_hip, vpx_highbd_d117_predictor_8x8_hip, 5, 8)
INTRA_TWIN(vpx_d117_predictor_16x16_hip, vpx_highbd_d117_predictor_16x16_hip, 5, 16)
INTRA_TWIN(vpx_d117_predictor_32x32_hip, vpx_highbd_d117_predictor_32x32_hip, 5, 32)
INTRA_TWIN(vpx_d153_predictor_4x4_hip, vpx_highbd_d153_predictor_4x4_hip, 6, 4)
INTRA_TWIN(vpx_d153_predictor_8x8_hip, vpx_highbd_d153_predictor_8x8_hip, 6, 8)
INTRA_TWIN(vpx_d153_predictor_16x16_hip, vpx_highbd_d153_predictor_16x16_hip, 6, 16)
INTRA_TWIN(vpx_d153_predictor_32x32_hip, vpx_highbd_d153_predictor_32x32_hip, 6, 32)
INTRA_TWIN(vpx_d207_predictor_4x4_hip, vpx_highbd_d207_predictor_4x4_hip, 7, 4)
INTRA_TWIN(vpx_d207_predictor_8x8_hip, vpx_highbd_d207_predictor_8x8_hip, 7, 8)
INTRA_TWIN(vpx_d207_predictor_16x16_hip, vpx_highbd_d207_predictor_16x16_hip, 7, 16)
INTRA_TWIN(vpx_d207_predictor_32x32_hip, vpx_highbd_d207_predictor_32x32_hip, 7, 32)
INTRA_TWIN(vpx_d63_predictor_4x4_hip, vpx_highbd_d63_predictor_4x4_hip, 8, 4)
INTRA_TWIN(vpx_d63_predictor_8x8_hip, vpx_highbd_d63_predictor_8x8_hip, 8, 8)
INTRA_TWIN(vpx_d63_predictor_16x16_hip, vpx_highbd_d63_predictor_16x16_hip, 8, 16)
INTRA_TWIN(vpx_d63_predictor_32x32_hip, vpx_highbd_d63_predictor_32x32_hip, 8, 32)
INTRA_TWIN(vpx_tm_predictor_4x4_hip, vpx_highbd_tm_predictor_4x4_hip, 9, 4)
INTRA_TWIN(vpx_tm_predictor_8x8_hip, vpx_highbd_tm_predictor_8x8_hip, 9, 8)
INTRA_TWIN(vpx_tm_predictor_16x16_hip, vpx_highbd_tm_predictor_16x16_hip, 9, 16)
INTRA_TWIN(vpx_tm_predictor_32x32_hip, vpx_highbd_tm_predictor_32x32_hip, 9, 32)
// 8-bit only (vpx_dsp_rtcd_defs.pl:46, 51, 57, 70 have no highbd counterparts)
#define INTRA_TWIN8(lo, mode) \
  extern "C" void lo(uint8_t *d, ptrdiff_t s, const uint8_t *a, const uint8_t *l) { twin_intra(mode, 4, d, s, a, l, 8, 0); }
INTRA_TWIN8(vpx_d45e_predictor_4x4_hip, 13)
INTRA_TWIN8(vpx_d63e_predictor_4x4_hip, 14)
INTRA_TWIN8(vpx_he_predictor_4x4_hip, 15)
INTRA_TWIN8(vpx_ve_predictor_4x4_hip, 16)
LPF_TWIN3(vpx_lpf_horizontal_4_hip, vpx_highbd_lpf_horizontal_4_hip, 0, 4, 0)
LPF_TWIN6(vpx_lpf_horizontal_4_dual_hip, vpx_highbd_lpf_horizontal_4_dual_hip, 0, 4)
LPF_TWIN3(vpx_lpf_horizontal_8_hip, vpx_highbd_lpf_horizontal_8_hip, 0, 8, 0)
LPF_TWIN6(vpx_lpf_horizontal_8_dual_hip, vpx_highbd_lpf_horizontal_8_dual_hip, 0, 8)
LPF_TWIN3(vpx_lpf_horizontal_16_hip, vpx_highbd_lpf_horizontal_16_hip, 0, 16, 0)
LPF_TWIN3(vpx_lpf_horizontal_16_dual_hip, vpx_highbd_lpf_horizontal_16_dual_hip, 0, 16, 1)
LPF_TWIN3(vpx_lpf_vertical_4_hip, vpx_highbd_lpf_vertical_4_hip, 1, 4, 0)
LPF_TWIN6(vpx_lpf_vertical_4_dual_hip, vpx_highbd_lpf_vertical_4_dual_hip, 1, 4)
LPF_TWIN3(vpx_lpf_vertical_8_hip, vpx_highbd_lpf_vertical_8_hip, 1, 8, 0)
LPF_TWIN6(vpx_lpf_vertical_8_dual_hip, vpx_highbd_lpf_vertical_8_dual_hip, 1, 8)
LPF_TWIN3(vpx_lpf_vertical_16_hip, vpx_highbd_lpf_vertical_16_hip, 1, 16, 0)
LPF_TWIN3(vpx_lpf_vertical_16_dual_hip, vpx_highbd_lpf_vertical_16_dual_hip, 1, 16, 1)
