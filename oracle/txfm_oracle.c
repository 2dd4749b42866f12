/*
 * txfm_oracle.c — CPU restatement of the VP9 inverse transforms.
 * TEST INFRASTRUCTURE ONLY (see vp9_oracle.h).
 *
 * Follows (paths relative to /root/reference/libvpx/):
 *   vpx_dsp/inv_txfm.c:18-94     iwht4x4 (8-bit), :1292-1371 highbd
 *   vpx_dsp/inv_txfm.c:96-131    iadst4_c, :196-269 iadst8_c, :389-555 iadst16_c
 *   vpx_dsp/inv_txfm.c:133-152   idct4_c, :271-324 idct8_c, :557-720 idct16_c,
 *                                :813-1178 idct32_c
 *   vpx_dsp/inv_txfm.c:1373-2170 highbd 1-D twins
 *   vpx_dsp/inv_txfm.c:154-194, 326-387, 722-811, 1180-1276  2-D drivers
 *   vp9/common/vp9_idct.c:20-204 hybrid drivers + eob dispatch, :208-396 highbd
 *   vpx_dsp/txfm_common.h:28-64  the 14-bit cosine / sine constants
 *
 * The DCTs are written as the recursive even/odd decomposition the reference's
 * straight-line code implements: idctN(x) = butterfly(idct(N/2)(even x),
 * oddN(odd x)); every multiply is followed by the same (x + 8192) >> 14
 * rounding and every intermediate store wraps at the width the reference's
 * variable has (int16 for the 8-bit path's step[] arrays, int32 for highbd).
 */
#include <string.h>

#include "vp9_oracle.h"

/* round(16384*cos(k*pi/64)), k = 0..32 (txfm_common.h:28-58) */
static const int C64[33] = { 16384, 16364, 16305, 16207, 16069, 15893, 15679, 15426, 15137,
                             14811, 14449, 14053, 13623, 13160, 12665, 12140, 11585, 11003,
                             10394, 9760,  9102,  8423,  7723,  7005,  6270,  5520,  4756,
                             3981,  3196,  2404,  1606,  804,   0 };
/* txfm_common.h:61-64 */
static const int S9[5] = { 0, 5283, 9929, 13377, 15212 };

static inline int64_t rs14(int64_t v) { return (v + 8192) >> 14; } /* inv_txfm.h:38-41 */
static inline int32_t w32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }
static inline int32_t w16(int64_t v) { return (int16_t)(uint16_t)(uint64_t)v; }

/* store-width of the reference's step arrays */
static inline int32_t wstep(int64_t v, int hbd) { return hbd ? w32(v) : w16(v); }

/* rotation: (a*c0 - b*c1, a*c1 + b*c0), rounded; operands already step-width */
static inline void rot(int32_t a, int32_t b, int c0, int c1, int hbd, int32_t *lo,
                       int32_t *hi) {
  *lo = wstep(rs14((int64_t)a * c0 - (int64_t)b * c1), hbd);
  *hi = wstep(rs14((int64_t)a * c1 + (int64_t)b * c0), hbd);
}

/* highbd 1-D transforms zero their output on out-of-range input
 * (inv_txfm.c:1278-1290 detect_invalid_highbd_input) */
static int hbd_invalid(const int32_t *in, int n) {
  for (int i = 0; i < n; ++i) {
    int64_t v = in[i];
    if (v < 0) v = -v;
    if (v >= (1 << 25)) return 1;
  }
  return 0;
}

/* ---- DCT core -------------------------------------------------------------
 * x[] holds the n inputs already narrowed to step width.  Result e[] has the
 * width of the reference's step arrays (caller decides the final width). */
static void idct_core(int n, const int32_t *x, int32_t *y, int hbd);

static void idct4_core(const int32_t *x, int32_t *y, int hbd) {
  /* inv_txfm.c:133-152 / :1418-1448 */
  int32_t s0, s1, s2, s3;
  /* hbd: (in0 + in2) is an int32 add; 8-bit: int16 + int16 in int */
  int64_t a = hbd ? (int64_t)w32((int64_t)x[0] + x[2]) : (int64_t)x[0] + x[2];
  int64_t b = hbd ? (int64_t)w32((int64_t)x[0] - x[2]) : (int64_t)x[0] - x[2];
  s0 = wstep(rs14(a * C64[16]), hbd);
  s1 = wstep(rs14(b * C64[16]), hbd);
  rot(x[1], x[3], C64[24], C64[8], hbd, &s2, &s3);
  y[0] = w32((int64_t)s0 + s3);
  y[1] = w32((int64_t)s1 + s2);
  y[2] = w32((int64_t)s1 - s2);
  y[3] = w32((int64_t)s0 - s3);
}

/* odd half of the 8-point IDCT: inputs x1,x3,x5,x7 -> o[0..3] so that
 * out[i] = e[i] + o[3-i], out[7-i] = e[i] - o[3-i]   (inv_txfm.c:279-323) */
static void idct8_odd(int32_t x1, int32_t x3, int32_t x5, int32_t x7, int hbd, int32_t *o) {
  int32_t a4, a7, a5, a6;
  rot(x1, x7, C64[28], C64[4], hbd, &a4, &a7);
  rot(x5, x3, C64[12], C64[20], hbd, &a5, &a6);
  int32_t b4 = wstep((int64_t)a4 + a5, hbd), b5 = wstep((int64_t)a4 - a5, hbd);
  int32_t b6 = wstep((int64_t)a7 - a6, hbd), b7 = wstep((int64_t)a6 + a7, hbd);
  int64_t d = hbd ? (int64_t)w32((int64_t)b6 - b5) : (int64_t)b6 - b5;
  int64_t s = hbd ? (int64_t)w32((int64_t)b5 + b6) : (int64_t)b5 + b6;
  o[0] = b4;
  o[1] = wstep(rs14(d * C64[16]), hbd);
  o[2] = wstep(rs14(s * C64[16]), hbd);
  o[3] = b7;
}

/* odd half of the 16-point IDCT: o[0..7] = step2[8..15] before stage 7
 * (inv_txfm.c:586-719) */
static void idct16_odd(const int32_t *x /* x1,x3,...,x15 as x[0..7] */, int hbd, int32_t *o) {
  int32_t a[8], b[8], c[8];
  /* stage 2: pairs (in1,in15) (in9,in7) (in5,in11) (in13,in3) */
  rot(x[0], x[7], C64[30], C64[2], hbd, &a[0], &a[7]);
  rot(x[4], x[3], C64[14], C64[18], hbd, &a[1], &a[6]);
  rot(x[2], x[5], C64[22], C64[10], hbd, &a[2], &a[5]);
  rot(x[6], x[1], C64[6], C64[26], hbd, &a[3], &a[4]);
  /* stage 3 */
  b[0] = wstep((int64_t)a[0] + a[1], hbd);
  b[1] = wstep((int64_t)a[0] - a[1], hbd);
  b[2] = wstep((int64_t)a[3] - a[2], hbd);
  b[3] = wstep((int64_t)a[2] + a[3], hbd);
  b[4] = wstep((int64_t)a[4] + a[5], hbd);
  b[5] = wstep((int64_t)a[4] - a[5], hbd);
  b[6] = wstep((int64_t)a[7] - a[6], hbd);
  b[7] = wstep((int64_t)a[6] + a[7], hbd);
  /* stage 4 */
  c[0] = b[0];
  c[7] = b[7];
  c[1] = wstep(rs14(-(int64_t)b[1] * C64[8] + (int64_t)b[6] * C64[24]), hbd);
  c[6] = wstep(rs14((int64_t)b[1] * C64[24] + (int64_t)b[6] * C64[8]), hbd);
  c[2] = wstep(rs14(-(int64_t)b[2] * C64[24] - (int64_t)b[5] * C64[8]), hbd);
  c[5] = wstep(rs14(-(int64_t)b[2] * C64[8] + (int64_t)b[5] * C64[24]), hbd);
  c[3] = b[3];
  c[4] = b[4];
  /* stage 5 */
  a[0] = wstep((int64_t)c[0] + c[3], hbd);
  a[1] = wstep((int64_t)c[1] + c[2], hbd);
  a[2] = wstep((int64_t)c[1] - c[2], hbd);
  a[3] = wstep((int64_t)c[0] - c[3], hbd);
  a[4] = wstep((int64_t)c[7] - c[4], hbd);
  a[5] = wstep((int64_t)c[6] - c[5], hbd);
  a[6] = wstep((int64_t)c[5] + c[6], hbd);
  a[7] = wstep((int64_t)c[4] + c[7], hbd);
  /* stage 6 */
  o[0] = a[0];
  o[1] = a[1];
  {
    int64_t d = hbd ? (int64_t)w32((int64_t)a[5] - a[2]) : (int64_t)a[5] - a[2];
    int64_t s = hbd ? (int64_t)w32((int64_t)a[2] + a[5]) : (int64_t)a[2] + a[5];
    o[2] = wstep(rs14(d * C64[16]), hbd);
    o[5] = wstep(rs14(s * C64[16]), hbd);
    d = hbd ? (int64_t)w32((int64_t)a[4] - a[3]) : (int64_t)a[4] - a[3];
    s = hbd ? (int64_t)w32((int64_t)a[3] + a[4]) : (int64_t)a[3] + a[4];
    o[3] = wstep(rs14(d * C64[16]), hbd);
    o[4] = wstep(rs14(s * C64[16]), hbd);
  }
  o[6] = a[6];
  o[7] = a[7];
}

/* odd half of the 32-point IDCT: o[0..15] = step1[16..31] before the final
 * stage (inv_txfm.c:836-1143) */
static void idct32_odd(const int32_t *x /* x1,x3,..,x31 as x[0..15] */, int hbd, int32_t *o) {
  int32_t a[16], b[16];
  /* stage 1: (1,31) (17,15) (9,23) (25,7) (5,27) (21,11) (13,19) (29,3) */
  rot(x[0], x[15], C64[31], C64[1], hbd, &a[0], &a[15]);
  rot(x[8], x[7], C64[15], C64[17], hbd, &a[1], &a[14]);
  rot(x[4], x[11], C64[23], C64[9], hbd, &a[2], &a[13]);
  rot(x[12], x[3], C64[7], C64[25], hbd, &a[3], &a[12]);
  rot(x[2], x[13], C64[27], C64[5], hbd, &a[4], &a[11]);
  rot(x[10], x[5], C64[11], C64[21], hbd, &a[5], &a[10]);
  rot(x[6], x[9], C64[19], C64[13], hbd, &a[6], &a[9]);
  rot(x[14], x[1], C64[3], C64[29], hbd, &a[7], &a[8]);
  /* stage 2 */
  for (int g = 0; g < 16; g += 4) {
    b[g + 0] = wstep((int64_t)a[g + 0] + a[g + 1], hbd);
    b[g + 1] = wstep((int64_t)a[g + 0] - a[g + 1], hbd);
    b[g + 2] = wstep((int64_t)a[g + 3] - a[g + 2], hbd);
    b[g + 3] = wstep((int64_t)a[g + 2] + a[g + 3], hbd);
  }
  /* stage 3 */
  a[0] = b[0];
  a[15] = b[15];
  a[1] = wstep(rs14(-(int64_t)b[1] * C64[4] + (int64_t)b[14] * C64[28]), hbd);
  a[14] = wstep(rs14((int64_t)b[1] * C64[28] + (int64_t)b[14] * C64[4]), hbd);
  a[2] = wstep(rs14(-(int64_t)b[2] * C64[28] - (int64_t)b[13] * C64[4]), hbd);
  a[13] = wstep(rs14(-(int64_t)b[2] * C64[4] + (int64_t)b[13] * C64[28]), hbd);
  a[3] = b[3];
  a[4] = b[4];
  a[5] = wstep(rs14(-(int64_t)b[5] * C64[20] + (int64_t)b[10] * C64[12]), hbd);
  a[10] = wstep(rs14((int64_t)b[5] * C64[12] + (int64_t)b[10] * C64[20]), hbd);
  a[6] = wstep(rs14(-(int64_t)b[6] * C64[12] - (int64_t)b[9] * C64[20]), hbd);
  a[9] = wstep(rs14(-(int64_t)b[6] * C64[20] + (int64_t)b[9] * C64[12]), hbd);
  a[7] = b[7];
  a[8] = b[8];
  a[11] = b[11];
  a[12] = b[12];
  /* stage 4 */
  b[0] = wstep((int64_t)a[0] + a[3], hbd);
  b[1] = wstep((int64_t)a[1] + a[2], hbd);
  b[2] = wstep((int64_t)a[1] - a[2], hbd);
  b[3] = wstep((int64_t)a[0] - a[3], hbd);
  b[4] = wstep((int64_t)a[7] - a[4], hbd);
  b[5] = wstep((int64_t)a[6] - a[5], hbd);
  b[6] = wstep((int64_t)a[5] + a[6], hbd);
  b[7] = wstep((int64_t)a[4] + a[7], hbd);
  b[8] = wstep((int64_t)a[8] + a[11], hbd);
  b[9] = wstep((int64_t)a[9] + a[10], hbd);
  b[10] = wstep((int64_t)a[9] - a[10], hbd);
  b[11] = wstep((int64_t)a[8] - a[11], hbd);
  b[12] = wstep((int64_t)a[15] - a[12], hbd);
  b[13] = wstep((int64_t)a[14] - a[13], hbd);
  b[14] = wstep((int64_t)a[13] + a[14], hbd);
  b[15] = wstep((int64_t)a[12] + a[15], hbd);
  /* stage 5 */
  a[0] = b[0];
  a[1] = b[1];
  a[2] = wstep(rs14(-(int64_t)b[2] * C64[8] + (int64_t)b[13] * C64[24]), hbd);
  a[13] = wstep(rs14((int64_t)b[2] * C64[24] + (int64_t)b[13] * C64[8]), hbd);
  a[3] = wstep(rs14(-(int64_t)b[3] * C64[8] + (int64_t)b[12] * C64[24]), hbd);
  a[12] = wstep(rs14((int64_t)b[3] * C64[24] + (int64_t)b[12] * C64[8]), hbd);
  a[4] = wstep(rs14(-(int64_t)b[4] * C64[24] - (int64_t)b[11] * C64[8]), hbd);
  a[11] = wstep(rs14(-(int64_t)b[4] * C64[8] + (int64_t)b[11] * C64[24]), hbd);
  a[5] = wstep(rs14(-(int64_t)b[5] * C64[24] - (int64_t)b[10] * C64[8]), hbd);
  a[10] = wstep(rs14(-(int64_t)b[5] * C64[8] + (int64_t)b[10] * C64[24]), hbd);
  a[6] = b[6];
  a[7] = b[7];
  a[8] = b[8];
  a[9] = b[9];
  a[14] = b[14];
  a[15] = b[15];
  /* stage 6 */
  for (int i = 0; i < 4; ++i) {
    b[i] = wstep((int64_t)a[i] + a[7 - i], hbd);
    b[7 - i] = wstep((int64_t)a[i] - a[7 - i], hbd);
    b[8 + i] = wstep((int64_t)a[15 - i] - a[8 + i], hbd);
    b[15 - i] = wstep((int64_t)a[8 + i] + a[15 - i], hbd);
  }
  /* stage 7 */
  o[0] = b[0];
  o[1] = b[1];
  o[2] = b[2];
  o[3] = b[3];
  for (int i = 4; i < 8; ++i) {
    int64_t d = hbd ? (int64_t)w32((int64_t)b[15 - i] - b[i]) : (int64_t)b[15 - i] - b[i];
    int64_t s = hbd ? (int64_t)w32((int64_t)b[i] + b[15 - i]) : (int64_t)b[i] + b[15 - i];
    o[i] = wstep(rs14(d * C64[16]), hbd);
    o[15 - i] = wstep(rs14(s * C64[16]), hbd);
  }
  o[12] = b[12];
  o[13] = b[13];
  o[14] = b[14];
  o[15] = b[15];
}

static void idct_core(int n, const int32_t *x, int32_t *y, int hbd) {
  if (n == 4) {
    idct4_core(x, y, hbd);
    return;
  }
  int h = n / 2;
  int32_t ev[16], od[16], e[16], o[16];
  for (int i = 0; i < h; ++i) {
    ev[i] = x[2 * i];
    od[i] = x[2 * i + 1];
  }
  idct_core(h, ev, e, hbd);
  /* the embedded half-size transform's outputs live in step arrays */
  for (int i = 0; i < h; ++i) e[i] = wstep(e[i], hbd);
  if (n == 8)
    idct8_odd(od[0], od[1], od[2], od[3], hbd, o);
  else if (n == 16)
    idct16_odd(od, hbd, o);
  else
    idct32_odd(od, hbd, o);
  for (int i = 0; i < h; ++i) {
    y[i] = w32((int64_t)e[i] + o[h - 1 - i]);
    y[n - 1 - i] = w32((int64_t)e[i] - o[h - 1 - i]);
  }
}

void vp9o_idct1d(int n, const int32_t *in, int32_t *out, int hbd) {
  int32_t x[32];
  if (hbd) {
    if (hbd_invalid(in, n)) { /* inv_txfm.c:1428, 1584, 1888, 2177 */
      memset(out, 0, sizeof(*out) * n);
      return;
    }
    for (int i = 0; i < n; ++i) x[i] = in[i];
  } else {
    for (int i = 0; i < n; ++i) x[i] = w16(in[i]); /* (int16_t)input[k] */
  }
  idct_core(n, x, out, hbd);
}

/* ---- ADST ----------------------------------------------------------------- */

static void iadst4(const int32_t *in, int32_t *out, int hbd) {
  /* inv_txfm.c:96-131 / :1373-1416 */
  int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
  if (!(x0 | x1 | x2 | x3)) {
    memset(out, 0, 4 * sizeof(*out));
    return;
  }
  int64_t s0, s1, s2, s3, s4, s5, s6, s7;
  if (hbd) {
    s0 = (int64_t)S9[1] * x0;
    s1 = (int64_t)S9[2] * x0;
    s2 = (int64_t)S9[3] * x1;
    s3 = (int64_t)S9[4] * x2;
    s4 = (int64_t)S9[1] * x2;
    s5 = (int64_t)S9[2] * x3;
    s6 = (int64_t)S9[4] * x3;
  } else { /* int * int32 products in the 8-bit function */
    s0 = w32((int64_t)S9[1] * x0);
    s1 = w32((int64_t)S9[2] * x0);
    s2 = w32((int64_t)S9[3] * x1);
    s3 = w32((int64_t)S9[4] * x2);
    s4 = w32((int64_t)S9[1] * x2);
    s5 = w32((int64_t)S9[2] * x3);
    s6 = w32((int64_t)S9[4] * x3);
  }
  s7 = w32((int64_t)w32((int64_t)x0 - x2) + x3);
  s0 = s0 + s3 + s5;
  s1 = s1 - s4 - s6;
  s3 = s2;
  s2 = (int64_t)S9[3] * s7;
  out[0] = w32(rs14(s0 + s3));
  out[1] = w32(rs14(s1 + s3));
  out[2] = w32(rs14(s2));
  out[3] = w32(rs14(s0 + s1 - s3));
}

/* 8-bit iadst8_c keeps its s-terms in `int` (inv_txfm.c:197); highbd in int64 */
static inline int64_t sterm(int64_t v, int hbd) { return hbd ? v : (int64_t)w32(v); }

static void iadst8(const int32_t *in, int32_t *out, int hbd) {
  /* inv_txfm.c:196-269 / :1496-1578 */
  int64_t x0 = in[7], x1 = in[0], x2 = in[5], x3 = in[2];
  int64_t x4 = in[3], x5 = in[4], x6 = in[1], x7 = in[6];
  if (!(x0 | x1 | x2 | x3 | x4 | x5 | x6 | x7)) {
    memset(out, 0, 8 * sizeof(*out));
    return;
  }
  int64_t s0, s1, s2, s3, s4, s5, s6, s7;
  /* stage 1 */
  s0 = sterm(C64[2] * x0 + C64[30] * x1, hbd);
  s1 = sterm(C64[30] * x0 - C64[2] * x1, hbd);
  s2 = sterm(C64[10] * x2 + C64[22] * x3, hbd);
  s3 = sterm(C64[22] * x2 - C64[10] * x3, hbd);
  s4 = sterm(C64[18] * x4 + C64[14] * x5, hbd);
  s5 = sterm(C64[14] * x4 - C64[18] * x5, hbd);
  s6 = sterm(C64[26] * x6 + C64[6] * x7, hbd);
  s7 = sterm(C64[6] * x6 - C64[26] * x7, hbd);
  /* 8-bit: s0 + s4 is an int add */
  x0 = w32(rs14(sterm(s0 + s4, hbd)));
  x1 = w32(rs14(sterm(s1 + s5, hbd)));
  x2 = w32(rs14(sterm(s2 + s6, hbd)));
  x3 = w32(rs14(sterm(s3 + s7, hbd)));
  x4 = w32(rs14(sterm(s0 - s4, hbd)));
  x5 = w32(rs14(sterm(s1 - s5, hbd)));
  x6 = w32(rs14(sterm(s2 - s6, hbd)));
  x7 = w32(rs14(sterm(s3 - s7, hbd)));
  /* stage 2 */
  s0 = x0;
  s1 = x1;
  s2 = x2;
  s3 = x3;
  s4 = sterm(C64[8] * x4 + C64[24] * x5, hbd);
  s5 = sterm(C64[24] * x4 - C64[8] * x5, hbd);
  s6 = sterm(-C64[24] * x6 + C64[8] * x7, hbd);
  s7 = sterm(C64[8] * x6 + C64[24] * x7, hbd);
  x0 = w32(sterm(s0 + s2, hbd));
  x1 = w32(sterm(s1 + s3, hbd));
  x2 = w32(sterm(s0 - s2, hbd));
  x3 = w32(sterm(s1 - s3, hbd));
  x4 = w32(rs14(sterm(s4 + s6, hbd)));
  x5 = w32(rs14(sterm(s5 + s7, hbd)));
  x6 = w32(rs14(sterm(s4 - s6, hbd)));
  x7 = w32(rs14(sterm(s5 - s7, hbd)));
  /* stage 3: highbd adds x2 + x3 as int32 (tran_low_t), 8-bit as int64 */
  {
    int64_t p23 = hbd ? (int64_t)w32(x2 + x3) : x2 + x3;
    int64_t m23 = hbd ? (int64_t)w32(x2 - x3) : x2 - x3;
    int64_t p67 = hbd ? (int64_t)w32(x6 + x7) : x6 + x7;
    int64_t m67 = hbd ? (int64_t)w32(x6 - x7) : x6 - x7;
    s2 = sterm(C64[16] * p23, hbd);
    s3 = sterm(C64[16] * m23, hbd);
    s6 = sterm(C64[16] * p67, hbd);
    s7 = sterm(C64[16] * m67, hbd);
  }
  x2 = w32(rs14(s2));
  x3 = w32(rs14(s3));
  x6 = w32(rs14(s6));
  x7 = w32(rs14(s7));
  out[0] = w32(x0);
  out[1] = w32(-x4);
  out[2] = w32(x6);
  out[3] = w32(-x2);
  out[4] = w32(x3);
  out[5] = w32(-x7);
  out[6] = w32(x5);
  out[7] = w32(-x1);
}

static void iadst16(const int32_t *in, int32_t *out, int hbd) {
  /* inv_txfm.c:389-555 / :1706-1881.  Both variants keep s-terms in 64 bit;
   * the highbd one holds x in int32 so x+x / -x wrap at 32 bits. */
  int64_t x[16], s[16];
  static const int perm[16] = { 15, 0, 13, 2, 11, 4, 9, 6, 7, 8, 5, 10, 3, 12, 1, 14 };
  int64_t any = 0;
  for (int i = 0; i < 16; ++i) {
    x[i] = in[perm[i]];
    any |= x[i];
  }
  if (!any) {
    memset(out, 0, 16 * sizeof(*out));
    return;
  }
#define XW(v) (hbd ? (int64_t)w32(v) : (int64_t)(v))
  /* stage 1 */
  static const int c1[8] = { 1, 5, 9, 13, 17, 21, 25, 29 };
  for (int i = 0; i < 8; ++i) {
    s[2 * i] = x[2 * i] * C64[c1[i]] + x[2 * i + 1] * C64[32 - c1[i]];
    s[2 * i + 1] = x[2 * i] * C64[32 - c1[i]] - x[2 * i + 1] * C64[c1[i]];
  }
  for (int i = 0; i < 8; ++i) {
    x[i] = w32(rs14(s[i] + s[i + 8]));
    x[i + 8] = w32(rs14(s[i] - s[i + 8]));
  }
  /* stage 2 */
  for (int i = 0; i < 8; ++i) s[i] = x[i];
  s[8] = x[8] * C64[4] + x[9] * C64[28];
  s[9] = x[8] * C64[28] - x[9] * C64[4];
  s[10] = x[10] * C64[20] + x[11] * C64[12];
  s[11] = x[10] * C64[12] - x[11] * C64[20];
  s[12] = XW(-x[12]) * C64[28] + x[13] * C64[4];
  s[13] = x[12] * C64[4] + x[13] * C64[28];
  s[14] = XW(-x[14]) * C64[12] + x[15] * C64[20];
  s[15] = x[14] * C64[20] + x[15] * C64[12];
  for (int i = 0; i < 4; ++i) {
    x[i] = w32(s[i] + s[i + 4]);
    x[i + 4] = w32(s[i] - s[i + 4]);
    x[i + 8] = w32(rs14(s[i + 8] + s[i + 12]));
    x[i + 12] = w32(rs14(s[i + 8] - s[i + 12]));
  }
  /* stage 3 */
  for (int g = 0; g < 16; g += 8) {
    s[g + 0] = x[g + 0];
    s[g + 1] = x[g + 1];
    s[g + 2] = x[g + 2];
    s[g + 3] = x[g + 3];
    s[g + 4] = x[g + 4] * C64[8] + x[g + 5] * C64[24];
    s[g + 5] = x[g + 4] * C64[24] - x[g + 5] * C64[8];
    s[g + 6] = XW(-x[g + 6]) * C64[24] + x[g + 7] * C64[8];
    s[g + 7] = x[g + 6] * C64[8] + x[g + 7] * C64[24];
    x[g + 0] = w32(s[g + 0] + s[g + 2]);
    x[g + 1] = w32(s[g + 1] + s[g + 3]);
    x[g + 2] = w32(s[g + 0] - s[g + 2]);
    x[g + 3] = w32(s[g + 1] - s[g + 3]);
    x[g + 4] = w32(rs14(s[g + 4] + s[g + 6]));
    x[g + 5] = w32(rs14(s[g + 5] + s[g + 7]));
    x[g + 6] = w32(rs14(s[g + 4] - s[g + 6]));
    x[g + 7] = w32(rs14(s[g + 5] - s[g + 7]));
  }
  /* stage 4 */
  s[2] = (int64_t)(-C64[16]) * XW(x[2] + x[3]);
  s[3] = (int64_t)C64[16] * XW(x[2] - x[3]);
  s[6] = (int64_t)C64[16] * XW(x[6] + x[7]);
  s[7] = (int64_t)C64[16] * XW(XW(-x[6]) + x[7]);
  s[10] = (int64_t)C64[16] * XW(x[10] + x[11]);
  s[11] = (int64_t)C64[16] * XW(XW(-x[10]) + x[11]);
  s[14] = (int64_t)(-C64[16]) * XW(x[14] + x[15]);
  s[15] = (int64_t)C64[16] * XW(x[14] - x[15]);
  x[2] = w32(rs14(s[2]));
  x[3] = w32(rs14(s[3]));
  x[6] = w32(rs14(s[6]));
  x[7] = w32(rs14(s[7]));
  x[10] = w32(rs14(s[10]));
  x[11] = w32(rs14(s[11]));
  x[14] = w32(rs14(s[14]));
  x[15] = w32(rs14(s[15]));
#undef XW
  out[0] = w32(x[0]);
  out[1] = w32(-x[8]);
  out[2] = w32(x[12]);
  out[3] = w32(-x[4]);
  out[4] = w32(x[6]);
  out[5] = w32(x[14]);
  out[6] = w32(x[10]);
  out[7] = w32(x[2]);
  out[8] = w32(x[3]);
  out[9] = w32(x[11]);
  out[10] = w32(x[15]);
  out[11] = w32(x[7]);
  out[12] = w32(x[5]);
  out[13] = w32(-x[13]);
  out[14] = w32(x[9]);
  out[15] = w32(-x[1]);
}

void vp9o_iadst1d(int n, const int32_t *in, int32_t *out, int hbd) {
  if (hbd && hbd_invalid(in, n)) {
    memset(out, 0, sizeof(*out) * n);
    return;
  }
  if (n == 4)
    iadst4(in, out, hbd);
  else if (n == 8)
    iadst8(in, out, hbd);
  else
    iadst16(in, out, hbd);
}

/* ---- WHT (lossless) --------------------------------------------------------
 * inv_txfm.c:18-94 / :1292-1371.  a1..e1 are tran_high_t (int64). */
static void iwht4x4_residual(const int32_t *in, int32_t *res) {
  int32_t tmp[16];
  for (int i = 0; i < 4; ++i) {
    int64_t a = in[4 * i + 0] >> 2, c = in[4 * i + 1] >> 2;
    int64_t d = in[4 * i + 2] >> 2, b = in[4 * i + 3] >> 2, e;
    a += c;
    d -= b;
    e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= b;
    d += c;
    tmp[4 * i + 0] = w32(a);
    tmp[4 * i + 1] = w32(b);
    tmp[4 * i + 2] = w32(c);
    tmp[4 * i + 3] = w32(d);
  }
  for (int i = 0; i < 4; ++i) {
    int64_t a = tmp[i], c = tmp[4 + i], d = tmp[8 + i], b = tmp[12 + i], e;
    a += c;
    d -= b;
    e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= b;
    d += c;
    res[0 + i] = w32(a);
    res[4 + i] = w32(b);
    res[8 + i] = w32(c);
    res[12 + i] = w32(d);
  }
}

/* ---- 2-D ------------------------------------------------------------------- */

void vp9o_inv_txfm_residual(int n, int tx_type, int lossless, int hbd, const int32_t *coeffs,
                            int32_t *res) {
  if (lossless) {
    iwht4x4_residual(coeffs, res);
    return;
  }
  /* rows use DCT unless tx_type has ADST in the horizontal direction
   * (vp9_idct.c:22-27: {cols, rows} = {iadst, idct} for ADST_DCT) */
  const int row_adst = (tx_type == VP9O_DCT_ADST || tx_type == VP9O_ADST_ADST) && n < 32;
  const int col_adst = (tx_type == VP9O_ADST_DCT || tx_type == VP9O_ADST_ADST) && n < 32;
  const int shift = n == 4 ? 4 : n == 8 ? 5 : 6;
  int32_t tmp[32 * 32], cin[32], cout[32];
  for (int r = 0; r < n; ++r) {
    if (row_adst)
      vp9o_iadst1d(n, coeffs + r * n, tmp + r * n, hbd);
    else
      vp9o_idct1d(n, coeffs + r * n, tmp + r * n, hbd);
  }
  for (int c = 0; c < n; ++c) {
    for (int r = 0; r < n; ++r) cin[r] = tmp[r * n + c];
    if (col_adst)
      vp9o_iadst1d(n, cin, cout, hbd);
    else
      vp9o_idct1d(n, cin, cout, hbd);
    /* ROUND_POWER_OF_TWO on an int32 (mem.h:31) */
    for (int r = 0; r < n; ++r) res[r * n + c] = w32((int64_t)cout[r] + (1 << (shift - 1))) >> shift;
  }
}

static inline uint8_t clip8(int v) { return v > 255 ? 255 : v < 0 ? 0 : v; }
static inline uint16_t cliphbd(int v, int bd) {
  int mx = (1 << bd) - 1;
  return v > mx ? mx : v < 0 ? 0 : v;
}

/* DC-only shortcut: vpx_idctNxN_1_add_c (inv_txfm.c:178-194, 373-387, 799-811,
 * 1262-1276), highbd (:1476-1494 ...) */
static int32_t dc_only_value(int n, int32_t dc, int hbd) {
  const int shift = n == 4 ? 4 : n == 8 ? 5 : 6;
  int32_t out;
  if (hbd) {
    out = w32(rs14((int64_t)dc * C64[16]));
    out = w32(rs14((int64_t)out * C64[16]));
  } else {
    out = w32(rs14((int64_t)w16(dc) * C64[16]));
    out = w32(rs14(w32((int64_t)out * C64[16]))); /* int * int16 product */
  }
  return w32((int64_t)out + (1 << (shift - 1))) >> shift;
}

/* WHT DC-only: vpx_iwht4x4_1_add_c (inv_txfm.c:71-94) */
static void iwht_dc_residual(int32_t dc, int32_t *res) {
  int64_t a1 = dc >> 2, e1 = a1 >> 1;
  a1 -= e1;
  int32_t t0 = w32(a1), t1 = w32(e1);
  for (int i = 0; i < 4; ++i) {
    int32_t ip = i == 0 ? t0 : t1;
    int64_t e = ip >> 1, a = ip - e;
    res[0 + i] = w32(a);
    res[4 + i] = w32(e);
    res[8 + i] = w32(e);
    res[12 + i] = w32(e);
  }
}

static void residual_for_eob(int n, int tx_type, int lossless, int hbd, const int32_t *coeffs,
                             int eob, int32_t *res) {
  /* vp9_idct.c:119-204: DCT_DCT takes the eob shortcuts; hybrid types always
   * run the full transform.  The intermediate-eob variants (_12/_10/_38/_34/
   * _135) equal the full transform on their own domain (upper-left-only
   * coefficients) — test/partial_idct_test.cc — so only eob<=1 is special. */
  if (lossless) {
    if (eob > 1)
      iwht4x4_residual(coeffs, res);
    else
      iwht_dc_residual(coeffs[0], res);
    return;
  }
  if (tx_type == VP9O_DCT_DCT || n == 32) {
    int dc_only = (n == 4) ? !(eob > 1) : (eob == 1);
    if (dc_only) {
      int32_t v = dc_only_value(n, coeffs[0], hbd);
      for (int i = 0; i < n * n; ++i) res[i] = v;
      return;
    }
  }
  vp9o_inv_txfm_residual(n, tx_type, 0, hbd, coeffs, res);
}

void vp9o_inv_txfm_add(int n, int tx_type, int lossless, const int32_t *coeffs, uint8_t *dest,
                       int stride, int eob) {
  int32_t res[32 * 32];
  residual_for_eob(n, tx_type, lossless, 0, coeffs, eob, res);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c)
      dest[r * stride + c] = clip8(w32((int64_t)dest[r * stride + c] + res[r * n + c]));
}

void vp9o_highbd_inv_txfm_add(int n, int tx_type, int lossless, const int32_t *coeffs,
                              uint16_t *dest, int stride, int eob, int bd) {
  int32_t res[32 * 32];
  residual_for_eob(n, tx_type, lossless, 1, coeffs, eob, res);
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < n; ++c)
      dest[r * stride + c] = cliphbd(w32((int64_t)dest[r * stride + c] + res[r * n + c]), bd);
}
