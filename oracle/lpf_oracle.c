/*
 * lpf_oracle.c — CPU restatement of the VP9 loop-filter kernels.
 * TEST INFRASTRUCTURE ONLY (see vp9_oracle.h).
 *
 * Follows (relative to /root/reference/libvpx/):
 *   vpx_dsp/loopfilter.c:33-110    filter_mask / flat_mask4 / flat_mask5 / hev_mask / filter4
 *   vpx_dsp/loopfilter.c:162-180   filter8,  :235-287 filter16
 *   vpx_dsp/loopfilter.c:112-357   the 12 entry points (8-bit)
 *   vpx_dsp/loopfilter.c:359-743   highbd twins: thresholds << (bd-8), int16 lanes
 *
 * One routine filters ONE line of samples across an edge: px[-8..7] addressed
 * as base[k*step]; the entry points are loops of that routine over 8 or 16
 * lines.  bd = 8 reproduces the 8-bit functions exactly (the int8 "^0x80"
 * arithmetic is the bd=8 case of the highbd "- (0x80 << shift)" arithmetic).
 */
#include <stdlib.h>

#include "vp9_oracle.h"

typedef struct {
  uint8_t *p8;
  uint16_t *p16;
} pixptr;

static inline int ld(pixptr b, ptrdiff_t i) { return b.p16 ? b.p16[i] : b.p8[i]; }
static inline void st(pixptr b, ptrdiff_t i, int v) {
  if (b.p16)
    b.p16[i] = (uint16_t)v;
  else
    b.p8[i] = (uint8_t)v;
}

static inline int sclamp(int t, int bd) {
  const int lo = -(128 << (bd - 8)), hi = (128 << (bd - 8)) - 1;
  return t < lo ? lo : t > hi ? hi : t;
}

/* narrow 4-tap filter (loopfilter.c:76-110 / :410-447) */
static void narrow(pixptr b, ptrdiff_t step, int mask, int thresh, int bd) {
  const int off = 0x80 << (bd - 8);
  const int p1 = ld(b, -2 * step), p0 = ld(b, -step), q0 = ld(b, 0), q1 = ld(b, step);
  const int t16 = thresh << (bd - 8);
  const int hev = (abs(p1 - p0) > t16 || abs(q1 - q0) > t16) ? -1 : 0;
  const int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
  int f = sclamp(ps1 - qs1, bd) & hev;
  f = sclamp(f + 3 * (qs0 - ps0), bd) & mask;
  const int f1 = sclamp(f + 4, bd) >> 3;
  const int f2 = sclamp(f + 3, bd) >> 3;
  st(b, 0, sclamp(qs0 - f1, bd) + off);
  st(b, -step, sclamp(ps0 + f2, bd) + off);
  f = ((f1 + 1) >> 1) & ~hev;
  st(b, step, sclamp(qs1 - f, bd) + off);
  st(b, -2 * step, sclamp(ps1 + f, bd) + off);
}

static void filter_line(pixptr b, ptrdiff_t step, int kind, int blimit, int limit, int thresh,
                        int bd) {
  const int sh = bd - 8;
  const int lim = limit << sh, blim = blimit << sh, one = 1 << sh;
  int p[8], q[8];
  for (int k = 0; k < (kind == 16 ? 8 : 4); ++k) {
    p[k] = ld(b, -(k + 1) * step);
    q[k] = ld(b, k * step);
  }
  /* filter_mask (loopfilter.c:33-47) */
  int mask = -1;
  if (abs(p[3] - p[2]) > lim || abs(p[2] - p[1]) > lim || abs(p[1] - p[0]) > lim ||
      abs(q[1] - q[0]) > lim || abs(q[2] - q[1]) > lim || abs(q[3] - q[2]) > lim ||
      abs(p[0] - q[0]) * 2 + abs(p[1] - q[1]) / 2 > blim)
    mask = 0;
  if (kind == 4) {
    narrow(b, step, mask, thresh, bd);
    return;
  }
  /* flat_mask4 with thresh 1 (:49-60) */
  const int flat = !(abs(p[1] - p[0]) > one || abs(q[1] - q[0]) > one || abs(p[2] - p[0]) > one ||
                     abs(q[2] - q[0]) > one || abs(p[3] - p[0]) > one || abs(q[3] - q[0]) > one);
  int flat2 = 0;
  if (kind == 16) /* flat_mask5 on p4..p7/q4..q7 against p0/q0 (:62-70, :300-303) */
    flat2 = !(abs(p[4] - p[0]) > one || abs(q[4] - q[0]) > one || abs(p[5] - p[0]) > one ||
              abs(q[5] - q[0]) > one || abs(p[6] - p[0]) > one || abs(q[6] - q[0]) > one ||
              abs(p[7] - p[0]) > one || abs(q[7] - q[0]) > one);
  if (kind == 16 && flat2 && flat && mask) {
    /* 15-tap [1 1 1 1 1 1 1 2 1 1 1 1 1 1 1] with edge replication (:235-283):
     * out at position i (0 = p6 .. 13 = q6) = sum over the 15-wide window
     * centred on i of the line clamped to [p7, q7], centre counted twice. */
    int line[16], out[14];
    for (int k = 0; k < 8; ++k) {
      line[7 - k] = p[k];
      line[8 + k] = q[k];
    }
    for (int i = 1; i <= 14; ++i) {
      int s = line[i];
      for (int j = i - 7; j <= i + 7; ++j) s += line[j < 0 ? 0 : j > 15 ? 15 : j];
      out[i - 1] = (s + 8) >> 4;
    }
    for (int i = 1; i <= 14; ++i) st(b, (i - 8) * step, out[i - 1]);
    return;
  }
  if (flat && mask) {
    /* 7-tap [1 1 1 2 1 1 1] with replication to p3/q3 (:165-176) */
    int line[8], out[6];
    for (int k = 0; k < 4; ++k) {
      line[3 - k] = p[k];
      line[4 + k] = q[k];
    }
    for (int i = 1; i <= 6; ++i) {
      int s = line[i];
      for (int j = i - 3; j <= i + 3; ++j) s += line[j < 0 ? 0 : j > 7 ? 7 : j];
      out[i - 1] = (s + 4) >> 3;
    }
    for (int i = 1; i <= 6; ++i) st(b, (i - 4) * step, out[i - 1]);
    return;
  }
  narrow(b, step, mask, thresh, bd);
}

static void run(pixptr s, int pitch, int vertical, int kind, int lines, const uint8_t *b,
                const uint8_t *l, const uint8_t *t, int bd) {
  /* horizontal edge: taps step by pitch, lines advance by 1; vertical: swap */
  const ptrdiff_t step = vertical ? 1 : pitch, adv = vertical ? pitch : 1;
  for (int i = 0; i < lines; ++i) {
    pixptr c = { s.p8 ? s.p8 + i * adv : NULL, s.p16 ? s.p16 + i * adv : NULL };
    filter_line(c, step, kind, *b, *l, *t, bd);
  }
}

static void lpf_any(pixptr s, int vertical, int kind, int dual, int pitch, const uint8_t *b0,
                    const uint8_t *l0, const uint8_t *t0, const uint8_t *b1, const uint8_t *l1,
                    const uint8_t *t1, int bd) {
  if (kind == 16) { /* _16_dual shares one threshold set (:323-327, :351-357) */
    run(s, pitch, vertical, 16, dual ? 16 : 8, b0, l0, t0, bd);
    return;
  }
  run(s, pitch, vertical, kind, 8, b0, l0, t0, bd);
  if (dual) {
    const ptrdiff_t off = vertical ? 8 * (ptrdiff_t)pitch : 8;
    pixptr s2 = { s.p8 ? s.p8 + off : NULL, s.p16 ? s.p16 + off : NULL };
    run(s2, pitch, vertical, kind, 8, b1, l1, t1, bd);
  }
}

void vp9o_lpf(int vertical, int kind, int dual, uint8_t *s, int pitch, const uint8_t *b0,
              const uint8_t *l0, const uint8_t *t0, const uint8_t *b1, const uint8_t *l1,
              const uint8_t *t1) {
  pixptr p = { s, NULL };
  lpf_any(p, vertical, kind, dual, pitch, b0, l0, t0, b1, l1, t1, 8);
}

void vp9o_highbd_lpf(int vertical, int kind, int dual, uint16_t *s, int pitch, const uint8_t *b0,
                     const uint8_t *l0, const uint8_t *t0, const uint8_t *b1, const uint8_t *l1,
                     const uint8_t *t1, int bd) {
  pixptr p = { NULL, s };
  lpf_any(p, vertical, kind, dual, pitch, b0, l0, t0, b1, l1, t1, bd);
}
