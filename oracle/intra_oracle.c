/*
 * intra_oracle.c — CPU restatement of the VP9 intra predictors and their edge
 * builder.  TEST INFRASTRUCTURE ONLY (see vp9_oracle.h).
 *
 * Follows (relative to /root/reference/libvpx/):
 *   vpx_dsp/intrapred.c:21-248    generic predictors (bs = 8,16,32; v/h/tm/dc* all sizes)
 *   vpx_dsp/intrapred.c:286-454   4x4 specials (d207,d63,d45,d117,d135,d153)
 *   vpx_dsp/intrapred.c:457-850   highbd twins (identical arithmetic on uint16)
 *   vp9/common/vp9_reconintra.c:40-112  mode -> needed edges, predictor tables
 *   vp9/common/vp9_reconintra.c:262-402 build_intra_predictors (8-bit)
 *   vp9/common/vp9_reconintra.c:113-259 build_intra_predictors_high
 *
 * Every predictor is written as a closed form per output pixel P(r,c) in terms
 * of A[i] = above[i] (i >= -1) and L[r] = left[r]; the derivations are noted at
 * each mode.  Pixels are handled as uint16 internally so one body serves both
 * bit depths.
 */
#include <string.h>

#include "vp9_oracle.h"

#define AVG2(a, b) (((a) + (b) + 1) >> 1)
#define AVG3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

static inline int clipmax(int v, int mx) { return v < 0 ? 0 : v > mx ? mx : v; }

/* A points at above[0]; A[-1] valid.  out is bs*bs row-major. */
static void predict_core(int mode, int bs, const uint16_t *A, const uint16_t *L, int bd,
                         uint16_t *out) {
  const int mx = (1 << bd) - 1;
  int r, c;
  switch (mode) {
    case VP9O_V_PRED: /* intrapred.c:158-168 */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) out[r * bs + c] = A[c];
      break;
    case VP9O_H_PRED: /* :170-180 */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) out[r * bs + c] = L[r];
      break;
    case VP9O_TM_PRED: /* :182-193 */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) out[r * bs + c] = (uint16_t)clipmax(L[r] + A[c] - A[-1], mx);
      break;
    case VP9O_DC_128: /* :195-206, highbd: 128 << (bd - 8) */
      for (r = 0; r < bs * bs; ++r) out[r] = (uint16_t)(128 << (bd - 8));
      break;
    case VP9O_DC_LEFT:
    case VP9O_DC_TOP:
    case VP9O_DC_PRED: { /* :208-262 */
      int sum = 0, cnt = 0;
      if (mode != VP9O_DC_LEFT) {
        for (c = 0; c < bs; ++c) sum += A[c];
        cnt += bs;
      }
      if (mode != VP9O_DC_TOP) {
        for (r = 0; r < bs; ++r) sum += L[r];
        cnt += bs;
      }
      const int dc = (sum + (cnt >> 1)) / cnt;
      for (r = 0; r < bs * bs; ++r) out[r] = (uint16_t)dc;
      break;
    }
    case VP9O_D45_PRED:
      /* generic (:65-81): row r is row 0 shifted left by r, tail filled with
       * A[bs-1]; row 0 = AVG3(A[x],A[x+1],A[x+2]) for x < bs-1, A[bs-1] at bs-1.
       * 4x4 (:354-373): pure diagonal over A[0..7], corner = A[7]. */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int i = r + c;
          int v;
          if (bs == 4)
            v = (i == 6) ? A[7] : AVG3(A[i], A[i + 1], A[i + 2]);
          else
            v = (i < bs - 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : A[bs - 1];
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    case VP9O_D63_PRED:
      /* generic (:47-63): rows 0/1 are AVG2/AVG3 of A at c; rows 2k/2k+1 are
       * those shifted left by k with the last k+1 entries replaced by A[bs-1].
       * 4x4 (:308-329): same shift, no replacement. */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int k = r >> 1, i = c + k;
          int v;
          if (bs != 4 && r >= 2 && c >= bs - 1 - k)
            v = A[bs - 1];
          else
            v = (r & 1) ? AVG3(A[i], A[i + 1], A[i + 2]) : AVG2(A[i], A[i + 1]);
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    case VP9O_D207_PRED:
      /* :21-45 / :293-306: P(r,c) = P(r+1,c-2); columns 0/1 are AVG2/AVG3 down
       * the left edge, everything at or past the last row is L[bs-1]. */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int i = r + (c >> 1);
          int v;
          if (i >= bs - 1)
            v = L[bs - 1];
          else if (!(c & 1))
            v = AVG2(L[i], L[i + 1]);
          else
            v = AVG3(L[i], L[i + 1], L[i + 2 < bs ? i + 2 : bs - 1]);
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    case VP9O_D117_PRED:
      /* :83-107 / :395-415: P(r,c) = P(r-2,c-1); seeds are rows 0,1 and col 0. */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int k = (r >> 1) < c ? (r >> 1) : c;
          const int rr = r - 2 * k, cc = c - k;
          int v;
          if (rr == 0)
            v = AVG2(A[cc - 1], A[cc]);
          else if (rr == 1)
            v = cc == 0 ? AVG3(L[0], A[-1], A[0]) : AVG3(A[cc - 2], A[cc - 1], A[cc]);
          else if (rr == 2)
            v = AVG3(A[-1], L[0], L[1]);
          else
            v = AVG3(L[rr - 3], L[rr - 2], L[rr - 1]);
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    case VP9O_D135_PRED:
      /* :109-139 / :417-436: constant along d = c - r */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int d = c - r;
          int v;
          if (d >= 2)
            v = AVG3(A[d - 2], A[d - 1], A[d]);
          else if (d == 1)
            v = AVG3(A[-1], A[0], A[1]);
          else if (d == 0)
            v = AVG3(L[0], A[-1], A[0]);
          else if (d == -1)
            v = AVG3(A[-1], L[0], L[1]);
          else
            v = AVG3(L[-d - 2], L[-d - 1], L[-d]);
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    case VP9O_D153_PRED:
      /* :141-165 / :438-454: P(r,c) = P(r-1,c-2); seeds are cols 0,1 and row 0 */
      for (r = 0; r < bs; ++r)
        for (c = 0; c < bs; ++c) {
          const int k = r < (c >> 1) ? r : (c >> 1);
          const int rr = r - k, cc = c - 2 * k;
          int v;
          if (cc == 0)
            v = rr == 0 ? AVG2(A[-1], L[0]) : AVG2(L[rr - 1], L[rr]);
          else if (cc == 1)
            v = rr == 0   ? AVG3(L[0], A[-1], A[0])
                : rr == 1 ? AVG3(A[-1], L[0], L[1])
                          : AVG3(L[rr - 2], L[rr - 1], L[rr]);
          else
            v = AVG3(A[cc - 3], A[cc - 2], A[cc - 1]);
          out[r * bs + c] = (uint16_t)v;
        }
      break;
    default: break;
  }
}

void vp9o_highbd_intra_predictor(int mode, int bs, uint16_t *dst, ptrdiff_t stride,
                                 const uint16_t *above, const uint16_t *left, int bd) {
  uint16_t out[32 * 32];
  predict_core(mode, bs, above, left, bd, out);
  for (int r = 0; r < bs; ++r)
    for (int c = 0; c < bs; ++c) dst[r * stride + c] = out[r * bs + c];
}

void vp9o_intra_predictor(int mode, int bs, uint8_t *dst, ptrdiff_t stride, const uint8_t *above,
                          const uint8_t *left) {
  uint16_t a[1 + 64], l[32], out[32 * 32];
  for (int i = -1; i < 2 * bs; ++i) a[1 + i] = above[i];
  for (int i = 0; i < bs; ++i) l[i] = left[i];
  predict_core(mode, bs, a + 1, l, 8, out);
  for (int r = 0; r < bs; ++r)
    for (int c = 0; c < bs; ++c) dst[r * stride + c] = (uint8_t)out[r * bs + c];
}

/* mode -> which edges are assembled (vp9_reconintra.c:36-53) */
enum { NEED_L = 1, NEED_A = 2, NEED_AR = 4 };
static const unsigned char kNeeds[10] = { NEED_A | NEED_L, NEED_A, NEED_L,          NEED_AR,
                                          NEED_L | NEED_A, NEED_L | NEED_A, NEED_L | NEED_A,
                                          NEED_L,          NEED_AR,         NEED_L | NEED_A };

/* Edge assembly on a uint16 view of the frame.  get(px,py) reads the
 * reconstructed plane at absolute plane coordinates.  As shown in DESIGN.md the
 * reference's "fast" and "slow" paths agree whenever the fast one is taken, so
 * only the frame-dimension-driven form is restated. */
typedef struct {
  const uint8_t *p8;
  const uint16_t *p16;
  int stride;
} plane_view; /* pointer at the tx block's top-left pixel */

static inline int px(const plane_view *v, int dx, int dy) {
  return v->p16 ? v->p16[dy * v->stride + dx] : v->p8[dy * v->stride + dx];
}

static void build_edges(const vp9o_intra_args *a, const plane_view *ref, int bd, uint16_t *A /*[-1..2bs)*/,
                        uint16_t *L) {
  const int bs = a->bs, base = 128 << (bd - 8);
  const int needs = kNeeds[a->mode];
  const int fw = a->frame_width, fh = a->frame_height;
  int i;
  if (needs & NEED_L) {
    if (a->have_left) {
      const int valid = (a->y + bs <= fh) ? bs : fh - a->y;
      for (i = 0; i < bs; ++i) L[i] = (uint16_t)px(ref, -1, i < valid ? i : valid - 1);
    } else {
      for (i = 0; i < bs; ++i) L[i] = (uint16_t)(base + 1);
    }
  }
  if (needs & (NEED_A | NEED_AR)) {
    const int n = (needs & NEED_AR) ? 2 * bs : bs;
    if (a->have_top) {
      /* how many pixels come from the frame row above before replication */
      int take;
      if (needs & NEED_AR) {
        const int ext = (bs == 4 && a->have_right); /* above-right is only read for 4x4 */
        if (a->x + 2 * bs <= fw)
          take = ext ? 2 * bs : bs;
        else if (a->x + bs <= fw)
          take = ext ? fw - a->x : bs;
        else
          take = fw - a->x;
      } else {
        take = (a->x + bs <= fw) ? bs : fw - a->x;
      }
      for (i = 0; i < n; ++i) A[i] = (uint16_t)px(ref, i < take ? i : take - 1, -1);
      A[-1] = a->have_left ? (uint16_t)px(ref, -1, -1) : (uint16_t)(base + 1);
    } else {
      for (i = -1; i < n; ++i) A[i] = (uint16_t)(base - 1);
    }
  }
}

static void predict_block(const vp9o_intra_args *a, const plane_view *ref, int bd, uint16_t *out) {
  uint16_t abuf[16 + 64], L[32];
  uint16_t *A = abuf + 16;
  memset(abuf, 0, sizeof(abuf));
  memset(L, 0, sizeof(L));
  build_edges(a, ref, bd, A, L);
  int mode = a->mode;
  if (mode == VP9O_DC_PRED) /* dc_pred[left][up] (vp9_reconintra.c:86-89) */
    mode = a->have_left ? (a->have_top ? VP9O_DC_PRED : VP9O_DC_LEFT)
                        : (a->have_top ? VP9O_DC_TOP : VP9O_DC_128);
  predict_core(mode, a->bs, A, L, bd, out);
}

void vp9o_predict_intra(const vp9o_intra_args *a, const uint8_t *ref, int ref_stride, uint8_t *dst,
                        int dst_stride) {
  uint16_t out[32 * 32];
  plane_view v = { ref, NULL, ref_stride };
  predict_block(a, &v, 8, out);
  for (int r = 0; r < a->bs; ++r)
    for (int c = 0; c < a->bs; ++c) dst[r * dst_stride + c] = (uint8_t)out[r * a->bs + c];
}

void vp9o_highbd_predict_intra(const vp9o_intra_args *a, const uint16_t *ref, int ref_stride,
                               uint16_t *dst, int dst_stride, int bd) {
  uint16_t out[32 * 32];
  plane_view v = { NULL, ref, ref_stride };
  predict_block(a, &v, bd, out);
  for (int r = 0; r < a->bs; ++r)
    for (int c = 0; c < a->bs; ++c) dst[r * dst_stride + c] = out[r * a->bs + c];
}
