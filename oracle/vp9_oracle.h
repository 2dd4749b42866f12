/*
 * vp9_oracle.h — CPU restatement of the VP9 block-reconstruction arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load liboracle.so, and there only as the checker.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/libvpx/) whose behaviour it restates.  The restatement is
 * pinned against (a) the reference's own object code built from its own
 * sources into oracle/_ref/libvpxref.so (see oracle/Makefile) and (b) the
 * known-answer MD5s in test/test_intra_pred_speed.cc.
 *
 * Integer overflow convention: the reference relies on values staying in
 * range for valid streams; where C leaves overflow undefined we (and the
 * _ref build, via -fwrapv) define it as two's-complement wrap.
 */
#ifndef VP9_ORACLE_H_
#define VP9_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- inverse transforms (SURVEY §8 a1–a3) -------------------------------- */

/* tx_type as in vp9/common/vp9_enums.h: 0 DCT_DCT, 1 ADST_DCT (adst on
 * columns), 2 DCT_ADST (adst on rows), 3 ADST_ADST. */
enum { VP9O_DCT_DCT = 0, VP9O_ADST_DCT = 1, VP9O_DCT_ADST = 2, VP9O_ADST_ADST = 3 };

/* 1-D transforms; n in {4,8,16,32} (adst: 4,8,16).  hbd==0 follows the 8-bit
 * functions (idct4_c.. / iadst4_c..), hbd!=0 the vpx_highbd_* ones. */
void vp9o_idct1d(int n, const int32_t *in, int32_t *out, int hbd);
void vp9o_iadst1d(int n, const int32_t *in, int32_t *out, int hbd);

/* Full 2-D inverse transform, residual only (post final rounding shift,
 * before the add): res[n*n] row-major.  lossless -> WHT (n must be 4). */
void vp9o_inv_txfm_residual(int n, int tx_type, int lossless, int hbd,
                            const int32_t *coeffs, int32_t *res);

/* eob-dispatching add, the semantics of vp9_idct{4x4,8x8,16x16,32x32}_add /
 * vp9_iht*_add / vp9_iwht4x4_add (vp9/common/vp9_idct.c:119-204) and their
 * highbd twins (:308-396).  bd==8 && !hbd -> uint8 dest; hbd -> uint16 dest. */
void vp9o_inv_txfm_add(int n, int tx_type, int lossless, const int32_t *coeffs,
                       uint8_t *dest, int stride, int eob);
void vp9o_highbd_inv_txfm_add(int n, int tx_type, int lossless,
                              const int32_t *coeffs, uint16_t *dest, int stride,
                              int eob, int bd);

/* ---- 8-tap convolve family -------------------------------- */

/* filter bank index as in vp9/common/vp9_filter.h:23-28 */
enum { VP9O_EIGHTTAP = 0, VP9O_EIGHTTAP_SMOOTH = 1, VP9O_EIGHTTAP_SHARP = 2,
       VP9O_BILINEAR = 3, VP9O_FOURTAP = 4 };
/* returns the [16][8] int16 kernel bank */
const int16_t (*vp9o_filter_kernels(int filter))[8];

/* Mirrors convolve_fn_t (vpx_dsp/vpx_convolve.h:22-27).
 * mode bit0: horizontal filter active, bit1: vertical filter active,
 * bit2: average into dst.  mode 0 = copy, 4 = avg.  "scaled" selects the
 * vpx_scaled_* variants (64x(64*2+7) temp, arbitrary step). */
void vp9o_convolve(int mode, int scaled, const uint8_t *src, ptrdiff_t src_stride,
                   uint8_t *dst, ptrdiff_t dst_stride, const int16_t (*kernel)[8],
                   int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vp9o_highbd_convolve(int mode, int scaled, const uint16_t *src,
                          ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride,
                          const int16_t (*kernel)[8], int x0_q4, int x_step_q4,
                          int y0_q4, int y_step_q4, int w, int h, int bd);

/* Block-level inter predictor with the decoder's border emulation:
 * dec_build_inter_predictors, vp9/decoder/vp9_decodeframe.c:563-690.  ref points at plane
 * sample (0,0); fw/fh = crop size of the reference plane. */
void vp9o_inter_predict_block(const uint8_t *ref, int ref_stride, int fw, int fh, int px_q4,
                              int py_q4, int xs, int ys, int filter, int w, int h, uint8_t *dst,
                              int dst_stride, int avg);
void vp9o_highbd_inter_predict_block(const uint16_t *ref, int ref_stride, int fw, int fh,
                                     int px_q4, int py_q4, int xs, int ys, int filter, int w,
                                     int h, uint16_t *dst, int dst_stride, int avg, int bd);

/* ---- intra predictors --------------------------------- */

/* mode numbering as PREDICTION_MODE (vp9/common/vp9_blockd.h): 0 DC, 1 V, 2 H,
 * 3 D45, 4 D135, 5 D117, 6 D153, 7 D207, 8 D63, 9 TM.  Extra ids for the dc
 * variants: 10 DC_128, 11 DC_LEFT, 12 DC_TOP. */
enum { VP9O_DC_PRED = 0, VP9O_V_PRED, VP9O_H_PRED, VP9O_D45_PRED, VP9O_D135_PRED,
       VP9O_D117_PRED, VP9O_D153_PRED, VP9O_D207_PRED, VP9O_D63_PRED, VP9O_TM_PRED,
       VP9O_DC_128, VP9O_DC_LEFT, VP9O_DC_TOP };

/* Raw predictor: the vpx_<mode>_predictor_NxN_c family (vpx_dsp/intrapred.c).
 * above must be readable on [-1, 2*bs) and left on [0, bs). */
void vp9o_intra_predictor(int mode, int bs, uint8_t *dst, ptrdiff_t stride,
                          const uint8_t *above, const uint8_t *left);
void vp9o_highbd_intra_predictor(int mode, int bs, uint16_t *dst, ptrdiff_t stride,
                                 const uint16_t *above, const uint16_t *left, int bd);

/* Edge builder + dispatch: build_intra_predictors (vp9_reconintra.c:262-402),
 * high variant (:113-259).  ref/dst point at the block's top-left pixel in the
 * frame being reconstructed.  frame_width/height are the plane's ALIGNED
 * dimensions (y_width / uv_width), x,y the block position in plane pixels. */
typedef struct {
  int mode;        /* 0..9 */
  int bs;          /* 4,8,16,32 */
  int have_top, have_left, have_right;
  int x, y;        /* position of the tx block inside the plane (pixels) */
  int frame_width, frame_height; /* aligned plane dims (0 => no edge handling) */
} vp9o_intra_args;
void vp9o_predict_intra(const vp9o_intra_args *a, const uint8_t *ref, int ref_stride,
                        uint8_t *dst, int dst_stride);
void vp9o_highbd_predict_intra(const vp9o_intra_args *a, const uint16_t *ref,
                               int ref_stride, uint16_t *dst, int dst_stride, int bd);

/* ---- loop filter ------------------------------------ */

/* kind: 4, 8, 16; dual: 0/1; vertical: 0 = horizontal edge fn, 1 = vertical.
 * Mirrors vpx_lpf_{horizontal,vertical}_{4,8,16}{,_dual}_c
 * (vpx_dsp/loopfilter.c:112-357).  For non-dual calls only b0/l0/t0 are used. */
void vp9o_lpf(int vertical, int kind, int dual, uint8_t *s, int pitch,
              const uint8_t *b0, const uint8_t *l0, const uint8_t *t0,
              const uint8_t *b1, const uint8_t *l1, const uint8_t *t1);
void vp9o_highbd_lpf(int vertical, int kind, int dual, uint16_t *s, int pitch,
                     const uint8_t *b0, const uint8_t *l0, const uint8_t *t0,
                     const uint8_t *b1, const uint8_t *l1, const uint8_t *t1, int bd);

/* Loop-filter driver for a whole frame: loop_filter_rows +
 * vp9_filter_block_plane_ss00/ss11 + filter_selectively_* (vp9_loopfilter.c:297-650,
 * 1241-1469).  lfm has the layout of LOOP_FILTER_MASK (vp9_loopfilter.h:60-68). */
typedef struct {
  uint64_t left_y[4], above_y[4], int_4x4_y;
  uint16_t left_uv[4], above_uv[4], int_4x4_uv;
  uint8_t lfl_y[64];
  uint8_t reserved[6];
} vp9o_lfm;
typedef struct {
  uint8_t mblim[64], lim[64], hev_thr[64];
} vp9o_lf_thresh;
void vp9o_loop_filter_frame(const vp9o_lfm *lfm, int sb_rows, int sb_cols, const vp9o_lf_thresh *th,
                            void *const planes[3], const int strides[3], int mi_rows, int bd, int hbd,
                            int nplanes);

/* ---- whole-frame sequential reconstruction from packed work lists ---------------------- */
typedef struct {
  void *plane[3];
  int32_t stride[3];
  int32_t width[3], height[3];
  int32_t awidth[3], aheight[3];
  int32_t bit_depth, hbd;
} vp9o_frame; /* same layout as vp9hip_frame, host pointers */
void vp9o_recon_inter_list(const void *tasks, int n, const vp9o_frame *refs, const vp9o_frame *dst);
void vp9o_recon_txb_list(const void *blocks, int n, const int32_t *coeffs, const vp9o_frame *f);
void vp9o_recon_intra_list(const void *tasks /* decode order */, int n, const int32_t *coeffs,
                           const vp9o_frame *f);

#ifdef __cplusplus
}
#endif
#endif /* VP9_ORACLE_H_ */
