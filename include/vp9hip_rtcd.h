/*
 * vp9hip_rtcd.h — block-level `_hip` twins of the reference's run-time-dispatch prototypes.
 *
 * Each function has the signature of the libvpx function it replaces (the `_c` function of the
 * same stem, and the function pointer / macro of that stem in the reference's
 * vpx-master/vpx_dsp_rtcd.h and vpx-master/vp9_rtcd.h; prototypes come from
 * libvpx/vpx_dsp/vpx_dsp_rtcd_defs.pl: intra :37-366, convolve :368-437, loop filter :439-515,
 * inverse transforms :605-705, and libvpx/vp9/common/vp9_rtcd_defs.pl:61-108).  A maintainer
 * assigns them in setup_rtcd_internal (vpx-master/vpx_dsp_rtcd.h:2074) — see INTEGRATION.md.
 *
 * They take HOST pointers, move one block to the GPU, run the same HIP kernels as the batched
 * entry points (include/vp9hip.h) and move the result back: meant for parity tests and
 * bring-up, not for speed.  They use a process-wide default context on device
 * $VP9HIP_DEVICE (default 0) and are not re-entrant.  They return void like their models;
 * failures (no device, unknown kernel table) are reported through vp9hip_rtcd_last_error() and
 * leave the destination untouched.
 *
 * tran_low_t is int32_t (CONFIG_VP9_HIGHBITDEPTH=1 build, vpx_dsp/vpx_dsp_common.h:36-37).
 * The highbd transform twins follow STOCK libvpx (uint16_t *dest, add + clip), not the fork's
 * residual-store edit (vpx_dsp/inv_txfm.c:1450-1471).
 */
#ifndef VP9HIP_RTCD_H_
#define VP9HIP_RTCD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t vp9hip_tran_low_t;
typedef int16_t vp9hip_interp_kernel[8]; /* InterpKernel, vpx_dsp/vpx_filter.h */

/* "" when the last twin call succeeded */
const char *vp9hip_rtcd_last_error(void);

/* ---- inverse transforms: vpx_dsp/inv_txfm.c:18-1276, vp9/common/vp9_idct.c:20-116 ---- */
void vpx_idct4x4_1_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct4x4_1_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct4x4_16_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct4x4_16_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct8x8_1_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct8x8_1_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct8x8_12_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct8x8_12_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct8x8_64_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct8x8_64_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct16x16_1_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct16x16_1_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct16x16_10_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct16x16_10_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct16x16_38_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct16x16_38_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct16x16_256_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct16x16_256_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct32x32_1_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct32x32_1_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct32x32_34_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct32x32_34_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct32x32_135_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct32x32_135_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_idct32x32_1024_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_idct32x32_1024_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_iwht4x4_1_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_iwht4x4_1_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vpx_iwht4x4_16_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride);
void vpx_highbd_iwht4x4_16_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int bd);
void vp9_iht4x4_16_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride, int tx_type);
void vp9_highbd_iht4x4_16_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int tx_type, int bd);
void vp9_iht8x8_64_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride, int tx_type);
void vp9_highbd_iht8x8_64_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int tx_type, int bd);
void vp9_iht16x16_256_add_hip(const vp9hip_tran_low_t *input, uint8_t *dest, int stride, int tx_type);
void vp9_highbd_iht16x16_256_add_hip(const vp9hip_tran_low_t *input, uint16_t *dest, int stride, int tx_type, int bd);

/* ---- convolve: convolve_fn_t / highbd_convolve_fn_t, vpx_dsp/vpx_convolve.h:22-35 ---- */
void vpx_convolve_copy_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve_copy_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve_avg_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve_avg_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_horiz_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_horiz_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_vert_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_vert_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_avg_horiz_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_avg_horiz_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_avg_vert_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_avg_vert_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_convolve8_avg_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_highbd_convolve8_avg_hip(const uint16_t *src, ptrdiff_t src_stride, uint16_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h, int bd);
void vpx_scaled_horiz_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_scaled_vert_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_scaled_2d_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_scaled_avg_horiz_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_scaled_avg_vert_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);
void vpx_scaled_avg_2d_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const vp9hip_interp_kernel *filter, int x0_q4, int x_step_q4, int y0_q4, int y_step_q4, int w, int h);

/* ---- intra predictors: vpx_dsp/intrapred.c:21-915 ---- */
void vpx_dc_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_left_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_left_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_left_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_left_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_left_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_left_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_left_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_left_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_top_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_top_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_top_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_top_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_top_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_top_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_top_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_top_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_128_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_128_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_128_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_128_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_128_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_128_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_dc_128_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_dc_128_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_v_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_v_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_v_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_v_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_v_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_v_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_v_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_v_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_h_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_h_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_h_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_h_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_h_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_h_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_h_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_h_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d45_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d45_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d45_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d45_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d45_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d45_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d45_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d45_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d135_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d135_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d135_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d135_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d135_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d135_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d135_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d135_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d117_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d117_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d117_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d117_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d117_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d117_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d117_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d117_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d153_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d153_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d153_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d153_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d153_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d153_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d153_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d153_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d207_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d207_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d207_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d207_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d207_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d207_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d207_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d207_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d63_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d63_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d63_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d63_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d63_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d63_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_d63_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_d63_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
/* 8-bit only, 4x4 only (libvpx/vpx_dsp/vpx_dsp_rtcd_defs.pl:46, 51, 57, 70; VP9 never selects them) */
void vpx_d45e_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_d63e_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_he_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_ve_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_tm_predictor_4x4_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_tm_predictor_4x4_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_tm_predictor_8x8_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_tm_predictor_8x8_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_tm_predictor_16x16_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_tm_predictor_16x16_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);
void vpx_tm_predictor_32x32_hip(uint8_t *dst, ptrdiff_t stride, const uint8_t *above, const uint8_t *left);
void vpx_highbd_tm_predictor_32x32_hip(uint16_t *dst, ptrdiff_t stride, const uint16_t *above, const uint16_t *left, int bd);

/* ---- loop filter: vpx_dsp/loopfilter.c:112-357, highbd :450-743 ---- */
void vpx_lpf_horizontal_4_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_horizontal_4_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_horizontal_4_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1);
void vpx_highbd_lpf_horizontal_4_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd);
void vpx_lpf_horizontal_8_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_horizontal_8_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_horizontal_8_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1);
void vpx_highbd_lpf_horizontal_8_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd);
void vpx_lpf_horizontal_16_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_horizontal_16_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_horizontal_16_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_horizontal_16_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_vertical_4_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_vertical_4_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_vertical_4_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1);
void vpx_highbd_lpf_vertical_4_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd);
void vpx_lpf_vertical_8_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_vertical_8_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_vertical_8_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1);
void vpx_highbd_lpf_vertical_8_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd);
void vpx_lpf_vertical_16_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_vertical_16_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);
void vpx_lpf_vertical_16_dual_hip(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
void vpx_highbd_lpf_vertical_16_dual_hip(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd);

#ifdef __cplusplus
}
#endif
#endif /* VP9HIP_RTCD_H_ */
