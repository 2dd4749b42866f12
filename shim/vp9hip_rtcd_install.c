/* shim/vp9hip_rtcd_install.c — the block-level integration: the reference's run-time dispatch pointers assigned to the
 * `_hip` twins (include/vp9hip_rtcd.h), exactly where its own setup_rtcd_internal assigns SIMD variants
 * (vpx-master/vpx_dsp_rtcd.h:2074-2300; the checked-in header is the Win64 one with run-time CPU detection, so the
 * names below ARE function pointers there — `RTCD_EXTERN void (*vpx_d45_predictor_16x16)(...)`, :138, and so on;
 * names it binds at compile time, `#define vpx_X vpx_X_sse2`, cannot be re-pointed and are left alone).
 *
 * vp9hip_install_rtcd() is called from initialize_dec (libvpx/vp9/decoder/vp9_decoder.c:39-49) right after
 * vpx_dsp_rtcd() and BEFORE vp9_init_intra_predictors(), which copies the predictor pointers into its mode x size
 * tables (vp9/common/vp9_reconintra.c:57-112) — oracle/patch_decodeframe.py --decoder-c --rtcd (edit E13).
 *
 * Bring-up mode: every call moves one block to the GPU and back.  shim/build/vpxdec_rtcd = the reference's vpxdec with
 * its CPU reconstruction (oracle/ref_stream_wraps.c) whose directional intra predictors and widest loop filter go
 * through these pointers; it must give the golden MD5s (tests/test_gpu_streams.py). */
#include <stdio.h>
#include <stdlib.h>

#include "./vpx_config.h"
#include "./vpx_dsp_rtcd.h"
#include "vp9hip_rtcd.h"

static int g_installed;

static void report(void) {
  if (getenv("VP9HIP_RTCD_TRACE")) fprintf(stderr, "vp9hip: %d rtcd pointers were assigned to _hip twins; last twin error: \"%s\"\n", g_installed, vp9hip_rtcd_last_error());
}

void vp9hip_install_rtcd(void) {
#define TWIN(name)          \
  do {                      \
    name = name##_hip;      \
    ++g_installed;          \
  } while (0)
  /* directional intra predictors (vpx_dsp/intrapred.c; pointers because SSSE3 variants exist) */
  TWIN(vpx_d45_predictor_16x16);
  TWIN(vpx_d45_predictor_32x32);
  TWIN(vpx_d63_predictor_4x4);
  TWIN(vpx_d63_predictor_8x8);
  TWIN(vpx_d63_predictor_16x16);
  TWIN(vpx_d63_predictor_32x32);
  TWIN(vpx_d153_predictor_4x4);
  TWIN(vpx_d153_predictor_8x8);
  TWIN(vpx_d153_predictor_16x16);
  TWIN(vpx_d153_predictor_32x32);
  TWIN(vpx_d207_predictor_8x8);
  TWIN(vpx_d207_predictor_16x16);
  TWIN(vpx_d207_predictor_32x32);
  /* the 16-wide loop filters (vpx_dsp/loopfilter.c:289-357; AVX2 variants exist) */
  TWIN(vpx_lpf_horizontal_16);
  TWIN(vpx_lpf_horizontal_16_dual);
  /* the 8-tap convolve family (vpx_dsp/vpx_convolve.c:116-240) — the fork's decoder reaches the _c functions through
   * sf->predict[][][] directly (vp9/common/vp9_scale.c:82-129), other callers (the scaler, the encoder) come here */
  TWIN(vpx_convolve8);
  TWIN(vpx_convolve8_horiz);
  TWIN(vpx_convolve8_vert);
  TWIN(vpx_convolve8_avg);
  TWIN(vpx_convolve8_avg_horiz);
  TWIN(vpx_convolve8_avg_vert);
#undef TWIN
  atexit(report);
}
