/*
 * ref_absent_simd.c — TEST INFRASTRUCTURE (oracle build only; used by the ENCODER that synthesizes
 * test streams, never by a decoder).
 *
 * The reference's checked-in Win64 dispatch header (vpx-master/vpx_dsp_rtcd.h:2194 etc.,
 * vp9_rtcd.h) points the full high-bitdepth inverse transforms at SSE2 / SSE4.1 functions whose
 * sources are not in the snapshot.  Every other absent SIMD name is mapped onto the reference's own
 * same-prototype *_c function by oracle/gen_simd_map.py; these eleven cannot be, because the fork
 * retyped their *_c twins to STORE an int64 residual instead of adding to uint16 pixels
 * (libvpx/vpx_dsp/inv_txfm.c:1450-1471, 1638-1659, 2075, 2598; vp9/common/vp9_idct.c:234, 266, 300).
 * Here each one is the composition the fork itself uses for those functions: its residual-storing
 * *_c function followed by highbd_clip_pixel_add (block_sum, vp9/decoder/vp9_decodeframe.c:290-341).
 */
#include "./vpx_config.h"
#include "./vp9_rtcd.h"
#include "./vpx_dsp_rtcd.h"
#include "vpx_dsp/inv_txfm.h"

#define ADD_CLIP(n)                                                                                \
  for (int y = 0; y < n; ++y)                                                                      \
    for (int x = 0; x < n; ++x) dest[y * stride + x] = highbd_clip_pixel_add(dest[y * stride + x], res[y * n + x], bd)

#define FULL_IDCT(name, cfn, n)                                                                    \
  void name(const tran_low_t *input, uint16_t *dest, int stride, int bd) {                        \
    tran_high_t res[n * n];                                                                        \
    cfn(input, res, n, bd);                                                                        \
    ADD_CLIP(n);                                                                                   \
  }
#define FULL_IHT(name, cfn, n)                                                                     \
  void name(const tran_low_t *input, uint16_t *dest, int stride, int tx_type, int bd) {           \
    tran_high_t res[n * n];                                                                        \
    cfn(input, res, n, tx_type, bd);                                                               \
    ADD_CLIP(n);                                                                                   \
  }

FULL_IDCT(vpx_highbd_idct4x4_16_add_sse2, vpx_highbd_idct4x4_16_add_c, 4)
FULL_IDCT(vpx_highbd_idct4x4_16_add_sse4_1, vpx_highbd_idct4x4_16_add_c, 4)
FULL_IDCT(vpx_highbd_idct8x8_64_add_sse2, vpx_highbd_idct8x8_64_add_c, 8)
FULL_IDCT(vpx_highbd_idct8x8_64_add_sse4_1, vpx_highbd_idct8x8_64_add_c, 8)
FULL_IDCT(vpx_highbd_idct16x16_256_add_sse2, vpx_highbd_idct16x16_256_add_c, 16)
FULL_IDCT(vpx_highbd_idct16x16_256_add_sse4_1, vpx_highbd_idct16x16_256_add_c, 16)
FULL_IDCT(vpx_highbd_idct32x32_1024_add_sse2, vpx_highbd_idct32x32_1024_add_c, 32)
FULL_IDCT(vpx_highbd_idct32x32_1024_add_sse4_1, vpx_highbd_idct32x32_1024_add_c, 32)
FULL_IHT(vp9_highbd_iht4x4_16_add_sse4_1, vp9_highbd_iht4x4_16_add_c, 4)
FULL_IHT(vp9_highbd_iht8x8_64_add_sse4_1, vp9_highbd_iht8x8_64_add_c, 8)
FULL_IHT(vp9_highbd_iht16x16_256_add_sse4_1, vp9_highbd_iht16x16_256_add_c, 16)
