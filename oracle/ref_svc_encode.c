/* oracle/ref_svc_encode.c — TEST INFRASTRUCTURE (never part of the product).
 *
 * Drives the reference's own VP9 encoder (its libvpx, compiled where it lies by oracle/build_refvpx.sh) through the
 * public encoder API to make streams its command-line encoder cannot:
 *   - spatial layers: every superframe carries a half-size frame and a full-size frame predicted from it, so the
 *     decoder takes a frame's size from / against references of ANOTHER size (setup_frame_size_with_refs,
 *     libvpx/vp9/decoder/vp9_decodeframe.c:1781) and predicts through scale factors (:3232-3237,
 *     vp9_setup_scale_factors_for_frame; the scaled convolve of vpx_dsp/vpx_convolve.c:242-290);
 *   - an INTRA-ONLY frame in mid-stream (:3182-3213): VP9E_SET_SVC_SPATIAL_LAYER_SYNC with base_layer_intra_only
 *     (vp9/vp9_cx_iface.c:1652-1662 -> set_intra_only_frame, vp9/encoder/vp9_ratectrl.c:2195): hidden, refreshes
 *     LAST / GOLDEN / ALTREF, resets nothing else.
 * Like `vpxenc --test-decode=fatal`, every superframe is decoded right away by the CPU stream oracle linked into this
 * binary (the patched frame driver + oracle/ref_stream_wraps.c) and its reference buffer 0 compared with the
 * encoder's own reconstruction (VP9_GET_REFERENCE on both sides — what the reference's example encoder does,
 * examples/vp9_spatial_svc_encoder.c:703-760): a stream only gets out if the oracle decodes it exactly.
 *
 *   ref_svc_encode in.yuv width height frames out.ivf layers intra_only_at [kbps] [speed]
 *     in.yuv: 8-bit I420; layers: 1..3 spatial layers (2:1 steps); intra_only_at: frame index whose base layer is
 *     coded intra-only (-1: none) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vpx/vp8cx.h"
#include "vpx/vp8dx.h"
#include "vpx/vpx_decoder.h"
#include "vpx/vpx_encoder.h"

static void put32(uint8_t *p, uint32_t v) {
  p[0] = (uint8_t)v, p[1] = (uint8_t)(v >> 8), p[2] = (uint8_t)(v >> 16), p[3] = (uint8_t)(v >> 24);
}

static void die(const char *what, vpx_codec_ctx_t *c) {
  fprintf(stderr, "ref_svc_encode: %s: %s (%s)\n", what, c ? vpx_codec_error(c) : "", c && vpx_codec_error_detail(c) ? vpx_codec_error_detail(c) : "");
  exit(1);
}

static int same_image(const vpx_image_t *a, const vpx_image_t *b) {
  if (a->d_w != b->d_w || a->d_h != b->d_h || a->fmt != b->fmt) return 0;
  for (int p = 0; p < 3; ++p) {
    const unsigned w = p ? (a->d_w + a->x_chroma_shift) >> a->x_chroma_shift : a->d_w;
    const unsigned h = p ? (a->d_h + a->y_chroma_shift) >> a->y_chroma_shift : a->d_h;
    const unsigned bps = (a->fmt & VPX_IMG_FMT_HIGHBITDEPTH) ? 2 : 1;
    for (unsigned y = 0; y < h; ++y)
      if (memcmp(a->planes[p] + (size_t)y * a->stride[p], b->planes[p] + (size_t)y * b->stride[p], (size_t)w * bps)) return 0;
  }
  return 1;
}

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: ref_svc_encode in.yuv width height frames out.ivf layers intra_only_at [kbps] [speed] [bit depth 8|10|12]\n");
    return 2;
  }
  const int w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]), layers = atoi(argv[6]), intra_at = atoi(argv[7]);
  const int kbps = argc > 8 ? atoi(argv[8]) : 1200, speed = argc > 9 ? atoi(argv[9]) : 6;
  const int depth = argc > 10 ? atoi(argv[10]) : 8; /* > 8: profile 2, 16-bit little-endian source samples */
  if (w <= 0 || h <= 0 || frames <= 0 || layers < 1 || layers > 3 || (depth != 8 && depth != 10 && depth != 12)) return 2;
  FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[5], "wb");
  if (!in || !out) return 2;

  vpx_codec_enc_cfg_t cfg;
  if (vpx_codec_enc_config_default(vpx_codec_vp9_cx(), &cfg, 0)) die("config", NULL);
  cfg.g_w = w;
  cfg.g_h = h;
  if (depth > 8) {
    cfg.g_profile = 2;
    cfg.g_bit_depth = depth == 10 ? VPX_BITS_10 : VPX_BITS_12;
    cfg.g_input_bit_depth = (unsigned)depth;
  }
  cfg.g_timebase.num = 1;
  cfg.g_timebase.den = 30;
  cfg.g_pass = VPX_RC_ONE_PASS;
  cfg.g_lag_in_frames = 0;
  cfg.g_threads = 1;
  cfg.g_error_resilient = 1; /* one frame context for all layers */
  cfg.rc_end_usage = VPX_CBR;
  cfg.rc_target_bitrate = kbps;
  cfg.rc_resize_allowed = 0;
  cfg.rc_min_quantizer = 2;
  cfg.rc_max_quantizer = 56;
  cfg.rc_undershoot_pct = 50;
  cfg.rc_overshoot_pct = 50;
  cfg.rc_buf_initial_sz = 500;
  cfg.rc_buf_optimal_sz = 600;
  cfg.rc_buf_sz = 1000;
  cfg.rc_dropframe_thresh = 0;
  cfg.kf_mode = VPX_KF_AUTO;
  cfg.kf_min_dist = cfg.kf_max_dist = 9999;
  cfg.ss_number_layers = layers;
  cfg.ts_number_layers = 1;
  cfg.temporal_layering_mode = VP9E_TEMPORAL_LAYERING_MODE_NOLAYERING;
  cfg.ts_rate_decimator[0] = 1;
  {
    /* the larger layers get the larger share */
    unsigned share[3] = { 1, 3, 8 }, sum = 0;
    for (int sl = 0; sl < layers; ++sl) sum += share[sl];
    for (int sl = 0; sl < layers; ++sl) {
      cfg.ss_target_bitrate[sl] = kbps * share[sl] / sum;
      cfg.layer_target_bitrate[sl] = cfg.ss_target_bitrate[sl];
    }
    cfg.ts_target_bitrate[0] = kbps;
  }
  vpx_codec_ctx_t enc, dec;
  if (vpx_codec_enc_init(&enc, vpx_codec_vp9_cx(), &cfg, depth > 8 ? VPX_CODEC_USE_HIGHBITDEPTH : 0)) die("encoder init", &enc);
  if (vpx_codec_dec_init(&dec, vpx_codec_vp9_dx(), NULL, 0)) die("decoder init", &dec);
  vpx_svc_extra_cfg_t sp;
  memset(&sp, 0, sizeof(sp));
  for (int sl = 0; sl < layers; ++sl) {
    sp.scaling_factor_num[sl] = 1;
    sp.scaling_factor_den[sl] = 1 << (layers - 1 - sl);
    sp.max_quantizers[sl] = cfg.rc_max_quantizer;
    sp.min_quantizers[sl] = cfg.rc_min_quantizer;
    sp.speed_per_layer[sl] = speed;
  }
  if (layers > 1) {
    if (vpx_codec_control(&enc, VP9E_SET_SVC, 1)) die("VP9E_SET_SVC", &enc);
    if (vpx_codec_control(&enc, VP9E_SET_SVC_PARAMETERS, &sp)) die("VP9E_SET_SVC_PARAMETERS", &enc);
    if (vpx_codec_control(&enc, VP9E_SET_SVC_INTER_LAYER_PRED, 0)) die("VP9E_SET_SVC_INTER_LAYER_PRED", &enc);
  }
  vpx_codec_control(&enc, VP8E_SET_CPUUSED, speed);
  vpx_codec_control(&enc, VP9E_SET_TILE_COLUMNS, w >= 512 ? 1 : 0);
  vpx_codec_control(&enc, VP9E_SET_AQ_MODE, 0);
  vpx_codec_control(&enc, VP9E_SET_NOISE_SENSITIVITY, 0);

  uint8_t hdr[32];
  memset(hdr, 0, sizeof(hdr));
  memcpy(hdr, "DKIF", 4);
  hdr[6] = 32;
  memcpy(hdr + 8, "VP90", 4);
  hdr[12] = (uint8_t)w, hdr[13] = (uint8_t)(w >> 8), hdr[14] = (uint8_t)h, hdr[15] = (uint8_t)(h >> 8);
  put32(hdr + 16, 30);
  put32(hdr + 20, 1);
  fwrite(hdr, 1, 32, out);

  vpx_image_t raw;
  if (!vpx_img_alloc(&raw, depth > 8 ? VPX_IMG_FMT_I42016 : VPX_IMG_FMT_I420, w, h, 32)) return 3;
  raw.bit_depth = (unsigned)depth;
  const size_t bps = depth > 8 ? 2 : 1;
  int written = 0, intra_only_seen = 0;
  for (int i = 0; i <= frames; ++i) { /* one more call to flush */
    vpx_image_t *img = NULL;
    if (i < frames) {
      for (int p = 0; p < 3; ++p) {
        const int pw = p ? (w + 1) / 2 : w, ph = p ? (h + 1) / 2 : h;
        for (int y = 0; y < ph; ++y)
          if (fread(raw.planes[p] + (size_t)y * raw.stride[p], bps, (size_t)pw, in) != (size_t)pw) die("short read of the source", NULL);
      }
      img = &raw;
      if (i == intra_at && layers > 1) {
        vpx_svc_spatial_layer_sync_t sync;
        memset(&sync, 0, sizeof(sync));
        for (int sl = 1; sl < layers; ++sl) sync.spatial_layer_sync[sl] = 1;
        sync.base_layer_intra_only = 1;
        if (vpx_codec_control(&enc, VP9E_SET_SVC_SPATIAL_LAYER_SYNC, &sync)) die("VP9E_SET_SVC_SPATIAL_LAYER_SYNC", &enc);
      }
    }
    if (vpx_codec_encode(&enc, img, i, 1, 0, VPX_DL_REALTIME)) die("encode", &enc);
    vpx_codec_iter_t it = NULL;
    const vpx_codec_cx_pkt_t *pkt;
    while ((pkt = vpx_codec_get_cx_data(&enc, &it)) != NULL) {
      if (pkt->kind != VPX_CODEC_CX_FRAME_PKT) continue;
      uint8_t fh[12];
      put32(fh, (uint32_t)pkt->data.frame.sz);
      put32(fh + 4, (uint32_t)written);
      put32(fh + 8, 0);
      fwrite(fh, 1, 12, out);
      fwrite(pkt->data.frame.buf, 1, pkt->data.frame.sz, out);
      ++written;
      /* the oracle decodes what the encoder just wrote; both sides' reference buffer 0 must hold the same frame */
      if (vpx_codec_decode(&dec, (const uint8_t *)pkt->data.frame.buf, (unsigned)pkt->data.frame.sz, NULL, 0)) die("oracle decode", &dec);
      vpx_codec_iter_t di = NULL;
      while (vpx_codec_get_frame(&dec, &di) != NULL) {
      }
      for (int idx = 0; idx < 3; ++idx) {
        struct vp9_ref_frame re, rd;
        memset(&re, 0, sizeof(re));
        memset(&rd, 0, sizeof(rd));
        re.idx = rd.idx = idx;
        if (vpx_codec_control(&enc, VP9_GET_REFERENCE, &re) || vpx_codec_control(&dec, VP9_GET_REFERENCE, &rd)) continue;
        if (!same_image(&re.img, &rd.img)) {
          fprintf(stderr, "ref_svc_encode: Encode/decode mismatch in reference buffer %d after superframe %d\n", idx, written - 1);
          return 1;
        }
      }
      /* (an intra-only frame is the only non-key frame whose first byte pattern is checked by the generator script;
       * here only counted through the encoder's flag) */
      if (i == intra_at && !(pkt->data.frame.flags & VPX_FRAME_IS_KEY)) intra_only_seen = 1;
    }
  }
  /* IVF frame count */
  fseek(out, 24, SEEK_SET);
  uint8_t cnt[4];
  put32(cnt, (uint32_t)written);
  fwrite(cnt, 1, 4, out);
  fclose(out);
  fclose(in);
  vpx_img_free(&raw);
  vpx_codec_destroy(&enc);
  vpx_codec_destroy(&dec);
  printf("ref_svc_encode: %d superframes, %d spatial layer(s)%s, every one decoded identically by the stream oracle\n", written, layers,
         intra_only_seen ? ", intra-only base layer requested" : "");
  return 0;
}
