/*
 * vp9hip_dec — a VP9 decoder built only from this repository: IVF container -> vp9hip_fe (bitstream front-end,
 * CPU) -> vp9hip_decoder (reconstruction on the GPU: inter prediction, inverse transforms, intra prediction,
 * loop filter) -> frames / per-frame MD5s in vpxdec's format.  No libvpx on either side.
 *
 *   vp9hip_dec [--md5] [-o pattern] [--noblit] [--fetch] [--summary] [--loops=N] [--threads=N] [--device=N] [--serial] [--stats]
 *              [--parse-only] file.ivf
 *
 * --md5 with -o 'img-%wx%h-%4.i420' prints what `vpxdec --rawvideo --md5 -o img-%wx%h-%4.i420` prints (vpxdec.c:285-302,
 * 1036-1042): one MD5 per shown frame over its visible samples (2 bytes each above 8 bits).  Without --md5 and
 * with -o the frames are written to the files the pattern names.  --summary prints vpxdec's line
 * (vpxdec.c:358-363) with the time spent decoding (file reading and hashing excluded).
 *
 * Frames are pipelined unless --serial: a second thread owns the GPU side — it packs frame N and queues its kernels
 * behind those of frame N - 1, then fetches frame N - 1 as soon as that frame's own run is through — while the first
 * thread parses frame N + 1 (the front-end rotates three
 * sets of output arrays, the coefficient arrays in page-locked memory).  A frame that is not shown is never
 * fetched; --noblit without --md5 fetches nothing unless --fetch is given.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vp9hip_decoder.h"
#include "vp9hip_fe.h"

/* ---- MD5 (RFC 1321) ------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t s[4];
  uint64_t len;
  uint8_t buf[64];
  int fill;
} Md5;
static const uint32_t kMd5K[64] = {
  0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
  0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
  0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
  0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
  0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
  0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391
};
static const uint8_t kMd5R[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                   4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
static void md5_block(Md5 *m, const uint8_t *p) {
  uint32_t w[16], a = m->s[0], b = m->s[1], c = m->s[2], d = m->s[3];
  for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  for (int i = 0; i < 64; ++i) {
    uint32_t f;
    int g;
    if (i < 16) {
      f = (b & c) | (~b & d);
      g = i;
    } else if (i < 32) {
      f = (d & b) | (~d & c);
      g = (5 * i + 1) & 15;
    } else if (i < 48) {
      f = b ^ c ^ d;
      g = (3 * i + 5) & 15;
    } else {
      f = c ^ (b | ~d);
      g = (7 * i) & 15;
    }
    const uint32_t x = a + f + kMd5K[i] + w[g];
    a = d;
    d = c;
    c = b;
    b = b + ((x << kMd5R[i]) | (x >> (32 - kMd5R[i])));
  }
  m->s[0] += a;
  m->s[1] += b;
  m->s[2] += c;
  m->s[3] += d;
}
static void md5_init(Md5 *m) {
  m->s[0] = 0x67452301;
  m->s[1] = 0xefcdab89;
  m->s[2] = 0x98badcfe;
  m->s[3] = 0x10325476;
  m->len = 0;
  m->fill = 0;
}
static void md5_update(Md5 *m, const uint8_t *p, size_t n) {
  m->len += n;
  while (n) {
    if (m->fill == 0 && n >= 64) {
      md5_block(m, p);
      p += 64;
      n -= 64;
      continue;
    }
    size_t k = 64 - (size_t)m->fill;
    if (k > n) k = n;
    memcpy(m->buf + m->fill, p, k);
    m->fill += (int)k;
    p += k;
    n -= k;
    if (m->fill == 64) {
      md5_block(m, m->buf);
      m->fill = 0;
    }
  }
}
static void md5_final(Md5 *m, uint8_t out[16]) {
  const uint64_t bits = m->len * 8;
  const uint8_t pad = 0x80, zero = 0;
  md5_update(m, &pad, 1);
  while (m->fill != 56) md5_update(m, &zero, 1);
  uint8_t l[8];
  for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
  md5_update(m, l, 8);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(m->s[i] >> (8 * j));
}

/* ---- helpers ------------------------------------------------------------------------------------------- */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void *pinned_alloc(void *user, size_t bytes) { return vp9hip_decoder_host_alloc((vp9hip_decoder *)user, bytes); }
static void pinned_free(void *user, void *p) { vp9hip_decoder_host_free((vp9hip_decoder *)user, p); }

static void make_name(const char *pattern, char *out, size_t cap, int w, int h, int frame) { /* generate_filename, vpxdec.c:423 */
  size_t o = 0;
  for (const char *p = pattern; *p && o + 16 < cap; ++p) {
    if (*p != '%') {
      out[o++] = *p;
      continue;
    }
    ++p;
    if (*p == 'w')
      o += (size_t)snprintf(out + o, cap - o, "%d", w);
    else if (*p == 'h')
      o += (size_t)snprintf(out + o, cap - o, "%d", h);
    else if (*p >= '1' && *p <= '9')
      o += (size_t)snprintf(out + o, cap - o, "%0*d", *p - '0', frame);
    else
      out[o++] = *p;
  }
  out[o] = 0;
}

typedef struct {
  vp9hip_decoder *dec;
  uint8_t *host[3];
  size_t host_cap[3];
  int do_md5, noblit, fetch, frame_out;
  const char *pattern;
  double t_fetch, t_hash;
} Output;

/* fetch pool slot `slot` and deliver it the way vpxdec does */
/* ring_set >= 0: the frame's own run is waited for, not what was queued behind it (vp9hip_decoder_download_after) */
static int deliver(Output *o, int slot, const vp9hip_frame_params *P, int ring_set) {
  if (o->noblit && !o->do_md5 && !o->fetch) {
    ++o->frame_out;
    return 0;
  }
  const double t0 = now_s();
  const int bps = P->hbd ? 2 : 1;
  const int aw = (P->width + 7) & ~7, ah = (P->height + 7) & ~7;
  vp9hip_host_frame hf;
  memset(&hf, 0, sizeof(hf));
  for (int p = 0; p < 3; ++p) {
    const int ss = p ? P->ss_x : 0;
    const size_t need = (size_t)(aw >> ss) * (size_t)(ah >> ss) * (size_t)bps;
    if (need > o->host_cap[p]) {
      if (o->host[p]) vp9hip_decoder_host_free(o->dec, o->host[p]);
      o->host[p] = (uint8_t *)vp9hip_decoder_host_alloc(o->dec, need);
      o->host_cap[p] = o->host[p] ? need : 0;
      if (!o->host[p]) return -1;
    }
    hf.plane[p] = o->host[p];
    hf.stride[p] = aw >> ss;
  }
  hf.width = P->width;
  hf.height = P->height;
  hf.ss_x = P->ss_x;
  hf.ss_y = P->ss_y;
  hf.bit_depth = P->bit_depth;
  hf.hbd = P->hbd;
  if (ring_set >= 0 ? vp9hip_decoder_download_after(o->dec, slot, &hf, ring_set) : vp9hip_decoder_download(o->dec, slot, &hf)) {
    fprintf(stderr, "vp9hip_dec: %s\n", vp9hip_decoder_error(o->dec));
    return -1;
  }
  const double t1 = now_s();
  o->t_fetch += t1 - t0;
  ++o->frame_out;
  if (o->noblit && !o->do_md5) return 0;
  char name[512];
  make_name(o->pattern ? o->pattern : "img-%wx%h-%4.i420", name, sizeof(name), P->width, P->height, o->frame_out);
  Md5 m;
  FILE *f = NULL;
  if (o->do_md5)
    md5_init(&m);
  else if (!(f = fopen(name, "wb")))
    return -1;
  for (int p = 0; p < 3; ++p) {
    const int ss = p ? P->ss_x : 0;
    const int w = ((P->width + ss) >> ss) * bps, hh = (P->height + ss) >> ss;
    for (int y = 0; y < hh; ++y) {
      const uint8_t *row = o->host[p] + (size_t)y * (size_t)hf.stride[p] * (size_t)bps;
      if (o->do_md5)
        md5_update(&m, row, (size_t)w);
      else
        fwrite(row, 1, (size_t)w, f);
    }
  }
  if (o->do_md5) {
    uint8_t d[16];
    md5_final(&m, d);
    for (int i = 0; i < 16; ++i) printf("%02x", d[i]);
    printf("  %s\n", name);
  } else {
    fclose(f);
  }
  o->t_hash += now_s() - t1;
  return 0;
}

/* ---- the GPU side: one frame at a time, the previous one delivered just before its successor is launched ---- */
typedef struct {
  vp9hip_decoder *dec;
  Output *out;
  int pend, pend_slot, pend_set;
  vp9hip_frame_params pend_params;
  double t_begin;
  int failed;
  /* hand-over from the parsing thread (a rendezvous: the parser waits until the previous frame has been taken care of) */
  pthread_mutex_t mu;
  pthread_cond_t cv;
  vp9hip_fe_frame frame;
  int frame_last; /* the last frame of its packet: the only one of a superframe a decoder puts out */
  int has, quit;
} Gpu;

/* the frame in flight: fetched once its own run is through */
static int gpu_flush(Gpu *g, int drain) {
  int rc = 0;
  if (g->pend) {
    g->pend = 0;
    if (g->pend_slot >= 0) rc = deliver(g->out, g->pend_slot, &g->pend_params, g->pend_set);
  }
  if (drain && !rc && vp9hip_decoder_sync(g->dec)) { /* also where a kernel-side error (loop-filter time-out) surfaces */
    fprintf(stderr, "vp9hip_dec: %s\n", vp9hip_decoder_error(g->dec));
    rc = -1;
  }
  return rc;
}

/* last: the frame is the last one of its packet.  One vpx_codec_decode call decodes every frame of a superframe and
 * vpx_codec_get_frame then returns the LAST decoded frame if that one is shown (vp9/vp9_dx_iface.c decoder_decode /
 * decoder_get_frame -> vp9_get_raw_frame: ready_for_new_data + cm->show_frame) — the lower spatial layers of a
 * superframe are decoded, may carry show_frame, and are never put out. */
static int gpu_frame(Gpu *g, const vp9hip_fe_frame *fr, int serial, int last) {
  if (fr->show_existing) {
    if (gpu_flush(g, 0)) return -1;
    if (!last) return 0;
    if (vp9hip_decoder_sync(g->dec)) return -1; /* any earlier frame may be the one shown */
    return deliver(g->out, fr->show_slot, &fr->params, -1);
  }
  const double t0 = now_s();
  const vp9hip_frame_params *P = &fr->params;
  int ok = !vp9hip_decoder_alloc_slot(g->dec, fr->new_slot, P->width, P->height, P->ss_x, P->bit_depth, P->hbd, 0);
  ok = ok && !vp9hip_decoder_begin_frame_ex(g->dec, P, fr->blocks, fr->n_blocks, &fr->layout, fr->dqcoeff, VP9HIP_BEGIN_HOST_PERSISTENT);
  const int set = vp9hip_decoder_current_set(g->dec);
  const int phases = VP9HIP_PHASE_INTRA | (fr->key_frame || fr->intra_only ? 0 : VP9HIP_PHASE_INTER) | (fr->filter_level ? VP9HIP_PHASE_LF : 0);
  ok = ok && !vp9hip_decoder_run(g->dec, phases, fr->ref_slot, fr->new_slot, NULL, fr->filter_level ? &fr->lf_thresh : NULL);
  g->t_begin += now_s() - t0;
  if (!ok) {
    fprintf(stderr, "vp9hip_dec: %s\n", vp9hip_decoder_error(g->dec));
    return -1;
  }
  /* this frame's kernels are queued behind the previous frame's: now fetch the previous one (the front-end never hands
   * out the previous frame's buffer, so nothing queued writes what is being fetched) */
  if (gpu_flush(g, 0)) return -1;
  g->pend = 1;
  g->pend_slot = (fr->show_frame && last) ? fr->new_slot : -1;
  g->pend_set = set;
  g->pend_params = *P;
  return serial ? gpu_flush(g, 1) : 0;
}

static void *gpu_main(void *arg) {
  Gpu *g = (Gpu *)arg;
  pthread_mutex_lock(&g->mu);
  for (;;) {
    while (!g->has && !g->quit) pthread_cond_wait(&g->cv, &g->mu);
    if (!g->has) break;
    pthread_mutex_unlock(&g->mu);
    const int rc = g->failed ? 0 : gpu_frame(g, &g->frame, 0, g->frame_last);
    pthread_mutex_lock(&g->mu);
    if (rc) g->failed = 1;
    g->has = 0;
    pthread_cond_broadcast(&g->cv);
  }
  pthread_mutex_unlock(&g->mu);
  return NULL;
}

int main(int argc, char **argv) {
  const char *path = NULL, *pattern = NULL;
  int do_md5 = 0, noblit = 0, fetch = 0, summary = 0, loops = 1, threads = 0, serial = 0, stats = 0, parse_only = 0, device = 0, wide_slots = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--md5"))
      do_md5 = 1;
    else if (!strcmp(argv[i], "--noblit"))
      noblit = 1;
    else if (!strcmp(argv[i], "--fetch"))
      fetch = 1;
    else if (!strcmp(argv[i], "--summary"))
      summary = 1;
    else if (!strcmp(argv[i], "--serial"))
      serial = 1;
    else if (!strcmp(argv[i], "--stats"))
      stats = 1;
    else if (!strcmp(argv[i], "--parse-only"))
      parse_only = 1;
    else if (!strcmp(argv[i], "--rawvideo") || !strcmp(argv[i], "--i420"))
      ;
    else if (!strcmp(argv[i], "--wide-slots"))
      wide_slots = 1; /* keep int32 coefficient slots (the reference's width) for every frame */
    else if (!strncmp(argv[i], "--loops=", 8))
      loops = atoi(argv[i] + 8);
    else if (!strncmp(argv[i], "--threads=", 10))
      threads = atoi(argv[i] + 10);
    else if (!strncmp(argv[i], "--device=", 9))
      device = atoi(argv[i] + 9);
    else if (!strcmp(argv[i], "-o") && i + 1 < argc)
      pattern = argv[++i];
    else if (argv[i][0] != '-')
      path = argv[i];
    else {
      fprintf(stderr, "vp9hip_dec: unknown option %s\n", argv[i]);
      return 2;
    }
  }
  if (!path) {
    fprintf(stderr, "usage: vp9hip_dec [--md5] [-o pattern] [--noblit] [--fetch] [--summary] [--loops=N] [--threads=N] [--device=N] [--serial] [--stats] [--wide-slots] [--parse-only] file.ivf\n");
    return 2;
  }
  if (!pattern && !do_md5) noblit = 1;
  FILE *f = fopen(path, "rb");
  if (!f) {
    perror(path);
    return 1;
  }
  fseek(f, 0, SEEK_END);
  const long fsz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *file = (uint8_t *)malloc((size_t)fsz + 8);
  if (!file || fread(file, 1, (size_t)fsz, f) != (size_t)fsz) {
    fprintf(stderr, "vp9hip_dec: cannot read %s\n", path);
    return 1;
  }
  fclose(f);
  if (fsz < 32 || memcmp(file, "DKIF", 4) || memcmp(file + 8, "VP90", 4)) {
    fprintf(stderr, "vp9hip_dec: %s is not a VP9 IVF file\n", path);
    return 1;
  }
  const size_t hdr = (size_t)file[6] | ((size_t)file[7] << 8);

  if (parse_only) { /* the front-end alone (no GPU is touched): entropy-stage rate */
    for (int loop = 0; loop < loops; ++loop) {
      vp9hip_fe *fe = NULL;
      if (vp9hip_fe_create(&fe, NULL, NULL, NULL, threads)) return 1;
      int n = 0;
      int64_t coefs = 0, blocks = 0;
      const double t0 = now_s();
      for (size_t pos = hdr; pos + 12 <= (size_t)fsz;) {
        const size_t psz = (size_t)file[pos] | ((size_t)file[pos + 1] << 8) | ((size_t)file[pos + 2] << 16) | ((size_t)file[pos + 3] << 24);
        pos += 12;
        if (pos + psz > (size_t)fsz) break;
        uint32_t sizes[8];
        const int nf = vp9hip_fe_split_superframe(file + pos, psz, sizes);
        size_t off = 0;
        for (int k = 0; k < nf; ++k) {
          vp9hip_fe_frame fr;
          if (nf > 1 && sizes[k] == 0) continue;
          if (vp9hip_fe_parse(fe, file + pos + off, sizes[k], &fr)) {
            fprintf(stderr, "vp9hip_dec: frame %d: %s\n", n, vp9hip_fe_error(fe));
            return 1;
          }
          off += sizes[k];
          ++n;
          coefs += fr.coeff_count;
          blocks += fr.n_blocks;
        }
        pos += psz;
      }
      const double dt = now_s() - t0;
      fprintf(stderr, "parse only: %d frames in %.0f us (%.2f fps), %.3f ms per frame, %.0f blocks and %.0f coefficients per frame\n", n, dt * 1e6,
              n / dt, 1e3 * dt / n, (double)blocks / n, (double)coefs / n);
      vp9hip_fe_destroy(fe);
    }
    free(file);
    return 0;
  }
  vp9hip_decoder *dec = NULL;
  if (vp9hip_decoder_create(device, &dec)) {
    fprintf(stderr, "vp9hip_dec: no HIP device / decoder (%s)\n", dec ? vp9hip_decoder_error(dec) : "create failed");
    return 1;
  }
  vp9hip_decoder_set_timing(dec, 0);
  int rc_all = 0;
  for (int loop = 0; loop < loops && !rc_all; ++loop) {
    vp9hip_fe *fe = NULL; /* a new stream per loop */
    if (vp9hip_fe_create(&fe, pinned_alloc, pinned_free, dec, threads)) return 1;
    vp9hip_fe_set_narrow_slots(fe, !wide_slots); /* int16 coefficient slots wherever a frame's coefficients fit */
    Output out;
    memset(&out, 0, sizeof(out));
    out.dec = dec;
    out.do_md5 = do_md5;
    out.noblit = noblit;
    out.fetch = fetch;
    out.pattern = pattern;
    Gpu g;
    memset(&g, 0, sizeof(g));
    g.dec = dec;
    g.out = &out;
    pthread_mutex_init(&g.mu, NULL);
    pthread_cond_init(&g.cv, NULL);
    pthread_t th;
    int have_thread = 0;
    if (!serial && pthread_create(&th, NULL, gpu_main, &g) == 0) have_thread = 1;
    int frames_in = 0;
    double t_parse = 0, t_hand = 0;
    const double t_loop = now_s();
    size_t pos = hdr;
    while (pos + 12 <= (size_t)fsz && !rc_all) {
      const size_t psz = (size_t)file[pos] | ((size_t)file[pos + 1] << 8) | ((size_t)file[pos + 2] << 16) | ((size_t)file[pos + 3] << 24);
      pos += 12;
      if (pos + psz > (size_t)fsz) break;
      uint32_t sizes[8];
      const int nf = vp9hip_fe_split_superframe(file + pos, psz, sizes);
      size_t off = 0;
      int last_k = nf - 1;
      while (last_k > 0 && nf > 1 && sizes[last_k] == 0) --last_k;
      for (int k = 0; k < nf && !rc_all; ++k) {
        if (nf > 1 && sizes[k] == 0) continue;
        vp9hip_fe_frame fr;
        double t0 = now_s();
        const int rc = vp9hip_fe_parse(fe, file + pos + off, sizes[k], &fr);
        off += sizes[k];
        t_parse += now_s() - t0;
        if (rc) {
          fprintf(stderr, "vp9hip_dec: frame %d: %s\n", frames_in, vp9hip_fe_error(fe));
          rc_all = 1;
          break;
        }
        ++frames_in;
        t0 = now_s();
        if (have_thread) {
          pthread_mutex_lock(&g.mu);
          while (g.has) pthread_cond_wait(&g.cv, &g.mu);
          if (g.failed) rc_all = 1;
          g.frame = fr;
          g.frame_last = k == last_k;
          g.has = 1;
          pthread_cond_broadcast(&g.cv);
          pthread_mutex_unlock(&g.mu);
        } else if (gpu_frame(&g, &fr, serial, k == last_k)) {
          rc_all = 1;
        }
        t_hand += now_s() - t0;
      }
      pos += psz;
    }
    if (have_thread) {
      pthread_mutex_lock(&g.mu);
      while (g.has) pthread_cond_wait(&g.cv, &g.mu);
      g.quit = 1;
      pthread_cond_broadcast(&g.cv);
      pthread_mutex_unlock(&g.mu);
      pthread_join(th, NULL);
      if (g.failed) rc_all = 1;
    }
    if (!rc_all && gpu_flush(&g, 1)) rc_all = 1;
    const double dt = now_s() - t_loop - out.t_hash;
    if (summary)
      fprintf(stderr, "%d decoded frames/%d showed frames in %.0f us (%.2f fps)\n", frames_in, out.frame_out, dt * 1e6,
              out.frame_out / (dt > 0 ? dt : 1));
    if (stats && frames_in)
      fprintf(stderr,
              "vp9hip_dec: per frame: parse %.3f ms, hand-over wait %.3f ms | GPU thread: pack + launch %.3f ms, wait for the frame's run + "
              "fetch %.3f ms, hash/write %.3f ms\n",
              1e3 * t_parse / frames_in, 1e3 * t_hand / frames_in, 1e3 * g.t_begin / frames_in, 1e3 * out.t_fetch / frames_in,
              1e3 * out.t_hash / frames_in);
    if (stats && frames_in)
      fprintf(stderr, "vp9hip_dec: coefficient slots: %s; %d of %d frames parsed again with int32 slots (a coefficient outside int16)\n",
              wide_slots ? "int32" : "int16 where a frame's coefficients fit", vp9hip_fe_wide_frames(fe), frames_in);
    for (int p = 0; p < 3; ++p)
      if (out.host[p]) vp9hip_decoder_host_free(dec, out.host[p]);
    vp9hip_decoder_sync(dec);
    vp9hip_fe_destroy(fe);
    pthread_mutex_destroy(&g.mu);
    pthread_cond_destroy(&g.cv);
  }
  vp9hip_decoder_destroy(dec);
  free(file);
  return rc_all;
}
