/* Two streams back to back through ONE front-end (CPU, built with -fsanitize=address,undefined by
 * tests/test_fe_concat.py): the second starts with a key frame of another size in mid-stream.
 *   - vp9hip_fe.h promises a frame's arrays for that call and the next two: the harness keeps the outputs of the two
 *     frames before the current one and reads every byte of them again AFTER each parse (what a packer thread and the
 *     device's coefficient fetch do in vp9hip_dec) — a front-end that frees or moves them when a larger frame arrives
 *     is an ASan report;
 *   - the frames of the second stream must parse to exactly the same blocks and coefficients as in a front-end that
 *     only ever saw the second stream (the previous frame's motion vectors, the segment map and the contexts are
 *     per-size state: libvpx/vp9/decoder/vp9_decodeframe.c:3507-3510).
 * fe_concat a.ivf b.ivf threads  ->  exit 0 and "fe_concat: N + M frames, second stream identical" */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vp9hip_fe.h"
#include "vp9hip_pack.h"

typedef struct {
  uint8_t *data;
  long size;
  size_t hdr;
} Ivf;

static int load(const char *path, Ivf *v) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 0, SEEK_END);
  v->size = ftell(f);
  fseek(f, 0, SEEK_SET);
  v->data = (uint8_t *)malloc((size_t)v->size);
  if (fread(v->data, 1, (size_t)v->size, f) != (size_t)v->size) return -1;
  fclose(f);
  v->hdr = (size_t)v->data[6] | ((size_t)v->data[7] << 8);
  return 0;
}

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

/* every byte of the arrays a frame hands out (for "did anything change while it was held") */
static uint64_t frame_sum(const vp9hip_fe_frame *fr) {
  uint64_t h = 1469598103934665603ull;
  h = fnv(h, &fr->params, sizeof(fr->params));
  if (fr->show_existing) return h;
  h = fnv(h, fr->blocks, sizeof(vp9hip_block) * (size_t)fr->n_blocks);
  h = fnv(h, fr->layout.block_off, sizeof(uint32_t) * 3 * (size_t)fr->n_blocks);
  const int aw = (fr->params.width + 7) & ~7, ah = (fr->params.height + 7) & ~7;
  for (int p = 0; p < 3; ++p) {
    const int w4 = (p ? aw >> fr->params.ss_x : aw) >> 2, h4 = (p ? ah >> fr->params.ss_y : ah) >> 2;
    for (int y = 0; y < h4; ++y) h = fnv(h, fr->layout.eob[p] + (size_t)y * fr->layout.eob_stride[p], sizeof(int32_t) * (size_t)w4);
  }
  for (int64_t r = 0; r < fr->layout.n_regions; ++r) {
    const vp9hip_coeff_region *g = &fr->layout.regions[r];
    const size_t esz = fr->layout.narrow ? 2 : 4;
    h = fnv(h, (const uint8_t *)fr->dqcoeff[g->plane] + esz * (size_t)g->start, esz * (size_t)g->count);
  }
  return h;
}

/* what the frame MEANS (for "same parse as in another front-end"): the eob planes only carry a value at the origin of
 * a transform block, the rest is whatever the array held before — so the eobs are taken from the packed work lists */
static vp9hip_packer *g_pk;
static uint64_t frame_meaning(const vp9hip_fe_frame *fr) {
  uint64_t h = 1469598103934665603ull;
  h = fnv(h, &fr->params, sizeof(fr->params));
  if (fr->show_existing) return h;
  h = fnv(h, fr->blocks, sizeof(vp9hip_block) * (size_t)fr->n_blocks);
  h = fnv(h, fr->layout.block_off, sizeof(uint32_t) * 3 * (size_t)fr->n_blocks);
  for (int64_t r = 0; r < fr->layout.n_regions; ++r) {
    const vp9hip_coeff_region *g = &fr->layout.regions[r];
    const size_t esz = fr->layout.narrow ? 2 : 4;
    h = fnv(h, (const uint8_t *)fr->dqcoeff[g->plane] + esz * (size_t)g->start, esz * (size_t)g->count);
  }
  vp9hip_packed out;
  if (vp9hip_pack_frame(g_pk, &fr->params, fr->blocks, fr->n_blocks, &fr->layout, &out)) return 0;
  /* (records without their first field, coeff_off: where a plane's slots are mirrored on the device depends on how
   * large the front-end's arrays have grown, not on the stream) */
  for (int i = 0; i < out.n_txb; ++i) h = fnv(h, (const uint8_t *)&out.txb[i] + 4, sizeof(vp9hip_txb) - 4);
  for (int i = 0; i < out.n_intra; ++i) h = fnv(h, (const uint8_t *)&out.intra_decode_order[i] + 4, sizeof(vp9hip_intra_task) - 4);
  h = fnv(h, out.inter, sizeof(vp9hip_inter_task) * (size_t)out.n_inter);
  return h;
}

/* parses every frame of `v`; sums[] gets one checksum per parsed frame; `held` = the two frames before the current
 * one across calls (re-read after every parse) */
static int run(vp9hip_fe *fe, const Ivf *v, uint64_t *sums, int max, vp9hip_fe_frame held[2], uint64_t held_sum[2], int *n_held) {
  int n = 0;
  for (size_t pos = v->hdr; pos + 12 <= (size_t)v->size;) {
    const size_t psz = (size_t)v->data[pos] | ((size_t)v->data[pos + 1] << 8) | ((size_t)v->data[pos + 2] << 16) | ((size_t)v->data[pos + 3] << 24);
    pos += 12;
    if (pos + psz > (size_t)v->size) break;
    uint32_t sizes[8];
    const int nf = vp9hip_fe_split_superframe(v->data + pos, psz, sizes);
    size_t off = 0;
    for (int k = 0; k < nf; ++k) {
      if (nf > 1 && sizes[k] == 0) continue;
      vp9hip_fe_frame fr;
      if (vp9hip_fe_parse(fe, v->data + pos + off, sizes[k], &fr)) {
        fprintf(stderr, "fe_concat: frame %d refused: %s\n", n, vp9hip_fe_error(fe));
        return -1;
      }
      off += sizes[k];
      /* the two frames before this one are still the caller's to read */
      for (int j = 0; j < *n_held; ++j)
        if (frame_sum(&held[j]) != held_sum[j]) {
          fprintf(stderr, "fe_concat: the arrays of an earlier frame changed while frame %d was parsed\n", n);
          return -1;
        }
      if (n < max) sums[n] = frame_meaning(&fr);
      ++n;
      if (fr.show_existing) continue;  /* (carries no arrays) */
      held[1] = held[0];
      held_sum[1] = held_sum[0];
      held[0] = fr;
      held_sum[0] = frame_sum(&fr);
      if (*n_held < 2) ++*n_held;
    }
    pos += psz;
  }
  return n;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  Ivf a, b;
  if (load(argv[1], &a) || load(argv[2], &b)) return 2;
  const int threads = atoi(argv[3]);
  enum { MAXF = 512 };
  static uint64_t sa[MAXF], sb[MAXF], sb_alone[MAXF];
  vp9hip_fe_frame held[2];
  uint64_t held_sum[2];
  int n_held = 0;
  vp9hip_fe *fe = NULL;
  if (vp9hip_packer_create(&g_pk)) return 3;
  if (vp9hip_fe_create(&fe, NULL, NULL, NULL, threads)) return 3;
  const int na = run(fe, &a, sa, MAXF, held, held_sum, &n_held);
  if (na < 0) return 1;
  const int nb = run(fe, &b, sb, MAXF, held, held_sum, &n_held);
  if (nb < 0) return 1;
  vp9hip_fe_destroy(fe);
  n_held = 0;
  if (vp9hip_fe_create(&fe, NULL, NULL, NULL, threads)) return 3;
  const int nb2 = run(fe, &b, sb_alone, MAXF, held, held_sum, &n_held);
  vp9hip_fe_destroy(fe);
  if (nb2 != nb) {
    fprintf(stderr, "fe_concat: %d frames after the first stream, %d alone\n", nb, nb2);
    return 1;
  }
  for (int i = 0; i < nb && i < MAXF; ++i)
    if (sb[i] != sb_alone[i]) {
      fprintf(stderr, "fe_concat: frame %d of the second stream parses differently after the first stream\n", i);
      return 1;
    }
  printf("fe_concat: %d + %d frames, second stream identical\n", na, nb);
  vp9hip_packer_destroy(g_pk);
  free(a.data);
  free(b.data);
  return 0;
}
