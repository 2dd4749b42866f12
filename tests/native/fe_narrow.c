/* int16 coefficient slots (vp9hip_fe_set_narrow_slots) against int32 ones, CPU, built with -fsanitize=address,undefined
 * by tests/test_fe_narrow.py: every frame of a stream through two front-ends; blocks, slot offsets and regions must be
 * identical, every int16 slot must be the int32 slot's value, and — with a test limit that makes ordinary frames "not
 * fit" — the frames that fall back must come out with int32 slots equal to the plain ones (the fall-back parses the
 * tile columns a second time: same lists, same stream state afterwards).
 *   fe_narrow file.ivf threads limit   (limit 1: real int16 range; > 1: magnitudes >= limit fall back) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vp9hip_fe.h"

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  fseek(f, 0, SEEK_END);
  const long fsz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *file = (uint8_t *)malloc((size_t)fsz);
  if (fread(file, 1, (size_t)fsz, f) != (size_t)fsz) return 2;
  fclose(f);
  const int threads = atoi(argv[2]), limit = atoi(argv[3]);
  vp9hip_fe *a = NULL, *b = NULL;
  if (vp9hip_fe_create(&a, NULL, NULL, NULL, threads) || vp9hip_fe_create(&b, NULL, NULL, NULL, threads)) return 3;
  vp9hip_fe_set_narrow_slots(a, limit);
  const size_t hdr = (size_t)file[6] | ((size_t)file[7] << 8);
  int frames = 0, narrow = 0;
  long long coefs = 0;
  for (size_t pos = hdr; pos + 12 <= (size_t)fsz;) {
    const size_t psz = (size_t)file[pos] | ((size_t)file[pos + 1] << 8) | ((size_t)file[pos + 2] << 16) | ((size_t)file[pos + 3] << 24);
    pos += 12;
    if (pos + psz > (size_t)fsz) break;
    uint32_t sizes[8];
    const int nf = vp9hip_fe_split_superframe(file + pos, psz, sizes);
    size_t off = 0;
    for (int k = 0; k < nf; ++k) {
      if (nf > 1 && sizes[k] == 0) continue;
      vp9hip_fe_frame x, y;
      if (vp9hip_fe_parse(a, file + pos + off, sizes[k], &x) || vp9hip_fe_parse(b, file + pos + off, sizes[k], &y)) {
        fprintf(stderr, "fe_narrow: frame %d refused: %s / %s\n", frames, vp9hip_fe_error(a), vp9hip_fe_error(b));
        return 1;
      }
      off += sizes[k];
      ++frames;
      if (x.show_existing) continue;
      if (y.layout.narrow) {
        fprintf(stderr, "fe_narrow: frame %d: int16 slots without being asked\n", frames - 1);
        return 1;
      }
      if (x.n_blocks != y.n_blocks || memcmp(x.blocks, y.blocks, sizeof(vp9hip_block) * (size_t)x.n_blocks) ||
          memcmp(x.layout.block_off, y.layout.block_off, sizeof(uint32_t) * 3 * (size_t)x.n_blocks) ||
          x.layout.n_regions != y.layout.n_regions || x.layout.total != y.layout.total || !x.layout.compact) {
        fprintf(stderr, "fe_narrow: frame %d: lists differ between the two slot widths\n", frames - 1);
        return 1;
      }
      narrow += x.layout.narrow != 0;
      for (int64_t r = 0; r < x.layout.n_regions; ++r) {
        const vp9hip_coeff_region *g = &x.layout.regions[r], *gy = &y.layout.regions[r];
        if (g->plane != gy->plane || g->start != gy->start || g->count != gy->count) {
          fprintf(stderr, "fe_narrow: frame %d: region %lld differs\n", frames - 1, (long long)r);
          return 1;
        }
        const int32_t *w = y.dqcoeff[g->plane] + g->start;
        for (int64_t i = 0; i < g->count; ++i) {
          const int32_t v = x.layout.narrow ? (int32_t)((const int16_t *)x.dqcoeff[g->plane])[g->start + i] : x.dqcoeff[g->plane][g->start + i];
          if (v != w[i]) {
            fprintf(stderr, "fe_narrow: frame %d plane %d coefficient %lld: %d, int32 slots hold %d\n", frames - 1, g->plane,
                    (long long)(g->start + i), v, w[i]);
            return 1;
          }
          if (x.layout.narrow && limit > 1 && (w[i] >= limit || w[i] <= -limit)) {
            fprintf(stderr, "fe_narrow: frame %d kept int16 slots although %d is over the limit\n", frames - 1, w[i]);
            return 1;
          }
        }
        coefs += g->count;
      }
    }
    pos += psz;
  }
  printf("fe_narrow: %d frames, %d with int16 slots, %d parsed again with int32 slots, %lld coefficients equal\n", frames, narrow,
         vp9hip_fe_wide_frames(a), coefs);
  vp9hip_fe_destroy(a);
  vp9hip_fe_destroy(b);
  free(file);
  return 0;
}
