/* Robustness harness (CPU, built with -fsanitize=address,undefined by tests/test_fe_fuzz.py): damaged VP9 frames
 * through vp9hip_fe_parse and, when the front-end accepts them, through vp9hip_pack_frame — neither may touch memory
 * it does not own, whatever the bytes say; what they hand on must be either refused or within the limits the
 * kernels assume (the packer's own checks).   fe_fuzz file.ivf iterations seed */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vp9hip_fe.h"

static uint32_t rng_state;
static uint32_t rnd(void) {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

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
  const int iters = atoi(argv[2]);
  rng_state = (uint32_t)atoi(argv[3]);
  const size_t hdr = (size_t)file[6] | ((size_t)file[7] << 8);
  int parsed = 0, refused = 0, packed = 0, pack_refused = 0;
  for (int it = 0; it < iters; ++it) {
    vp9hip_fe *fe = NULL;
    vp9hip_packer *pk = NULL;
    if (vp9hip_fe_create(&fe, NULL, NULL, NULL, 1 + (it & 1)) || vp9hip_packer_create(&pk)) return 3;
    const int kind = it % 4; /* 0: flip bytes, 1: truncate, 2: flip bits in the headers only, 3: garbage tail */
    for (size_t pos = hdr; pos + 12 <= (size_t)fsz;) {
      const size_t psz = (size_t)file[pos] | ((size_t)file[pos + 1] << 8) | ((size_t)file[pos + 2] << 16) | ((size_t)file[pos + 3] << 24);
      pos += 12;
      if (pos + psz > (size_t)fsz) break;
      uint8_t *pkt = (uint8_t *)malloc(psz + 1); /* exact size: an overread is an ASan report */
      memcpy(pkt, file + pos, psz);
      size_t use = psz;
      if (rnd() % 3 == 0) {
        if (kind == 0)
          for (int k = 0; k < 1 + (int)(rnd() % 6); ++k) pkt[rnd() % psz] ^= (uint8_t)(1u << (rnd() % 8));
        else if (kind == 1)
          use = 1 + rnd() % psz;
        else if (kind == 2)
          for (int k = 0; k < 1 + (int)(rnd() % 3); ++k) pkt[rnd() % (psz < 24 ? psz : 24)] ^= (uint8_t)(1u << (rnd() % 8));
        else
          for (size_t k = psz / 2 + rnd() % (psz / 2 + 1); k < psz; ++k) pkt[k] = (uint8_t)rnd();
      }
      uint32_t sizes[8];
      const int nf = vp9hip_fe_split_superframe(pkt, use, sizes);
      size_t off = 0;
      for (int k = 0; k < nf; ++k) {
        vp9hip_fe_frame fr;
        if (nf > 1 && sizes[k] == 0) continue;
        if (off + sizes[k] > use) break;
        if (vp9hip_fe_parse(fe, pkt + off, sizes[k], &fr)) {
          ++refused;
        } else {
          ++parsed;
          if (!fr.show_existing) {
            vp9hip_packed out;
            if (vp9hip_pack_frame(pk, &fr.params, fr.blocks, fr.n_blocks, &fr.layout, &out))
              ++pack_refused;
            else
              ++packed;
          }
        }
        off += sizes[k];
      }
      free(pkt);
      pos += psz;
    }
    vp9hip_packer_destroy(pk);
    vp9hip_fe_destroy(fe);
  }
  printf("fe_fuzz: %d frames parsed, %d refused by the front-end; %d packed, %d refused by the packer\n", parsed, refused, packed, pack_refused);
  free(file);
  return 0;
}
