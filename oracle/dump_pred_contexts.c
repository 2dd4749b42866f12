/* TEST INFRASTRUCTURE (oracle/): the reference's own neighbour-context functions (libvpx/vp9/common/vp9_pred_common.h /
 * .c, compiled from the reference's sources into oracle/_ref/vpx/libvpxfull.a) over an enumeration of (above, left)
 * neighbour pairs — every combination of absent / intra / single-reference / compound neighbours under every
 * sign-bias pattern, plus random skip / transform-size / filter fields.  tests/golden/make_pred_ctx.py runs it and
 * stores the table (tests/golden/pred_ctx.npz); tests/test_fe_contexts.py holds the product's restatement
 * (cuda-vp9_amd/csrc/fe/vp9fe.c) against it.  Output: int32 records of 24 values on stdout. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "./vpx_config.h"
#include "vp9/common/vp9_onyxc_int.h"
#include "vp9/common/vp9_pred_common.h"
#include "vp9/common/vp9_common_data.h"

static unsigned rng = 12345;
static unsigned rnd(void) {
  rng = rng * 1664525u + 1013904223u;
  return rng >> 8;
}

/* kind: 0 absent, 1 intra, 2..4 single LAST/GOLDEN/ALTREF, 5..10 compound (ordered pairs of distinct references) */
static const int kPair[6][2] = { { 1, 2 }, { 1, 3 }, { 2, 1 }, { 2, 3 }, { 3, 1 }, { 3, 2 } };
static void fill(MODE_INFO *m, int kind) {
  memset(m, 0, sizeof(*m));
  m->skip = rnd() & 1;
  m->tx_size = (TX_SIZE)(rnd() & 3);
  m->sb_type = BLOCK_64X64;
  if (kind == 1) {
    m->ref_frame[0] = INTRA_FRAME;
    m->ref_frame[1] = NONE;
    m->interp_filter = SWITCHABLE_FILTERS;
  } else if (kind <= 4) {
    m->ref_frame[0] = (MV_REFERENCE_FRAME)(kind - 1);
    m->ref_frame[1] = NONE;
    m->interp_filter = (INTERP_FILTER)(rnd() % 3);
  } else {
    m->ref_frame[0] = (MV_REFERENCE_FRAME)kPair[kind - 5][0];
    m->ref_frame[1] = (MV_REFERENCE_FRAME)kPair[kind - 5][1];
    m->interp_filter = (INTERP_FILTER)(rnd() % 3);
  }
}

int main(void) {
  static VP9_COMMON cm;
  static MACROBLOCKD xd;
  static const BLOCK_SIZE cur_sizes[3] = { BLOCK_8X8, BLOCK_16X16, BLOCK_64X64 }; /* max tx 8x8 / 16x16 / 32x32 */
  for (int rep = 0; rep < 4; ++rep)
    for (int bias = 0; bias < 8; ++bias)
      for (int ka = 0; ka < 11; ++ka)
        for (int kl = 0; kl < 11; ++kl) {
          MODE_INFO above, left, cur, *cur_p = &cur;
          int rec[24];
          memset(&cm, 0, sizeof(cm));
          memset(&xd, 0, sizeof(xd));
          cm.ref_frame_sign_bias[LAST_FRAME] = bias & 1;
          cm.ref_frame_sign_bias[GOLDEN_FRAME] = (bias >> 1) & 1;
          cm.ref_frame_sign_bias[ALTREF_FRAME] = (bias >> 2) & 1;
          vp9_setup_compound_reference_mode(&cm);
          if (ka) fill(&above, ka);
          if (kl) fill(&left, kl);
          memset(&cur, 0, sizeof(cur));
          cur.sb_type = cur_sizes[rnd() % 3];
          xd.mi = &cur_p;
          xd.above_mi = ka ? &above : NULL;
          xd.left_mi = kl ? &left : NULL;
          rec[0] = ka ? 1 : 0;
          rec[1] = ka ? above.ref_frame[0] : 0;
          rec[2] = ka ? above.ref_frame[1] : -1;
          rec[3] = ka ? above.skip : 0;
          rec[4] = ka ? above.tx_size : 0;
          rec[5] = ka ? above.interp_filter : 0;
          rec[6] = kl ? 1 : 0;
          rec[7] = kl ? left.ref_frame[0] : 0;
          rec[8] = kl ? left.ref_frame[1] : -1;
          rec[9] = kl ? left.skip : 0;
          rec[10] = kl ? left.tx_size : 0;
          rec[11] = kl ? left.interp_filter : 0;
          rec[12] = cm.ref_frame_sign_bias[1];
          rec[13] = cm.ref_frame_sign_bias[2];
          rec[14] = cm.ref_frame_sign_bias[3];
          rec[15] = max_txsize_lookup[cur.sb_type];
          rec[16] = vp9_get_skip_context(&xd);
          rec[17] = get_intra_inter_context(&xd);
          rec[18] = get_pred_context_switchable_interp(&xd);
          rec[19] = get_tx_size_context(&xd);
          rec[20] = vp9_get_reference_mode_context(&cm, &xd);
          rec[21] = vp9_get_pred_context_comp_ref_p(&cm, &xd);
          rec[22] = vp9_get_pred_context_single_ref_p1(&xd);
          rec[23] = vp9_get_pred_context_single_ref_p2(&xd);
          fwrite(rec, sizeof(int), 24, stdout);
        }
  return 0;
}
