/*
 * ref_decodeframe_prelude.h — TEST INFRASTRUCTURE (oracle build only).
 *
 * Force-included (gcc -include) in front of the reference's libvpx/vp9/decoder/vp9_decodeframe.c,
 * which is compiled where it lies.  That file declares `MODE_INFO *set_offsets(...)` without
 * `static` at :54 and defines it `static` at :830; MSVC (the reference's compiler) accepts that,
 * gcc rejects "static declaration follows non-static".  Declaring the function static FIRST makes
 * both later declarations legal C (C11 6.2.2p5: a declaration without storage class takes the
 * linkage of the visible prior one).  Nothing else is declared or defined here.
 */
#include "./vpx_config.h"
#include "vp9/common/vp9_onyxc_int.h"
static MODE_INFO *set_offsets(VP9_COMMON *const cm, MACROBLOCKD *const xd, BLOCK_SIZE bsize, int mi_row, int mi_col,
                              int bw, int bh, int x_mis, int y_mis, int bwl, int bhl);
