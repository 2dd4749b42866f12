#!/usr/bin/env python3
"""oracle/patch_decodeframe.py — the INTEGRATION.md §3 edits of the reference's frame driver, as a recipe.

    patch_decodeframe.py <reference vp9_decodeframe.c> <output .c> [--keep-cpu-loop-filter]

Reads the reference's libvpx/vp9/decoder/vp9_decodeframe.c, applies the edits a maintainer makes to
run 8-bit (and high-bitdepth) streams with the inverse transforms and the loop filter behind the
wrap_cuda_* entry points, and writes the result to <output> — a scratch file the build compiles and
deletes (oracle/build_refvpx.sh); neither the input nor the output is kept in this repository.
Every edit is anchored on the exact reference text and the script fails if an anchor is missing or
ambiguous.  The same patched object is linked against the CPU stream oracle
(oracle/ref_stream_wraps.c) and against libvp9hip_shim.so (the product).

Edits (decode_tiles = vp9_decodeframe.c:2303-2639):
  E1  after `initBuf(frameBuffer, n, cm);` (:2316): vp9hip_shim_attach_frame_buffer(pbi, frameBuffer)
      [+ vp9hip_shim_set_gpu_loop_filter(pbi, 1)]                               (mode B [mode C])
  E2  delete both "frame idct" loops (:2443-2534): phase B, the CPU inverse transforms into the
      int64 residual plane — type-confused on 8-bit frame buffers (inverse_transform_block_inter
      hands a tran_high_t* to vp9_idct4x4_add, :205-214), and the second loop reads size_for_mb past
      its allocation unless width and height are multiples of 64
  E3  [mode C] the two `if (cm->lf.filter_level && !cm->skip_loop_filter)` of phase E (:2589, :2612)
      become `if (0)`: the frame wrap_cuda_intra_prediction delivers is already filtered
  E4  initBuf (:2244-2246): no malloc + memset of the frame-sized int64 residual plane
  E5  `X_Fuel(pbi);` (:3567) only for high-bitdepth buffers (it reinterprets the buffer as uint16)
  E7  initBuf / freeBuf (:2244-2297): the frame-sized eob plane and the three coefficient arrays are
      taken from vp9hip_shim_frame_memory() — page-locked, kept from frame to frame — instead of
      malloc + free per frame (~70 MB of fresh pages per 1440p frame, and pageable memory makes every
      host-to-device copy a staged, synchronous one)
  E8  (other file: `patch_decodeframe.py --decoder-c <vp9_decoder.c> <out>`) vp9_decoder_remove calls
      vp9hip_shim_release(pbi): GPU state does not outlive the decoder (`vpxdec --loops=N`)
  E13 (other file, `--decoder-c ... --rtcd`) initialize_dec (vp9_decoder.c:39-49) calls vp9hip_install_rtcd()
      (shim/vp9hip_rtcd_install.c): the block-level integration, run-time dispatch pointers -> _hip twins
  E9  vp9hip_shim_mark(pbi, k) at five points of decode_tiles: where the host time of a frame goes, printed
      with VP9HIP_SHIM_TRACE=1 (no effect otherwise)
  E10 (--mt) tile-parallel entropy stage: one thread per tile column with private list segments /
      coefficient regions / counts / error trap, merged into the canonical lists afterwards (f2)
  E11 detoken_block (:951, :969, :1000, :1018): the eob of a transform block goes to a plane with one int per 4x4
      position (16 times denser than one int per sample position: the packer's reads stay in cache);
      vp9hip_shim_set_eob_layout(pbi, 2) tells the other side
  E12 (--mt) detoken_block (:958, :971, :1006, :1020): a transform block's slot takes only the rows the clearing
      rule a few lines further down leaves non-zero (vp9hip_coeff_extent, include/vp9hip_pack.h) and nothing at
      eob 0; the slots of a tile column follow each other without gaps.  The threads copy, and the frame driver
      uploads, a fraction of the bytes (S-2160: 49.6 -> 26.8 MB per frame, S-1440: 3.8 -> 1.7).  vp9hip_shim_set_tile_layout(...,
      VP9HIP_SHIM_COEFF_COMPACT) tells the other side; the serial loop (row_mt, inverse tile order) keeps full slots.
  E6  `int n = cm->width * cm->height;` (:2314) sizes dqcoeff[plane] (initBuf :2266) and the block
      lists; coefficient slots cover whole transform blocks, so a frame whose size is not a multiple
      of 8 (or whose last 32x32 transform block overhangs the frame) overruns it (heap corruption,
      e.g. 350x286).  n becomes the superblock-aligned area.
"""
import sys


def replace_once(text, old, new, what, start=0):
    i = text.find(old, start)
    if i < 0:
        sys.exit(f"patch_decodeframe: anchor not found: {what}")
    if text.find(old, i + 1) >= 0 and start == 0:
        sys.exit(f"patch_decodeframe: anchor ambiguous: {what}")
    return text[:i] + new + text[i + len(old):]


def patch_decoder_c(src, dst):
    """E8: vp9_decoder_remove (libvpx/vp9/decoder/vp9_decoder.c:216) releases the shim's per-decoder state."""
    t = open(src, encoding="utf-8", errors="surrogateescape").read()
    t = replace_once(t, "void vp9_decoder_remove(VP9Decoder *pbi) {\n  int i;\n\n  if (!pbi) return;\n",
                     "#include \"vp9hip_libvpx_shim.h\"\nvoid vp9_decoder_remove(VP9Decoder *pbi) {\n  int i;\n\n"
                     "  if (!pbi) return;\n  vp9hip_shim_release(pbi);\n", "E8 vp9_decoder_remove")
    if "--rtcd" in sys.argv:
        # E13 (bring-up mode, shim/build/vpxdec_rtcd only): the run-time dispatch pointers get their _hip twins right after
        # the reference's own setup and before vp9_init_intra_predictors copies them into its tables
        t = replace_once(t, "    vpx_scale_rtcd();\n    vp9_init_intra_predictors();\n",
                         "    vpx_scale_rtcd();\n    { extern void vp9hip_install_rtcd(void); vp9hip_install_rtcd(); }\n"
                         "    vp9_init_intra_predictors();\n", "E13 initialize_dec")
    open(dst, "w", encoding="utf-8", errors="surrogateescape").write(t)


MT_CODE = r'''
/* ---- E10: tile-parallel entropy stage (SURVEY 8f-2) — inserted by oracle/patch_decodeframe.py --mt ---------
 * One thread per TILE COLUMN walks its tiles top to bottom (tile rows share the above context) with private
 * cursors: a segment of block-list entries, a region of every plane's coefficient array, its own FRAME_COUNTS
 * and error trap (as libvpx's tile_worker_hook does, :2663).  Afterwards the segments are merged into the
 * canonical decode-order lists the entry points take; coefficient slots stay where the threads put them, and
 * every block's slot offsets go to the shim (vp9hip_shim_set_tile_layout). */
#include <setjmp.h>
typedef struct {
  ModeInfoBuf mb;
  int *subsize;
  uint32_t (*coef_off)[3];
  int n_blocks;
  int64_t coef_start[3], coef_used[3];
} vp9hip_tile_seg;
typedef struct {
  VP9Decoder *pbi;
  int tile_rows, tile_cols, sb_cols;
  frameBuf *fb;
  int *size_for_mb;
  vp9hip_tile_seg seg[64];
} vp9hip_mt_job;

static void vp9hip_parse_tile_col(void *argp, int col) {
  vp9hip_mt_job *job = (vp9hip_mt_job *)argp;
  VP9Decoder *const pbi = job->pbi;
  VP9_COMMON *const cm = &pbi->common;
  vp9hip_tile_seg *seg = &job->seg[col];
  frameBuf fb = *job->fb;
  tran_low_t *dq[3];
  ModeInfoBuf mb = seg->mb;
  int *subsize = seg->subsize;
  TileWorkerData *volatile td = NULL;
  int p, tile_row, mi_row, mi_col;
  fb.dqcoeff = dq;
  for (p = 0; p < 3; ++p) {
    dq[p] = job->fb->dqcoeff[p] + seg->coef_start[p];
    vp9hip_tl_coef_base[p] = job->fb->dqcoeff[p];
  }
  vp9hip_tl_coef_off = seg->coef_off;
  vp9hip_tl_block = 0;
  for (tile_row = 0; tile_row < job->tile_rows; ++tile_row) {
    TileInfo tile;
    td = pbi->tile_worker_data + job->tile_cols * tile_row + col;
    vp9_tile_set_row(&tile, cm, tile_row);
    vp9_tile_set_col(&tile, cm, col);
    vp9_zero(td->counts);
    td->xd.counts = cm->frame_parallel_decoding_mode ? NULL : &td->counts;
    td->xd.error_info = &td->error_info;
    td->error_info.setjmp = 1;
    if (setjmp(td->error_info.jmp)) {
      td->error_info.setjmp = 0;
      td->xd.corrupted = 1;
      break;
    }
    for (mi_row = tile.mi_row_start; mi_row < tile.mi_row_end; mi_row += MI_BLOCK_SIZE) {
      vp9_zero(td->xd.left_context);
      vp9_zero(td->xd.left_seg_context);
      for (mi_col = tile.mi_col_start; mi_col < tile.mi_col_end; mi_col += MI_BLOCK_SIZE) {
        int *cnt = &job->size_for_mb[(mi_row >> 3) * job->sb_cols + (mi_col >> 3)];
        decode_partition(td, pbi, mi_row, mi_col, BLOCK_64X64, 4, &fb, &mb, cnt, subsize);
        mb.mi_col += *cnt;
        mb.mi += *cnt;
        mb.mi_row += *cnt;
        mb.bhl += *cnt;
        mb.bwl += *cnt;
        subsize += *cnt;
      }
    }
    td->error_info.setjmp = 0;
  }
  seg->n_blocks = vp9hip_tl_block;
  for (p = 0; p < 3; ++p) seg->coef_used[p] = dq[p] - (job->fb->dqcoeff[p] + seg->coef_start[p]);
  vp9hip_tl_coef_off = NULL;
}

static void vp9hip_parse_frame_mt(VP9Decoder *pbi, int tile_rows, int tile_cols, frameBuf *frameBuffer, ModeInfoBuf *MiBuf,
                                  int *size_for_mb, int *subsize_array, int n) {
  VP9_COMMON *const cm = &pbi->common;
  const int sb_cols = mi_cols_aligned_to_sb(cm->mi_cols) >> 3, sb_rows = mi_cols_aligned_to_sb(cm->mi_rows) >> 3;
  const int cap = sb_cols * sb_rows * 64; /* blocks are 8x8 or larger */
  vp9hip_mt_job *job = (vp9hip_mt_job *)calloc(1, sizeof(*job));
  MODE_INFO **s_mi = (MODE_INFO **)malloc(sizeof(MODE_INFO *) * cap);
  int *s_int = (int *)malloc(sizeof(int) * 5 * (size_t)cap);
  uint32_t(*s_off)[3] = (uint32_t(*)[3])malloc(sizeof(uint32_t) * 3 * (size_t)cap);
  int64_t start[64][3], used[64][3];
  int t, p, r, c, total = 0, corrupted = 0;
  if (!job || !s_mi || !s_int || !s_off) vpx_internal_error(&cm->error, VPX_CODEC_MEM_ERROR, "tile-parallel parse: out of memory");
  job->pbi = pbi;
  job->tile_rows = tile_rows;
  job->tile_cols = tile_cols;
  job->sb_cols = sb_cols;
  job->fb = frameBuffer;
  job->size_for_mb = size_for_mb;
  for (t = 0; t < tile_cols; ++t) {
    TileInfo tile;
    vp9_tile_set_col(&tile, cm, t);
    {
      const int first = (tile.mi_col_start >> 3) * sb_rows * 64; /* entries / coefficient share before this tile */
      vp9hip_tile_seg *seg = &job->seg[t];
      seg->mb.mi = s_mi + first;
      seg->mb.mi_row = s_int + first;
      seg->mb.mi_col = s_int + cap + first;
      seg->mb.bwl = s_int + 2 * (size_t)cap + first;
      seg->mb.bhl = s_int + 3 * (size_t)cap + first;
      seg->subsize = s_int + 4 * (size_t)cap + first;
      seg->coef_off = s_off + first;
      for (p = 0; p < 3; ++p) seg->coef_start[p] = (int64_t)(tile.mi_col_start >> 3) * sb_rows * 4096;
    }
  }
  (void)n;
  vp9hip_shim_run_parallel(pbi, tile_cols, vp9hip_parse_tile_col, job);
  for (t = 0; t < tile_cols * tile_rows; ++t) corrupted |= pbi->tile_worker_data[t].xd.corrupted;
  if (!corrupted) {
    /* merge: superblocks in raster order = the order of the serial loop */
    int cur[64];
    uint32_t *block_off;
    for (t = 0; t < tile_cols; ++t) {
      cur[t] = 0;
      total += job->seg[t].n_blocks;
    }
    block_off = vp9hip_shim_block_off_buffer(pbi, total);
    {
      int k = 0, tcol[256];
      for (t = 0; t < tile_cols; ++t) {
        TileInfo tile;
        vp9_tile_set_col(&tile, cm, t);
        for (c = tile.mi_col_start >> 3; c < (tile.mi_col_end + 7) >> 3 && c < 256; ++c) tcol[c] = t;
      }
      for (r = 0; r < sb_rows; ++r)
        for (c = 0; c < sb_cols; ++c) {
          const int cnt = size_for_mb[r * sb_cols + c];
          vp9hip_tile_seg *seg = &job->seg[tcol[c]];
          const int a = cur[tcol[c]];
          memcpy(MiBuf->mi + k, seg->mb.mi + a, sizeof(MODE_INFO *) * cnt);
          memcpy(MiBuf->mi_row + k, seg->mb.mi_row + a, sizeof(int) * cnt);
          memcpy(MiBuf->mi_col + k, seg->mb.mi_col + a, sizeof(int) * cnt);
          memcpy(MiBuf->bwl + k, seg->mb.bwl + a, sizeof(int) * cnt);
          memcpy(MiBuf->bhl + k, seg->mb.bhl + a, sizeof(int) * cnt);
          memcpy(subsize_array + k, seg->subsize + a, sizeof(int) * cnt);
          if (block_off) memcpy(block_off + 3 * (size_t)k, seg->coef_off + a, sizeof(uint32_t) * 3 * cnt);
          cur[tcol[c]] = a + cnt;
          k += cnt;
        }
    }
    for (t = 0; t < tile_cols; ++t)
      for (p = 0; p < 3; ++p) {
        start[t][p] = job->seg[t].coef_start[p];
        used[t][p] = job->seg[t].coef_used[p];
      }
    vp9hip_shim_set_tile_layout(pbi, total, tile_cols, &start[0][0], &used[0][0], VP9HIP_SHIM_COEFF_COMPACT);
    if (!cm->frame_parallel_decoding_mode)
      for (t = 0; t < tile_cols * tile_rows; ++t) vp9_accumulate_frame_counts(&cm->counts, &pbi->tile_worker_data[t].counts, 1);
  }
  free(s_off);
  free(s_int);
  free(s_mi);
  free(job);
  if (corrupted) vpx_internal_error(&cm->error, VPX_CODEC_CORRUPT_FRAME, "Failed to decode tile data");
}
'''


def patch_mt(t):
    """E10: tile-parallel entropy stage (the serial loop stays for row_mt / inverse tile order); E12: compact slots."""
    t = replace_once(t, "#define MAX_VP9_HEADER_SIZE 80\n",
                     "#define MAX_VP9_HEADER_SIZE 80\n"
                     "#include \"vp9hip_pack.h\" /* E12: vp9hip_coeff_extent */\n"
                     "/* E10: where the running tile-column thread records each block's coefficient slot offsets */\n"
                     "static __thread uint32_t (*vp9hip_tl_coef_off)[3];\n"
                     "static __thread tran_low_t *vp9hip_tl_coef_base[3];\n"
                     "static __thread int vp9hip_tl_block;\n", "E10 thread-locals")
    t = replace_once(t, "  if (!mi->skip) detoken_block(twd, mi, frameBuffer, mi_col, mi_row);\n",
                     "  if (vp9hip_tl_coef_off) {\n"
                     "    for (int p_ = 0; p_ < 3; ++p_)\n"
                     "      vp9hip_tl_coef_off[vp9hip_tl_block][p_] = (uint32_t)(frameBuffer->dqcoeff[p_] - vp9hip_tl_coef_base[p_]);\n"
                     "    ++vp9hip_tl_block;\n"
                     "  }\n"
                     "  if (!mi->skip) detoken_block(twd, mi, frameBuffer, mi_col, mi_row);\n", "E10 decode_block")
    t = replace_once(t, "static const uint8_t *decode_tiles(VP9Decoder *pbi, const uint8_t *data, const uint8_t *data_end) {",
                     MT_CODE + "\nstatic const uint8_t *decode_tiles(VP9Decoder *pbi, const uint8_t *data, const uint8_t *data_end) {",
                     "E10 helper insertion")
    # E12: compact coefficient slots while a tile-column thread is parsing
    a = t.find("static void detoken_block(")
    b = t.find("static void intra_decode(", a)
    body = t[a:b]
    old_copy = "          memcpy(frameBuffer->dqcoeff[plane], dq, n * sizeof(tran_low_t));\n"
    old_step = "          frameBuffer->dqcoeff[plane] += n;\n"
    if a < 0 or b < 0 or body.count(old_copy) != 2 or body.count(old_step) != 2 or body.count("const TX_TYPE tx_type") != 1:
        sys.exit("patch_decodeframe: anchor not found: E12 detoken_block")
    for txt in ("tx_type", "DCT_DCT"):  # the intra branch has the block's tx_type, the inter branch is DCT_DCT
        body = body.replace(old_copy, "          const int ext_ = vp9hip_tl_coef_off ? vp9hip_coeff_extent(eob, %s, tx_size) : n;\n"
                            "          memcpy(frameBuffer->dqcoeff[plane], dq, ext_ * sizeof(tran_low_t));\n" % txt, 1)
    body = body.replace(old_step, "          frameBuffer->dqcoeff[plane] += ext_;\n")
    t = t[:a] + body + t[b:]
    a = t.find("  //entropy decoder\n")
    b = t.find("  //go to start\n", a)
    if a < 0 or b < 0:
        sys.exit("patch_decodeframe: anchor not found: E10 entropy loop")
    loop = t[a + len("  //entropy decoder\n"):b]
    if loop.count("decode_partition(tile_data, pbi, mi_row, mi_col, BLOCK_64X64, 4, frameBuffer, &MiBuf, size_for_mb, subsize_array);") != 1:
        sys.exit("patch_decodeframe: E10 range does not look like the entropy loop")
    marks = ""
    if loop.rstrip().endswith("vp9hip_shim_mark(pbi, 2);"):
        loop = loop[:loop.rstrip().rfind("vp9hip_shim_mark(pbi, 2);")]
        marks = "  vp9hip_shim_mark(pbi, 2);\n"
    t = (t[:a] + "  //entropy decoder\n  if (pbi->row_mt != 1 && !pbi->inv_tile_order) {\n"
         "    vp9hip_parse_frame_mt(pbi, tile_rows, tile_cols, frameBuffer, &MiBuf, size_for_mb, subsize_array, n);\n"
         "  } else {\n" + loop + "  }\n" + marks + t[b:])
    return t


def main():
    if sys.argv[1] == "--decoder-c":
        return patch_decoder_c(sys.argv[2], sys.argv[3])
    src, dst = sys.argv[1], sys.argv[2]
    gpu_lf = "--keep-cpu-loop-filter" not in sys.argv[3:]
    mt = "--mt" in sys.argv[3:]
    t = open(src, encoding="utf-8", errors="surrogateescape").read()

    # E1
    hook = "  initBuf(frameBuffer, n, cm);\n  vp9hip_shim_attach_frame_buffer(pbi, frameBuffer);\n"
    if gpu_lf:
        hook += "  vp9hip_shim_set_gpu_loop_filter(pbi, 1);\n"
    t = replace_once(t, "  initBuf(frameBuffer, n, cm);\n", hook, "E1 initBuf call")

    # E2
    a = t.find("  //frame idct\n")
    b = t.find("  if (cm->frame_type == INTER_FRAME) {\n", a)
    if a < 0 or b < 0 or t.count("  //frame idct\n") != 1:
        sys.exit("patch_decodeframe: anchor not found: E2 phase B")
    if t[a:b].count("inter_decode(") != 2 or t[a:b].count("intra_decode(") != 2:
        sys.exit("patch_decodeframe: E2 range does not look like the two transform loops")
    t = t[:a] + "  /* phase B (CPU inverse transforms) removed: they run behind wrap_cuda_* */\n" + t[b:]

    # E3
    if gpu_lf:
        c = t.find("  wrap_cuda_intra_prediction(&gpu_copy, &gpu_run, size_for_mb, &MiBuf, cm, pbi, tile_rows, tile_cols, frameBuffer);")
        if c < 0:
            sys.exit("patch_decodeframe: anchor not found: E3 intra call")
        end = t.find("  // Get last tile data.\n", c)
        body = t[c:end]
        cond = "if (cm->lf.filter_level && !cm->skip_loop_filter) {"
        if end < 0 or body.count(cond) != 2:
            sys.exit("patch_decodeframe: E3 expects two loop-filter conditions after the intra call")
        t = t[:c] + body.replace(cond, "if (0 /* phase E runs behind wrap_cuda_intra_prediction */) {") + t[end:]

    # E4
    t = replace_once(t, "  buffer->residuals = (tran_high_t *)malloc(src->frame_size * sizeof(tran_high_t));\n",
                     "  buffer->residuals = NULL; /* residual plane not needed: transforms run behind wrap_cuda_* */\n",
                     "E4 residual malloc")
    t = replace_once(t, "  memset(buffer->residuals, 0, src->frame_size * sizeof(tran_high_t));\n", "", "E4 residual memset")

    # E6
    t = replace_once(t, "  int n = cm->width * cm->height;\n",
                     "  int n = 64 * mi_cols_aligned_to_sb(cm->mi_cols) * mi_cols_aligned_to_sb(cm->mi_rows);\n", "E6 n")

    # E7
    t = replace_once(t, "  buffer->eob = (int *)malloc(src->frame_size * sizeof(int));\n",
                     "  buffer->eob = (int *)vp9hip_shim_frame_memory(cm, 3, src->frame_size * sizeof(int));\n", "E7 eob malloc")
    t = replace_once(t, "    buffer->dqcoeff[plane] = (tran_low_t *)malloc(n * sizeof(tran_low_t));\n",
                     "    buffer->dqcoeff[plane] = (tran_low_t *)vp9hip_shim_frame_memory(cm, plane, n * sizeof(tran_low_t));\n",
                     "E7 dqcoeff malloc")
    t = replace_once(t, "static void freeBuf(frameBuf *buffer) {\n  for (int plane = 0; plane < MAX_MB_PLANE; ++plane) {\n"
                        "    free(buffer->dqcoeff[plane]);\n  }\n\n  free(buffer->residuals);\n  free(buffer->eob);\n",
                     "static void freeBuf(frameBuf *buffer) {\n  /* eob plane + coefficient arrays belong to the shim */\n"
                     "  free(buffer->residuals);\n", "E7 freeBuf")
    t = replace_once(t, "static void initBuf(frameBuf *buffer, int n, VP9_COMMON *cm) {",
                     "#include \"vp9hip_libvpx_shim.h\"\nstatic void initBuf(frameBuf *buffer, int n, VP9_COMMON *cm) {", "E7 include")

    # E9 (measurement only): phase marks of decode_tiles for VP9HIP_SHIM_TRACE
    t = replace_once(t, "  frameBuf *frameBuffer = (frameBuf *)malloc(sizeof(frameBuf));\n",
                     "  vp9hip_shim_mark(pbi, 0);\n  frameBuf *frameBuffer = (frameBuf *)malloc(sizeof(frameBuf));\n", "E9 mark 0")
    t = replace_once(t, "  //entropy decoder\n", "  vp9hip_shim_mark(pbi, 1);\n  //entropy decoder\n", "E9 mark 1")
    i = t.find("  //go to start\n")
    if i < 0:
        sys.exit("patch_decodeframe: anchor not found: E9 mark 2")
    t = t[:i] + "  vp9hip_shim_mark(pbi, 2);\n" + t[i:]
    t = replace_once(t, "  // Get last tile data.\n", "  vp9hip_shim_mark(pbi, 3);\n  // Get last tile data.\n", "E9 mark 3")
    t = replace_once(t, "  ++fr;\n", "  vp9hip_shim_mark(pbi, 4);\n  ++fr;\n", "E9 mark 4")

    # E11: one eob per 4x4 position instead of one per sample position
    old_base = "      int *eob_buf = frameBuffer->plane_eob[plane] + by * stride + bx;\n"
    old_store = "eob_buf[4 * row * stride + 4 * col] = eob;"
    a = t.find("static void detoken_block(")
    b = t.find("static void intra_decode(", a)
    body = t[a:b]
    if a < 0 or b < 0 or body.count(old_base) != 2 or body.count(old_store) != 2:
        sys.exit("patch_decodeframe: anchor not found: E11 detoken_block")
    body = body.replace(old_base, "      int *eob_buf = frameBuffer->plane_eob[plane] + (by >> 2) * (stride >> 2) + (bx >> 2);\n")
    body = body.replace(old_store, "eob_buf[row * (stride >> 2) + col] = eob;")
    t = t[:a] + body + t[b:]
    t = replace_once(t, "  vp9hip_shim_attach_frame_buffer(pbi, frameBuffer);\n",
                     "  vp9hip_shim_attach_frame_buffer(pbi, frameBuffer);\n  vp9hip_shim_set_eob_layout(pbi, 2);\n", "E11 layout call")

    # E5
    t = replace_once(t, "    X_Fuel(pbi);\n",
                     "    if (get_frame_new_buffer(&pbi->common)->flags & YV12_FLAG_HIGHBITDEPTH) X_Fuel(pbi);\n", "E5 X_Fuel")

    if mt:
        t = patch_mt(t)
    open(dst, "w", encoding="utf-8", errors="surrogateescape").write(t)


if __name__ == "__main__":
    main()
