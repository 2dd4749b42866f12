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
  E9  vp9hip_shim_mark(pbi, k) at five points of decode_tiles: where the host time of a frame goes, printed
      with VP9HIP_SHIM_TRACE=1 (no effect otherwise)
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
    open(dst, "w", encoding="utf-8", errors="surrogateescape").write(t)


def main():
    if sys.argv[1] == "--decoder-c":
        return patch_decoder_c(sys.argv[2], sys.argv[3])
    src, dst = sys.argv[1], sys.argv[2]
    gpu_lf = "--keep-cpu-loop-filter" not in sys.argv[3:]
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

    # E5
    t = replace_once(t, "    X_Fuel(pbi);\n",
                     "    if (get_frame_new_buffer(&pbi->common)->flags & YV12_FLAG_HIGHBITDEPTH) X_Fuel(pbi);\n", "E5 X_Fuel")

    open(dst, "w", encoding="utf-8", errors="surrogateescape").write(t)


if __name__ == "__main__":
    main()
