#!/bin/bash
# oracle/build_refvpx.sh — TEST INFRASTRUCTURE.  Builds the reference's own vpxdec / vpxenc from the
# reference's sources WHERE THEY LIE under $REF (nothing is copied into this repository), with its
# checked-in Win64 configuration headers (vpx-master/vpx_config.h, *_rtcd.h, vpx_version.h), plain gcc
# and no reference build system (no configure, no rtcd.pl, no make of theirs).  Outputs only under
# oracle/_ref/vpx/ (git-ignored; built files travel to the GPU box) and, for the HIP-linked decoder,
# shim/build/.
#
#   oracle/_ref/vpx/libvpxfull.a   every libvpx C file of the tree (vp8, vp9, vpx_dsp, vpx_scale, ...)
#   oracle/_ref/vpx/vpxdec_cA      vpxdec + UNCHANGED vp9_decodeframe.c + CPU wrap_cuda_* bodies
#                                  (oracle/ref_stream_wraps.c)          — the reference as it is, mode A
#   oracle/_ref/vpx/vpxdec_c       vpxdec + PATCHED frame driver (oracle/patch_decodeframe.py, the
#                                  INTEGRATION.md edits) + CPU bodies   — the stream oracle, 8/10/12 bit
#   oracle/_ref/vpx/vpxenc_c       the reference's encoder linked with that same decoder
#                                  (`--test-decode=fatal` pins the oracle; also synthesizes test streams)
#   oracle/_ref/vpx/ref_svc_encode oracle/ref_svc_encode.c: the reference's encoder API driven for spatial layers
#                                  (references of another size) and an intra-only frame, same decoder linked in
#   shim/build/vpxdec_hipA         vpxdec + UNCHANGED frame driver + libvp9hip_shim.so   (PRODUCT, mode A)
#   shim/build/vpxdec_hip          vpxdec + PATCHED frame driver  + libvp9hip_shim.so   (PRODUCT, mode C)
#   shim/build/vpxdec_hip_mt       the same + tile-parallel entropy stage (patch_decodeframe.py --mt, E10)
#
# What stands between the checked-in headers and a Linux gcc build, and how it is bridged without
# stand-ins for anything the image lacks:
#   * <cuda_runtime.h> (libvpx/vpx_dsp/vpx_convolve.h:16): NVIDIA's real header from the triton wheel.
#   * SIMD names in the Win64 rtcd headers (sources absent): oracle/gen_simd_map.py maps each to the
#     reference's own same-prototype *_c function; the 11 with a retyped *_c twin are composed from
#     reference functions in oracle/ref_absent_simd.c.
#   * `static` after non-static declaration of set_offsets (MSVC-only C): oracle/ref_decodeframe_prelude.h.
#   * the patched driver is written to a mktemp scratch file, compiled, and deleted.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${REF:-/root/reference}"
OUT="$HERE/_ref/vpx"
JOBS="${JOBS:-8}"
if [ ! -d "$REF/libvpx" ]; then echo "build_refvpx: $REF/libvpx absent, keeping prebuilt files under $OUT"; exit 0; fi
CUDA_INC="${CUDA_INC:-$(python3 -c "import triton,os;print(os.path.join(os.path.dirname(triton.__file__),'backends','nvidia','include'))")}"
[ -f "$CUDA_INC/cuda_runtime.h" ] || { echo "build_refvpx: cuda_runtime.h not found"; exit 1; }
L="$REF/libvpx"
mkdir -p "$OUT/obj" "$OUT/tools" "$ROOT/shim/build"
python3 "$HERE/gen_simd_map.py" "$REF" > "$OUT/simd_to_c.h"
INC="-I$REF/vpx-master -I$L -I$CUDA_INC -I$ROOT/include"
CF="-O2 -fPIC -fwrapv -w -ffunction-sections -fdata-sections $INC"
export REF L OUT CF HERE

# ---- libvpx: every C file (config-disabled ones compile to nothing or are skipped below) -------------
SRCS=$(ls $L/vp8/*.c $L/vp8/common/*.c $L/vp8/common/generic/*.c $L/vp8/decoder/*.c $L/vp8/encoder/*.c \
          $L/vp9/*.c $L/vp9/common/*.c $L/vp9/encoder/*.c $L/vpx_dsp/*.c $L/vpx_util/*.c \
          $L/vpx_scale/generic/*.c $L/vpx_scale/vpx_scale_rtcd.c $L/vpx_mem/vpx_mem.c $L/vpx/src/*.c \
          $L/vpx_ports/emms_mmx.c $REF/vpx-master/vpx_config.c \
          $L/vp9/decoder/vp9_decodemv.c $L/vp9/decoder/vp9_decoder.c $L/vp9/decoder/vp9_detokenize.c \
          $L/vp9/decoder/vp9_dsubexp.c $L/vp9/decoder/vp9_job_queue.c \
        | grep -v -e '/vp8/encoder/mr_dissim.c$' -e '/vp9/common/vp9_mfqe.c$' -e '/vp9/encoder/vp9_denoiser.c$')
# (those three belong to features the checked-in vpx_config.h disables: CONFIG_MULTI_RES_ENCODING,
#  CONFIG_VP9_POSTPROC, CONFIG_VP9_TEMPORAL_DENOISING)
cc_one() {
  f="$1"; o="$OUT/obj/$(echo "${f#$REF/}" | tr '/' '_' | sed 's/\.c$/.o/')"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -c "$f" -o "$o" || exit 255; fi
}
export -f cc_one
echo "$SRCS" | xargs -P "$JOBS" -I{} bash -c 'cc_one {}'
gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -c "$HERE/ref_absent_simd.c" -o "$OUT/obj/ref_absent_simd.o"
rm -f "$OUT/libvpxfull.a"; ar rcs "$OUT/libvpxfull.a" "$OUT"/obj/*.o

# ---- the frame driver, unchanged and patched ---------------------------------------------------------
DF="$L/vp9/decoder/vp9_decodeframe.c"
gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -include "$HERE/ref_decodeframe_prelude.h" -c "$DF" -o "$OUT/decodeframe_unchanged.o"
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
python3 "$HERE/patch_decodeframe.py" "$DF" "$TMP/vp9_decodeframe_patched.c"
gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -include "$HERE/ref_decodeframe_prelude.h" -c "$TMP/vp9_decodeframe_patched.c" -o "$OUT/decodeframe_patched.o"
python3 "$HERE/patch_decodeframe.py" "$DF" "$TMP/vp9_decodeframe_patched_mt.c" --mt
gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -include "$HERE/ref_decodeframe_prelude.h" -c "$TMP/vp9_decodeframe_patched_mt.c" -o "$OUT/decodeframe_patched_mt.o"
python3 "$HERE/patch_decodeframe.py" --decoder-c "$L/vp9/decoder/vp9_decoder.c" "$TMP/vp9_decoder_patched.c"
gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -c "$TMP/vp9_decoder_patched.c" -o "$OUT/decoder_patched.o"
rm -rf "$TMP"
# the patched driver variants take the patched vp9_decoder.o: the archive's own copy must not be pulled in first
PATCHED="$OUT/decodeframe_patched.o $OUT/decoder_patched.o"

# ---- the command-line tools and their third-party C++ (libwebm, libyuv: CONFIG_WEBM_IO / CONFIG_LIBYUV) -
TCF="-O2 -fPIC -w -ffunction-sections -fdata-sections $INC -I$L/third_party/libwebm -I$L/third_party/libyuv/include"
for f in vpxdec vpxenc args ivfdec ivfenc md5_utils tools_common y4menc y4minput rate_hist vpxstats warnings; do
  [ -f "$OUT/tools/$f.o" ] || gcc -std=gnu99 $TCF -c "$L/$f.c" -o "$OUT/tools/$f.o"
done
for f in webmdec webmenc; do [ -f "$OUT/tools/$f.o" ] || g++ -std=gnu++11 $TCF -c "$L/$f.cc" -o "$OUT/tools/$f.o"; done
for f in $L/third_party/libwebm/mkvparser/*.cc $L/third_party/libwebm/mkvmuxer/*.cc $L/third_party/libwebm/common/*.cc \
         $L/third_party/libyuv/source/{cpu_id,planar_functions,row_common,row_any,row_gcc,scale,scale_any,scale_argb,scale_common,scale_gcc}.cc; do
  o="$OUT/tools/tp_$(basename "$f" .cc).o"; [ -f "$o" ] || g++ -std=gnu++11 $TCF -c "$f" -o "$o"
done
DEC_TOOLS="$OUT/tools/vpxdec.o $OUT/tools/args.o $OUT/tools/ivfdec.o $OUT/tools/md5_utils.o $OUT/tools/tools_common.o $OUT/tools/y4menc.o \
  $OUT/tools/webmdec.o $OUT/tools/tp_mkvparser.o $OUT/tools/tp_mkvreader.o $OUT/tools/tp_cpu_id.o $OUT/tools/tp_planar_functions.o \
  $OUT/tools/tp_row_common.o $OUT/tools/tp_row_any.o $OUT/tools/tp_row_gcc.o $OUT/tools/tp_scale.o $OUT/tools/tp_scale_any.o \
  $OUT/tools/tp_scale_argb.o $OUT/tools/tp_scale_common.o $OUT/tools/tp_scale_gcc.o"
ENC_TOOLS="$OUT/tools/vpxenc.o $OUT/tools/args.o $OUT/tools/ivfdec.o $OUT/tools/ivfenc.o $OUT/tools/tools_common.o $OUT/tools/y4minput.o \
  $OUT/tools/rate_hist.o $OUT/tools/vpxstats.o $OUT/tools/warnings.o $OUT/tools/webmenc.o $OUT/tools/tp_mkvmuxer.o $OUT/tools/tp_mkvmuxerutil.o \
  $OUT/tools/tp_mkvwriter.o $OUT/tools/tp_file_util.o $OUT/tools/tp_hdr_util.o $OUT/tools/tp_mkvparser.o $OUT/tools/tp_mkvreader.o \
  $OUT/tools/tp_cpu_id.o $OUT/tools/tp_planar_functions.o $OUT/tools/tp_row_common.o $OUT/tools/tp_row_any.o $OUT/tools/tp_row_gcc.o \
  $OUT/tools/tp_scale.o $OUT/tools/tp_scale_any.o $OUT/tools/tp_scale_argb.o $OUT/tools/tp_scale_common.o $OUT/tools/tp_scale_gcc.o"

# ---- CPU stream oracle ------------------------------------------------------------------------------
gcc -std=gnu99 $CF -Wall -Wno-unused-function -include "$OUT/simd_to_c.h" -c "$HERE/ref_stream_wraps.c" -o "$OUT/ref_stream_wraps.o"
LINK="-Wl,--gc-sections -lm -lpthread"
g++ -o "$OUT/vpxdec_cA" $DEC_TOOLS "$OUT/decodeframe_unchanged.o" "$OUT/ref_stream_wraps.o" "$OUT/libvpxfull.a" $LINK
g++ -o "$OUT/vpxdec_c" $DEC_TOOLS $PATCHED "$OUT/ref_stream_wraps.o" "$OUT/libvpxfull.a" $LINK
g++ -o "$OUT/vpxenc_c" $ENC_TOOLS $PATCHED "$OUT/ref_stream_wraps.o" "$OUT/libvpxfull.a" $LINK

# ---- spatial-layer / intra-only stream maker: own driver of the reference's encoder API, decoded by the oracle as it goes
gcc -std=gnu99 -O2 -w $INC -c "$HERE/ref_svc_encode.c" -o "$OUT/ref_svc_encode.o"
g++ -o "$OUT/ref_svc_encode" "$OUT/ref_svc_encode.o" $PATCHED "$OUT/ref_stream_wraps.o" "$OUT/libvpxfull.a" $LINK

# ---- the product: the same objects against libvp9hip_shim.so ----------------------------------------
if [ -f "$ROOT/shim/build/libvp9hip_shim.so" ]; then
  RP="-Wl,-rpath,\$ORIGIN -Wl,-rpath,\$ORIGIN/../../cuda-vp9_amd"
  g++ -o "$ROOT/shim/build/vpxdec_hipA" $DEC_TOOLS "$OUT/decodeframe_unchanged.o" "$OUT/libvpxfull.a" \
      -L"$ROOT/shim/build" -lvp9hip_shim -L"$ROOT/cuda-vp9_amd" -lvp9hip $RP $LINK
  g++ -o "$ROOT/shim/build/vpxdec_hip" $DEC_TOOLS $PATCHED "$OUT/libvpxfull.a" \
      -L"$ROOT/shim/build" -lvp9hip_shim -L"$ROOT/cuda-vp9_amd" -lvp9hip $RP $LINK
  g++ -o "$ROOT/shim/build/vpxdec_hip_mt" $DEC_TOOLS "$OUT/decodeframe_patched_mt.o" "$OUT/decoder_patched.o" "$OUT/libvpxfull.a" \
      -L"$ROOT/shim/build" -lvp9hip_shim -L"$ROOT/cuda-vp9_amd" -lvp9hip $RP $LINK
  # bring-up mode: the reference's CPU reconstruction with its run-time dispatch pointers assigned to the _hip twins (E13)
  TMP2="$(mktemp -d)"
  python3 "$HERE/patch_decodeframe.py" --decoder-c "$L/vp9/decoder/vp9_decoder.c" "$TMP2/vp9_decoder_rtcd.c" --rtcd
  gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -c "$TMP2/vp9_decoder_rtcd.c" -o "$OUT/decoder_rtcd.o"
  rm -rf "$TMP2"
  gcc -std=gnu99 $CF -include "$OUT/simd_to_c.h" -c "$ROOT/shim/vp9hip_rtcd_install.c" -o "$OUT/rtcd_install.o"
  g++ -o "$ROOT/shim/build/vpxdec_rtcd" $DEC_TOOLS "$OUT/decodeframe_patched.o" "$OUT/decoder_rtcd.o" "$OUT/rtcd_install.o" \
      "$OUT/ref_stream_wraps.o" "$OUT/libvpxfull.a" -L"$ROOT/cuda-vp9_amd" -lvp9hip $RP $LINK
  echo "built shim/build/vpxdec_hipA, vpxdec_hip, vpxdec_hip_mt, vpxdec_rtcd"
else
  echo "build_refvpx: shim/build/libvp9hip_shim.so absent — run make -C shim first for vpxdec_hip*"
fi
# ---- the format's constant tables of the product's bitstream front-end: regenerate from the reference's objects and
# compare with the committed file (cuda-vp9_amd/csrc/fe/vp9fe_tables.inc)
gcc -std=gnu99 -O1 -w $INC -include "$OUT/simd_to_c.h" "$HERE/dump_vp9_tables.c" "$OUT/libvpxfull.a" -lm -lpthread -o "$OUT/tools/dump_vp9_tables"
"$OUT/tools/dump_vp9_tables" > "$OUT/vp9fe_tables.check"
if ! cmp -s "$OUT/vp9fe_tables.check" "$ROOT/cuda-vp9_amd/csrc/fe/vp9fe_tables.inc"; then
  echo "build_refvpx: cuda-vp9_amd/csrc/fe/vp9fe_tables.inc differs from what the reference's objects hold"; exit 1
fi
echo "built $OUT/{vpxdec_cA,vpxdec_c,vpxenc_c}; vp9fe_tables.inc verified"
