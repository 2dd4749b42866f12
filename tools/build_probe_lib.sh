#!/bin/bash
# Probe build of the library for the stamp tools (tools/island_stamps.py, tools/conv_stamps.py): the kernels compiled with
# -DVP9HIP_STAMPS, everything else from the product's objects.  Output: tools/build/libvp9hip_stamps.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(dirname "$HERE")"
mkdir -p "$HERE/build/obj"
for f in lf_kernels inter_kernels intra_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DVP9HIP_STAMPS -I"$ROOT/include" \
    -c "$ROOT/cuda-vp9_amd/csrc/$f.hip" -o "$HERE/build/obj/$f.o" &
done
wait
OBJS=$(ls "$ROOT"/cuda-vp9_amd/build/*.o | grep -v "/lf_kernels.o\|/inter_kernels.o\|/intra_kernels.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$HERE/build/libvp9hip_stamps.so" $OBJS "$HERE"/build/obj/{lf_kernels,inter_kernels,intra_kernels}.o -lpthread
echo "built $HERE/build/libvp9hip_stamps.so"
