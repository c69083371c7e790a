#!/bin/bash
# Developer tool: build a variant of the library with extra device-compile flags, for A/B runs on the GPU box (ADYPT_LIB=<path>).
#   tools/build_variant.sh <name> [--transform adypt_amd/csrc/measure/x.py]... [-DADYPT_PATH_SLOTS=320 ...]   ->  adypt_amd/libadypt_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd "$ROOT/adypt_amd/csrc"
make -s -j8 all
SRC=device
while [ "$1" = "--transform" ]; do   # measurement variants live as source transforms under csrc/measure/: applied (in the order given) to a scratch copy of the device sources
    if [ "$SRC" = "device" ]; then SRC=.variant_$name; rm -rf $SRC; cp -r device $SRC; fi   # (a sibling of device/: the relative includes keep working)
    python3 "$ROOT/$2" "$PWD/$SRC"; shift 2
done
FLAGS="-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-parameter -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize -mllvm -disable-machine-licm -DADYPT_BUILD"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC/tracer.hip -o build/tracer_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libadypt_$name.so build/config.o build/scene_loader.o build/image_loader.o build/jpeg_decoder.o build/sbvh_builder.o \
    build/wide_builder.o build/host_api.o build/tracer_$name.o build/multi.o -lz -lpthread -ldl -lrt -Wl,--no-undefined
echo "built adypt_amd/libadypt_$name.so ($*)"
