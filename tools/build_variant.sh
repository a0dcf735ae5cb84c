#!/bin/bash
# A second build of the library with extra compiler flags, for same-box A/B runs through FNR_LIB_PATH:
#   bash tools/build_variant.sh <name> "<extra flags>"   ->  fruitnerf_amd/lib/variants/<name>/libfruitnerf_hip.so
# (the .so travels to the GPU box with the snapshot; objects under build/obj_<name>)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/fruitnerf_amd/lib/variants/$NAME
make -j8 -C $ROOT/fruitnerf_amd/csrc OBJDIR=../../build/obj_$NAME OUT=../lib/variants/$NAME/libfruitnerf_hip.so EXTRA="$*" >/dev/null
ls -la $ROOT/fruitnerf_amd/lib/variants/$NAME/libfruitnerf_hip.so
