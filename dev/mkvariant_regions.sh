#!/bin/bash
# dev/mkvariant_regions.sh NAME [-Dflags...]: dev build of regions.hip linked with the
# other objects of the shipped library into dev/variants/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p dev/variants /tmp/mfvar
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imerlin_amd/csrc \
  "$@" -c merlin_amd/csrc/regions.hip -o /tmp/mfvar/regions_$name.o
b=merlin_amd/csrc/build
hipcc --offload-arch=gfx950 -shared -fPIC -o dev/variants/lib_$name.so /tmp/mfvar/regions_$name.o \
  $b/common.o $b/decode_g0.o $b/decode_g1.o $b/decode_g2.o $b/decode_g3.o $b/preprocess.o $b/lowpass.o
echo built dev/variants/lib_$name.so
