#!/bin/bash
# dev/mkvariant_pp.sh NAME [-Dflags...]: dev build of preprocess.hip linked with
# the existing objects into dev/variants/libpp_NAME.so (MERFISH_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p dev/variants /tmp/mfvar
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imerlin_amd/csrc \
  "$@" -c merlin_amd/csrc/preprocess.hip -o /tmp/mfvar/pp_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o dev/variants/libpp_$name.so /tmp/mfvar/pp_$name.o \
  merlin_amd/csrc/build/common.o merlin_amd/csrc/build/regions.o merlin_amd/csrc/build/decode.o
echo built dev/variants/libpp_$name.so
