#!/bin/bash
# dev/mkvariant.sh NAME [-Dflags...]: dev build of decode.hip linked with the
# existing common/regions objects into dev/variants/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p dev/variants /tmp/mfvar
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imerlin_amd/csrc \
  -DMF_DEV_BUILD "$@" -c merlin_amd/csrc/decode.hip -o /tmp/mfvar/decode_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o dev/variants/lib_$name.so /tmp/mfvar/decode_$name.o \
  merlin_amd/csrc/build/common.o merlin_amd/csrc/build/regions.o merlin_amd/csrc/build/preprocess.o merlin_amd/csrc/build/lowpass.o
echo built dev/variants/lib_$name.so
