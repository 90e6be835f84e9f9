#!/bin/bash
# dev/mkvariant_pp.sh NAME [-Dflags...]: dev build of preprocess.hip (MF_LR_TUNE:
# MF_LR_PAIRS / MF_LR_SPLIT / MF_LR_P read from the environment) linked with the
# existing objects into dev/variants/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p dev/variants /tmp/mfvar
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imerlin_amd/csrc \
  -DMF_LR_TUNE $PPFLAGS "$@" -c merlin_amd/csrc/preprocess.hip -o /tmp/mfvar/pp_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC -o dev/variants/lib_$name.so /tmp/mfvar/pp_$name.o \
  merlin_amd/csrc/build/common.o merlin_amd/csrc/build/regions.o merlin_amd/csrc/build/decode.o
echo built dev/variants/lib_$name.so
