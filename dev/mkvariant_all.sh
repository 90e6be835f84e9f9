#!/bin/bash
# dev/mkvariant_all.sh NAME "decode flags" "regions flags" "preprocess flags":
# dev build of all three kernels files into dev/variants/lib_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1
mkdir -p dev/variants /tmp/mfvar
base="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imerlin_amd/csrc"
hipcc $base $2 -c merlin_amd/csrc/decode.hip -o /tmp/mfvar/decode_$name.o &
hipcc $base $3 -c merlin_amd/csrc/regions.hip -o /tmp/mfvar/regions_$name.o &
hipcc $base -mllvm -amdgpu-sched-strategy=max-ilp $4 -c merlin_amd/csrc/preprocess.hip -o /tmp/mfvar/pp_$name.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o dev/variants/lib_$name.so /tmp/mfvar/decode_$name.o \
  /tmp/mfvar/regions_$name.o /tmp/mfvar/pp_$name.o merlin_amd/csrc/build/common.o
echo built dev/variants/lib_$name.so
