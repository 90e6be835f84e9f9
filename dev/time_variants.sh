#!/bin/bash
# time every dev/variants/lib_*.so with dev/time_decode.py (one process each)
cd "$(dirname "$0")/.."
for f in dev/variants/lib_*.so; do
  MERFISH_HIP_LIB=$PWD/$f timeout -k 10 120 python3 dev/time_decode.py 2>/dev/null | sed "s#$PWD/dev/variants/##" || exit 1
done
