#!/bin/bash
# run_fovs period (dev/pipedepth.py) for every dev/variants/lib_*.so
cd "$(dirname "$0")/.."
for f in dev/variants/lib_*.so; do
  echo "== $f"
  MERFISH_HIP_LIB=$PWD/$f timeout -k 10 200 python3 dev/pipedepth.py 2>/dev/null | grep "frameLag 2" || exit 1
  MERFISH_HIP_LIB=$PWD/$f timeout -k 10 120 python3 dev/time_decode.py 2>/dev/null | head -1
done
