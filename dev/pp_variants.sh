#!/bin/bash
# ms per frame (112 frames, one batch) for every dev/variants/lib_*.so
for f in dev/variants/lib_*.so; do
  echo -n "$(basename $f .so): "
  MERFISH_HIP_LIB=$PWD/$f timeout -k 10 200 python dev/bench_preprocess.py --images 112 --batches 112 --reps 3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms' % d['ms_per_image'])" || exit 1
done
