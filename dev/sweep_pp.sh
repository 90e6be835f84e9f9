#!/bin/bash
# sweep the Lucy-Richardson prefetch depth x batch size (tuning build)
export MERFISH_HIP_LIB=$PWD/dev/variants/libpp_tune.so
for p in 1 2 4 6 8; do
  echo "P=$p"
  MF_LR_P=$p timeout -k 10 120 python3 -u dev/bench_preprocess.py --images 14 --batches 1,2,4,14 --reps 2 2>&1 \
    | grep -o "ms_per_image\": [0-9.]*, \"batch\": [0-9]*" || exit 1
done
