#!/bin/bash
# compare preprocess variant libraries (dev/mkvariant_pp.sh) x batch size
for v in main "$@"; do
  echo "variant=$v"
  if [ $v = main ]; then unset MERFISH_HIP_LIB; else export MERFISH_HIP_LIB=$PWD/dev/variants/libpp_$v.so; fi
  timeout -k 10 120 python3 -u dev/bench_preprocess.py --images 28 --batches 7,14,28 --reps 2 2>&1 \
    | grep -o "ms_per_image\": [0-9.]*, \"batch\": [0-9]*" || exit 1
done
