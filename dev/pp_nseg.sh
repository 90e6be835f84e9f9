#!/bin/bash
# dev/pp_nseg.sh BATCH IMAGES nseg...: ms per frame of the tuning build at a batch size, by segment count
export MERFISH_HIP_LIB=dev/variants/lib_pp_tune.so
b=$1; im=$2; shift; shift
for n in auto "$@"; do
  if [ $n = auto ]; then unset MF_LR_NSEG; else export MF_LR_NSEG=$n; fi
  echo -n "batch $b nseg $n: "
  timeout -k 10 120 python dev/bench_preprocess.py --images $im --batches $b --reps 2 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms' % d['ms_per_image'])" || exit 1
done
