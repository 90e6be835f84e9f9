#!/bin/bash
# dev/pp_pad_sweep.sh LIBTAG "padx:padf" ...: ms per frame by J-buffer padding (doubles)
export MERFISH_HIP_LIB=dev/variants/lib_$1.so; shift
for c in "$@"; do
  export MF_LR_PADX=${c%%:*} MF_LR_PADF=${c##*:}
  echo -n "padx=$MF_LR_PADX padf=$MF_LR_PADF  "
  timeout -k 10 120 python dev/bench_preprocess.py --images ${IMAGES:-56} --batches ${BATCH:-28} | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  frac %.3f' % (d['ms_per_image'], d['roofline']['frac']))" || exit 1
done
