#!/bin/bash
# dev/pp_sweep.sh LIBTAG [nseg...]: ms per 2048^2 frame of a tuning build by segment count / kernel
export MERFISH_HIP_LIB=dev/variants/lib_$1.so; shift
run() { timeout -k 10 120 python dev/bench_preprocess.py --images ${IMAGES:-56} --batches ${BATCH:-28} | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  frac %.3f' % (d['ms_per_image'], d['roofline']['frac']))"; }
echo "auto pairs"; run || exit 1
for n in "$@"; do echo "nseg=$n pairs"; MF_LR_NSEG=$n run || exit 1; done
echo "auto 1-col"; MF_LR_PAIRS=0 run || exit 1
