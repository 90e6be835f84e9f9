#!/bin/bash
# pipelined bench.py period for every dev/variants/lib_*.so (two runs each, alternating)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for f in dev/variants/lib_*.so; do
  MERFISH_HIP_LIB=$PWD/$f timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-host-pipeline --no-preprocess --no-codebooks 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f'.split('/')[-1], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['alone']['kernel_ms'],4))" || exit 1
done; done
