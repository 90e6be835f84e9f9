#!/bin/bash
# Profiling passes for one round (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the bench command
#   2./3. separate PMC passes (FETCH_SIZE, WRITE_SIZE) over dev/run_decode.py
# Output under gpurun_out/prof_$1/.
set -o pipefail
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- \
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-pipeline --no-preprocess --no-codebooks --no-config4 --no-image-output > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex decode_kernel --output-format csv \
    -d $out/fetch -o fetch -- python3 dev/run_decode.py > $out/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex decode_kernel --output-format csv \
    -d $out/write -o write -- python3 dev/run_decode.py > $out/write.log 2>&1 || exit 3
find $out -name '*.csv' | head -20
