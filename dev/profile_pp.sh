#!/bin/bash
# rocprofv3 passes over the preprocess stage: kernel stats of the bench command,
# then PMC passes (separate runs) over dev/run_pp.py.  Output: gpurun_out/prof_pp_$1/
set -o pipefail
tag=${1:-x}; batch=${2:-29}
out=gpurun_out/prof_pp_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 -u dev/bench_preprocess.py --images 58 --batches $batch --cpu > $out/bench.json 2> $out/bench.err || exit 5
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- \
    python3 dev/bench_preprocess.py --images 58 --batches $batch > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES \
    --kernel-include-regex lr_fused --output-format csv -d $out/sq -o sq -- python3 dev/run_pp.py $batch $batch > $out/sq.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex lr_fused --output-format csv \
    -d $out/fetch -o fetch -- python3 dev/run_pp.py $batch $batch > $out/fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex lr_fused --output-format csv \
    -d $out/write -o write -- python3 dev/run_pp.py $batch $batch > $out/write.log 2>&1 || exit 4
find $out -name '*.csv' | head -20
