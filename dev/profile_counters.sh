#!/bin/bash
# Compute-side PMC passes over the decode kernel (dev/run_decode.py), one
# rocprofv3 run per pass (<= 8 SQ counters each).  Output: gpurun_out/pmc_$1/
set -o pipefail
tag=${1:-x}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -L > $out/counter_list.txt 2>&1 || true
pass() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-include-regex decode_kernel \
        --output-format csv -d $out/$name -o $name -- python3 dev/run_decode.py > $out/$name.log 2>&1
}
pass a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM || exit 2
pass c SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS || exit 3
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 4
python3 dev/pmc_summary.py decode_kernel $out/decode_counters.json $out
