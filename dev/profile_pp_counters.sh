#!/bin/bash
# PMC passes over the Lucy-Richardson iteration kernel (dev/run_pp.py 28 28:
# one launch = 28 frames of 2048 x 2048), one rocprofv3 run per pass.
# Output: gpurun_out/pmcpp_$1/pp_counters.json
set -o pipefail
tag=${1:-x}
out=gpurun_out/pmcpp_$tag
mkdir -p $out
export TMPDIR=/tmp
pass() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-include-regex lr_fused \
        --output-format csv -d $out/$name -o $name -- python3 dev/run_pp.py 28 28 > $out/$name.log 2>&1
}
pass a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS || exit 2
pass m1 FETCH_SIZE TCC_HIT_sum || exit 3
pass m2 WRITE_SIZE TCC_MISS_sum TCC_TAG_STALL_sum || exit 4
pass m3 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum || exit 5
pass t TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 6
pass u TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 8
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 7
python3 dev/pmc_summary.py lr_fused $out/pp_counters.json $out
