#!/bin/bash
# Second set of PMC passes over the decode kernel (dev/run_decode.py): what the
# VALU instructions are, how the vector-memory path behaves, TA / L1 stalls.
# One rocprofv3 run per pass.  Output: gpurun_out/pmc2_$1/
set -o pipefail
tag=${1:-x}
out=gpurun_out/pmc2_$tag
mkdir -p $out
export TMPDIR=/tmp
pass() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-include-regex decode_kernel \
        --output-format csv -d $out/$name -o $name -- python3 dev/run_decode.py > $out/$name.log 2>&1
}
pass a SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU || exit 1
pass b SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM || exit 2
pass c TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 3
pass d SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH SQ_CYCLES SQ_WAVE_CYCLES || exit 4
pass e SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_INSTS || exit 5
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decode_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]; print('%-40s %16.1f  (%d)' % (k, sum(v) / len(v), len(v)))
PY
