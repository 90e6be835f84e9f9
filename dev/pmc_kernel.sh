#!/bin/bash
# Three PMC passes over one kernel of dev/run_serial.py (name regex $1), averages printed.
# usage (GPU box): bash dev/pmc_kernel.sh region_stats <tag>
set -o pipefail
k=$1; tag=${2:-x}
export TMPDIR=/tmp
out=gpurun_out/pmck_$tag
mkdir -p $out
pass() { local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-include-regex "$k" --output-format csv -d $out/$name -o $name -- python3 dev/run_serial.py > $out/$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS || exit 1
pass b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 2
pass c FETCH_SIZE TCC_EA0_RDREQ_sum || exit 3
pass d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 4
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        acc['_vgpr'].append(float(r.get('VGPR_Count') or 0)); acc['_lds'].append(float(r.get('LDS_Block_Size') or 0))
for k in sorted(acc):
    v = acc[k]; print('%-40s %16.1f  (%d)' % (k, sum(v) / len(v), len(v)))
PY
