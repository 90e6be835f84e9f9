#!/bin/bash
# PMC passes over the decode kernel of config 5 (dev/run_cb5.py).  GPU box: bash dev/pmc_cb5.sh <tag>
set -o pipefail
tag=${1:-x}
export TMPDIR=/tmp
out=gpurun_out/pmc_cb5_$tag
mkdir -p $out
rocprofv3 -L > $out/avail.txt 2>&1
pass() { local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-include-regex decode_kernel --output-format csv -d $out/$name -o $name -- python3 dev/run_cb5.py > $out/$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS || exit 1
pass b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE || exit 2
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]; print('%-40s %16.1f  (%d)' % (k, sum(v) / len(v), len(v)))
PY
