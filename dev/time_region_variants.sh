#!/bin/bash
# rocprofv3 kernel trace of dev/run_serial.py for every dev/variants/lib_*.so:
# per-FOV time of the sort / region kernels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for f in dev/variants/lib_*.so; do
  n=$(basename $f .so)
  export MERFISH_HIP_LIB=$PWD/$f
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rv/$n -o s -- python3 dev/run_serial.py > gpurun_out/prof_rv_$n.log 2>&1 || { echo "$n failed"; tail -3 gpurun_out/prof_rv_$n.log; exit 1; }
  echo "== $n"
  python3 - "$(find gpurun_out/prof_rv/$n -name "*kernel_stats.csv" | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:30]:
    name = r['Name'].replace('(anonymous namespace)::', '')
    if 'at::' in name or 'rocclr' in name or 'decode_kernel' in name or 'indexing' in name: continue
    short = 'rocprim sort kernel' if 'radix_sort' in name else name[:60]
    print('%5d calls avg %8.1f us  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, short))
PY
done
