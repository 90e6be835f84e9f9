#!/bin/bash
# rocprofv3 kernel trace of dev/run_serial.py; prints the per-FOV time of the library kernels
# usage (GPU box): bash dev/serial_trace.sh <tag>
set -o pipefail
tag=${1:-x}
export TMPDIR=/tmp
out=gpurun_out/serial_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 dev/run_serial.py > $out.log 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
rows = []
for r in csv.DictReader(open(f)):
    n = r['Name']; c = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    lib = (('anonymous namespace)::' in n or 'mf_decode::' in n) and 'at::' not in n)
    if lib and c % 12 == 0:
        rows.append((c // 12 * avg, c // 12, avg, n.replace('(anonymous namespace)::', '').replace('mf_decode::', '')[:60]))
        tot += c // 12 * avg
for per, k, avg, n in sorted(rows, reverse=True):
    print('%7.1f us/FOV  %d x %6.1f  %s' % (per, k, avg, n))
print('total %.1f' % tot)
PY
