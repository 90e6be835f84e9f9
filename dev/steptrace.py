"""Per-step kernel breakdown from a rocprofv3 kernel_trace.csv (between two
consecutive decode_kernel launches in the steady state)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decode_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = idx[k], idx[k + 1]
acc = collections.OrderedDict()
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name'][:90]; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    acc.setdefault(n, [0, 0]); acc[n][0] += 1; acc[n][1] += d
busy = 0
for n, (c, d) in acc.items():
    busy += d
    if d > 8: print('%-90s %3d %8.1f' % (n, c, d))
print('busy us %.1f window us %.1f kernels %d' % (busy, (int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
