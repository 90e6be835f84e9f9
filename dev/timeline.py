"""Compact GPU timeline of the steady state from a rocprofv3 kernel_trace.csv:
for one period between decode launches k and k+1, list big kernels with
start offsets, stream/queue ids, and idle gaps on the union of all queues."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decode_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
t1 = int(rows[b]['Start_Timestamp'])
print('period us', (t1 - t0) / 1e3)
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e < t0 or s > t1: continue
    ev.append((s, e, r))
busy = 0; cur_s = cur_e = None
for s, e, r in ev:
    s = max(s, t0); e = min(e, t1)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('union busy us', busy / 1e3)
for s, e, r in ev:
    d = (e - s) / 1e3
    if d > 40:
        n = re.sub(r'rocprim::ROCPRIM_\d+_NS::detail::', 'rp::', r['Kernel_Name'])
        print('%9.1f %8.1f q%s  %s' % ((s - t0) / 1e3, d, r.get('Queue_Id', '?'), n[:80]))
