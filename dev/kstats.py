"""Per-FOV time of every kernel in a rocprofv3 kernel_stats.csv (bench run):
python3 dev/kstats.py gpurun_out/prof_x/kt/kt_kernel_stats.csv [fovs]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
fovs = int(sys.argv[2]) if len(sys.argv) > 2 else 23
tot = 0.0
for r in rows:
    n = r['Name']
    if 'at::' in n and 'cub' not in n:
        continue
    if int(r['Calls']) < fovs - 5:
        continue
    short = re.sub(r'\(anonymous namespace\)::|mf_decode::', '', n)
    short = re.sub(r'rocprim::ROCPRIM_\d+_NS::(detail::)?', 'rp::', short)
    per = float(r['TotalDurationNs']) / fovs / 1e3
    if 'decode_kernel' in n:
        per = float(r['AverageNs']) / 1e3
    tot += per
    print('%5s calls %8.1f us/FOV  %s' % (r['Calls'], per, short[:110]))
print('total %.1f us/FOV' % tot)
