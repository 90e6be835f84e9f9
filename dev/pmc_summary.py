"""Average PMC counter values per dispatch of one kernel from rocprofv3
counter-collection CSVs:
  python3 dev/pmc_summary.py <kernel-regex> <out.json> <dir-or-csv> [...]
Every directory is searched for *counter_collection.csv.  Also records the
register / LDS / scratch footprint columns of the dispatch rows."""
import csv, glob, json, os, re, sys

rx = re.compile(sys.argv[1])
out_path = sys.argv[2]
files = []
for p in sys.argv[3:]:
    if os.path.isdir(p):
        files += glob.glob(os.path.join(p, '**', '*counter_collection.csv'), recursive=True)
    else:
        files.append(p)
sums, counts, meta = {}, {}, {}
for f in files:
    for r in csv.DictReader(open(f)):
        if not rx.search(r['Kernel_Name']):
            continue
        c = r['Counter_Name']
        sums[c] = sums.get(c, 0.0) + float(r['Counter_Value'])
        counts[c] = counts.get(c, 0) + 1
        for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size',
                  'Scratch_Size', 'Grid_Size', 'Workgroup_Size', 'Kernel_Name'):
            if k in r:
                meta[k] = r[k]
res = {'kernel_regex': sys.argv[1], 'dispatch': meta,
       'per_launch': {c: sums[c] / counts[c] for c in sorted(sums)},
       'launches_averaged': {c: counts[c] for c in sorted(counts)}}
json.dump(res, open(out_path, 'w'), indent=1)
print(json.dumps(res['per_launch'], indent=1))
