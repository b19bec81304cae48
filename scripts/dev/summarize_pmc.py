"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import collections, csv, glob, json, sys
out = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_%s_%s/*counter_collection.csv' % (sys.argv[1], counter))
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == counter:
                agg[row['Kernel_Name']].append(float(row['Counter_Value']))
    out[counter] = {k: {'mean': sum(v) / len(v), 'n': len(v), 'max': max(v)} for k, v in agg.items() if 'empose' in k}
json.dump(out, open('gpurun_out/pmc_%s.json' % sys.argv[1], 'w'), indent=1)
for c, d in out.items():
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]['mean'] * kv[1]['n']):
        print(c, k[:90], 'mean %.1f' % v['mean'], 'n', v['n'], 'max %.1f' % v['max'])
