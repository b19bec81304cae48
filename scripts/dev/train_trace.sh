#!/bin/bash
# Dev: kernel trace of scripts/train.py at a given batch: kernel-time sum vs wall per step, top kernels by time.
export TMPDIR=/tmp
R=$PWD; BS=${1:-12}
OUT=$R/gpurun_out/train_trace_$BS; rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/scripts/train.py --steps 10 --bs_train $BS --no_graph --json > $OUT.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob('$OUT/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last full step: between the last two adam launches
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
s, e = idx[-2] + 1, idx[-1] + 1
seg = rows[s:e]
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
print('launches', len(seg), 'wall ms %.3f' % wall, 'kernel-time sum ms %.3f' % busy)
agg = {}
for r in seg:
    k = r['Kernel_Name'].split('(')[0][-50:]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %-52s %4d  %8.1f us  avg %6.1f' % (k, n, t, t / n))
# launch sequence of that step (name, us), for locating the copies / fills by their neighbours
with open('$OUT.seq.txt', 'w') as o:
    for r in seg:
        o.write('%-60s %7.1f\n' % (r['Kernel_Name'].split('(')[0][-60:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
