#!/bin/bash
# Dev: launch sequence of one headline forward (python bench.py ...): every kernel / copy in order with its duration.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/bench_trace; rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_profile "$@" > $OUT.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob('$OUT/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_inputs_kernel' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
seg = rows[s:e]
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
print('launches', len(seg), 'wall ms %.3f' % wall, 'kernel-time sum ms %.3f' % busy)
out, i = [], 0
name = lambda r: r['Kernel_Name'].split('(')[0].split('::')[-1][-40:]
while i < len(seg):
    j = i
    while j < len(seg) and name(seg[j]) == name(seg[i]): j += 1
    out.append('%s x%d (%.0f us)' % (name(seg[i]), j - i, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg[i:j]) / 1e3))
    i = j
print('\n'.join(out))
PY
