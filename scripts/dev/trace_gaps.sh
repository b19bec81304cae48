#!/bin/bash
# Dev: rocprofv3 kernel trace of the bench (no counters) -> inter-kernel gaps of the last step.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/trace_gaps
rm -rf $OUT
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --steps 5 --warmup 1 --no_cpu_baseline --no_profile > $OUT.log 2>&1 )
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_inputs' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]          # one full step
step = rows[a:b]
wall = (int(rows[b]['Start_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
gaps = collections.defaultdict(list)
for p, r in zip(step[:-1], step[1:]):
    gaps[r['Kernel_Name'][:36]].append((int(r['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3)
print('step wall %.1f us, kernels busy %.1f us, idle %.1f us (%d kernels)' % (wall, busy, wall - busy, len(step)))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print('  gap before %-38s n=%3d  mean %.2f us  total %.1f us' % (k, len(v), sum(v) / len(v), sum(v)))
PY
