#!/bin/bash
# kernel timeline of the training step (start / end / queue per dispatch): trace_train.sh <tag> <train.py args...>
tag=$1; shift
out=${PROF_OUT:-gpurun_out/r04c}
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace -d /tmp/trace_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/train.py --steps 4 --warmup 3 --json "$@" > /tmp/trace_$tag.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/trace_$tag/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
keep = rows[int(n * 0.70):]          # the last steps
with open('$GRAFT_REPO_ROOT/$out/train_trace_$tag.csv', 'w') as f:
    w = csv.writer(f)
    w.writerow(['start_ns', 'end_ns', 'queue', 'name'])
    t0 = int(keep[0]['Start_Timestamp'])
    for r in keep:
        w.writerow([int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Queue_Id', ''), r['Kernel_Name'][:60]])
PY
tail -1 /tmp/trace_$tag.log | cut -c1-120
