#!/bin/bash
# Dev: training step time at 12 / 64 / 256 windows per GPU, eager and as a replayed HIP graph; then the kernel list of the
# 256-window graph run (rocprofv3 --kernel-trace --stats) so that library kernels still on the path show up by name.
TAG=${1:-r02}
mkdir -p gpurun_out
for bs in 12 64 256; do
  for g in "" "--graph"; do
    echo "bs_train $bs $g: $(python scripts/train.py --steps 10 --bs_train $bs $g --json 2>&1 | tail -1)"
  done
done | tee gpurun_out/${TAG}_train_batch_scaling.txt
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train_prof -o t -- python $R/scripts/train.py --steps 5 --bs_train 256 --graph > $R/gpurun_out/${TAG}_train_prof.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/${TAG}_train_prof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel,calls,total_ms,avg_us,pct')
for r in rows[:40]:
    print(r['Name'][:90], r['Calls'], '%.2f' % (float(r['TotalDurationNs']) / 1e6), '%.1f' % (float(r['AverageNs']) / 1e3), '%.1f' % (100 * float(r['TotalDurationNs']) / tot), sep=',')
PY
